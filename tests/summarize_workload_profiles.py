"""Turns gpurun_out/prof_<tag>/ of `tests/collect_profiles.sh <tag> --workload configN` into committed summaries under profiles/:
<round>_<name>_bench.json, ..._bench_under_rocprof.json, ..._kernel_stats.csv and ..._pmc_summary.txt (HBM traffic of the DP
kernels from FETCH_SIZE x 2 + WRITE_SIZE, instruction counts per 64 band cells AND per DP row, the slowest problem of a launch
when the bench ran with VGA_TRACE=1).  Leaves profiles/traffic.json and instr.json (config 3, read back by bench.py) alone.
usage: python tests/summarize_workload_profiles.py r4c5 r04 c5"""
import collections, csv, json, os, re, shutil, sys

tag, rnd, name = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
last = lambda f: json.loads(open(os.path.join(src, f)).read().strip().splitlines()[-1])
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"{rnd}_{name}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{rnd}_{name}_bench.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{rnd}_{name}_bench_under_rocprof.json"))
b = last("bench.json")
out = [f"# {b['config']['workload']}",
       f"# bench line: {b['value']:.1f} aligned reads/s, {b['ms_per_step']:.1f} ms per step, roofline.frac {b['roofline']['frac']:.4f} "
       f"(busy {b['roofline']['busy_ms_per_launch']:.1f} ms per launch x {b['roofline']['launches_per_step']:.0f} launches)"]


def sums(path, kernels_of_step):
    tot = collections.Counter()
    per_kernel = collections.Counter()
    rows = list(csv.DictReader(open(path)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    keep = set(ids[:kernels_of_step]) if kernels_of_step else set(ids)
    for r in rows:
        if int(r["Dispatch_Id"]) in keep:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            per_kernel[(r["Kernel_Name"].split("(")[0][:40], r["Counter_Name"])] += float(r["Counter_Value"])
    return tot, per_kernel, len(ids)


vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(src, f"pmc_{c}", "pmc_counter_collection.csv")
    if os.path.exists(p):
        bj = last(f"bench_pmc_{c}.json")
        t, _, n = sums(p, int(bj["roofline"]["launches"]))
        vals[c] = t[c]
        out.append(f"# {c}: {t[c]:.0f} KiB over the {bj['roofline']['launches']} DP dispatches of one step (of {n} in the process)")
if len(vals) == 2:
    bj = last("bench_pmc_FETCH_SIZE.json")
    hbm = (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024
    alg = bj["roofline"]["launches"] * bj["roofline"]["algorithmic_bytes_per_launch"]
    out.append(f"# HBM traffic (gfx950: FETCH_SIZE x 2 + WRITE_SIZE): {hbm:.4g} B per step = {hbm / alg:.2f} x the SURVEY 8d bytes ({alg:.4g})")
p = os.path.join(src, "pmc_INSTS", "pmc_counter_collection.csv")
if os.path.exists(p):
    bi = last("bench_pmc_INSTS.json")
    t, pk, n = sums(p, int(bi["roofline"]["launches"]))
    cells, rows = bi["per_step"]["poa_cells"], bi["per_step"]["poa_rows"]
    out.append(f"# instruction counters of the {bi['roofline']['launches']} DP dispatches of one step ({cells} band cells, {rows} DP rows):")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES"):
        if k in t:
            out.append(f"#   {k}: {t[k]:.4g}  = {t[k] * 64 / cells:.1f} per 64 cells = {t[k] / rows:.1f} per row")
    out.append("# by kernel:")
    for (kn, cn), v in sorted(pk.items()):
        if cn in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVES"):
            out.append(f"#   {kn:40s} {cn}: {v:.4g}")
err = os.path.join(src, "bench.err")
if os.path.exists(err):
    slow = [l.strip() for l in open(err, errors="replace") if "slowest problem" in l]
    if slow:
        key = lambda l: float(re.search(r"slowest problem: ([0-9.]+) ms", l).group(1))
        out.append("# slowest problem of a launch (VGA_TRACE=1, bench.err): " + max(slow, key=key).split("poa:")[-1].strip())
ks = [r for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))) if "k_poa_dp" in r["Name"]]
for r in ks:
    out.append(f"# rocprofv3 --kernel-trace --stats: {r['Name'][:60]} calls {r['Calls']} average {float(r['AverageNs']) / 1e6:.2f} ms total {float(r['TotalDurationNs']) / 1e6:.1f} ms")
open(os.path.join(dst, f"{rnd}_{name}_pmc_summary.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
