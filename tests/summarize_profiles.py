"""Turns gpurun_out/prof_<tag>/ (tests/collect_profiles.sh) into the committed summaries under profiles/:
r01_<tag>_bench_10k.json, r01_<tag>_bench_10k_under_rocprof.json, r01_<tag>_kernel_stats.csv, r01_<tag>_pmc_poa_dp.txt
and profiles/traffic.json (HBM bytes per step of the dominant kernel: FETCH_SIZE x 2 + WRITE_SIZE, in KiB, per
MI355X_MICROARCH.md's gfx950 note).   usage: python tests/summarize_profiles.py v8"""
import csv, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"r01_{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"r01_{tag}_bench_10k.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"r01_{tag}_bench_10k_under_rocprof.json"))
last = lambda f: json.loads(open(os.path.join(src, f)).read().strip().splitlines()[-1])
out, vals, launches, alg = [], {}, 0, 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(os.path.join(src, f"pmc_{c}", "pmc_counter_collection.csv"))))
    out.append(f"# rocprofv3 --pmc {c} --kernel-trace --kernel-include-regex k_poa_dp -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0")
    out.append("dispatch,kernel,grid,workgroup,counter,value")
    tot = 0.0
    for r in rows:
        out.append(",".join([r["Dispatch_Id"], r["Kernel_Name"][:32].replace(",", ";"), r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"]]))
        tot += float(r["Counter_Value"])
    out.append(f"# sum {c} = {tot}")
    vals[c] = tot
    b = last(f"bench_pmc_{c}.json")
    launches, alg = b["roofline"]["launches"], b["roofline"]["launches"] * b["roofline"]["algorithmic_bytes_per_launch"]
open(os.path.join(dst, f"r01_{tag}_pmc_poa_dp.txt"), "w").write("\n".join(out) + "\n")
hbm = (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024
t = {"kernel": "poa_band_dp", "reads": 10000, "read_len": 10000, "fetch_size_kib_per_step": vals["FETCH_SIZE"],
     "write_size_kib_per_step": vals["WRITE_SIZE"],
     "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
     "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --kernel-include-regex k_poa_dp -- python3 bench.py "
               f"--steps 1 --warmup 0 --cpu-sample 0 (tests/collect_profiles.sh {tag}; raw rows in profiles/r01_{tag}_pmc_poa_dp.txt)",
     "launches_in_pmc_step": launches, "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": alg, "ratio": hbm / alg,
     "note": "bench.py divides hbm_bytes_per_step by its own launches per step; the traceback is fused into the DP kernel, so its reads of the direction bytes are included"}
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
b, br = last("bench.json"), last("bench_under_rocprof.json")
ks = next(r for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))) if "k_poa_dp_pk" in r["Name"])
print("bench", b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline_all_cores"]["value"])
print("under rocprof", br["value"], "avg_launch_ms", br["roofline"]["avg_launch_ms"], "rocprof avg ms", float(ks["AverageNs"]) / 1e6, "calls", ks["Calls"])
print("traffic ratio", t["ratio"], "launches", launches)
