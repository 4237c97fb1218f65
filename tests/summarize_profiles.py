"""Turns gpurun_out/prof_<tag>/ (tests/collect_profiles.sh) into the committed summaries under profiles/:
<round>_<tag>_bench_10k.json, ..._bench_10k_under_rocprof.json, ..._kernel_stats.csv, ..._pmc_poa_dp.txt,
profiles/traffic.json (HBM bytes per step of the dominant kernel: FETCH_SIZE x 2 + WRITE_SIZE, in KiB, per
MI355X_MICROARCH.md's gfx950 note) and profiles/instr.json (SQ_INSTS_VALU / SQ_INSTS_SALU per 64 band cells).
usage: python tests/summarize_profiles.py v12 [r02]"""
import csv, json, os, shutil, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "kt", "kt_kernel_stats.csv"), os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{rnd}_{tag}_bench_10k.json"))
shutil.copy(os.path.join(src, "bench_under_rocprof.json"), os.path.join(dst, f"{rnd}_{tag}_bench_10k_under_rocprof.json"))
last = lambda f: json.loads(open(os.path.join(src, f)).read().strip().splitlines()[-1])


def step_rows(path, launches):
    """the counter rows of the timed step: its `launches` dispatches come first; what follows them in the same process (bench.py's
    look at the other remain rule, unless VGA_BENCH_NO_OTHER_RULE=1 was set) is not part of the step"""
    rows = list(csv.DictReader(open(path)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    keep = set(ids[:launches])
    return [r for r in rows if int(r["Dispatch_Id"]) in keep], len(ids)


out, vals, launches, alg = [], {}, 0, 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    b = last(f"bench_pmc_{c}.json")
    rows, n_disp = step_rows(os.path.join(src, f"pmc_{c}", "pmc_counter_collection.csv"), b["roofline"]["launches"])
    out.append(f"# the {b['roofline']['launches']} dispatches of the timed step (of {n_disp} in the process)")
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
open(os.path.join(dst, f"{rnd}_{tag}_pmc_poa_dp.txt"), "w").write("\n".join(out) + "\n")
hbm = (vals["FETCH_SIZE"] * 2 + vals["WRITE_SIZE"]) * 1024
t = {"kernel": "poa_band_dp", "reads": 10000, "read_len": 10000, "fetch_size_kib_per_step": vals["FETCH_SIZE"],
     "write_size_kib_per_step": vals["WRITE_SIZE"],
     "correction": "gfx950: FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE as reported",
     "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace --kernel-include-regex k_poa_dp -- python3 bench.py "
               f"--steps 1 --warmup 0 --cpu-sample 0 (tests/collect_profiles.sh {tag}; raw rows in profiles/{rnd}_{tag}_pmc_poa_dp.txt)",
     "launches_in_pmc_step": launches, "hbm_bytes_per_step": hbm, "algorithmic_bytes_per_step": alg, "ratio": hbm / alg,
     "remain_rule": 1 if (b.get("config") or {}).get("poa_remain_rule") == "first-out-edge" else 0, "tag": tag,
     "note": "bench.py divides hbm_bytes_per_step by its own launches per step; the traceback is fused into the DP kernel, so its reads of the direction bytes are included"}
json.dump(t, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
b, br = last("bench.json"), last("bench_under_rocprof.json")
ks = max((r for r in csv.DictReader(open(os.path.join(src, "kt", "kt_kernel_stats.csv"))) if "k_poa_dp" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]))
print("bench", b["value"], b["ms_per_step"], b["roofline"]["achieved"], b["roofline"]["frac"], "cpu", b["cpu_baseline"]["value"], b["cpu_baseline_all_cores"]["value"])
print("under rocprof", br["value"], "avg_launch_ms", br["roofline"]["avg_launch_ms"], "rocprof avg ms", float(ks["AverageNs"]) / 1e6, "calls", ks["Calls"])
print("traffic ratio", t["ratio"], "launches", launches)

# instruction counters -> profiles/instr.json (read by bench.py for roofline.valu)
ip = os.path.join(src, "pmc_INSTS", "pmc_counter_collection.csv")
if os.path.exists(ip):
    import collections
    tot, kname = collections.Counter(), ""
    lines = ["# rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT "
             "--kernel-trace --kernel-include-regex k_poa_dp -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0", "dispatch,kernel,counter,value"]
    bi = last("bench_pmc_INSTS.json")
    irows, n_disp = step_rows(ip, bi["roofline"]["launches"])
    lines.insert(0, f"# the {bi['roofline']['launches']} dispatches of the timed step (of {n_disp} in the process)")
    for r in irows:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        kname = r["Kernel_Name"].split("(")[0]
        lines.append(",".join([r["Dispatch_Id"], r["Kernel_Name"][:32].replace(",", ";"), r["Counter_Name"], r["Counter_Value"]]))
    for k, v in sorted(tot.items()): lines.append(f"# sum {k} = {v}")
    open(os.path.join(dst, f"{rnd}_{tag}_pmc_insts.txt"), "w").write("\n".join(lines) + "\n")
    cells = (bi.get("per_step") or {}).get("poa_cells") or bi.get("poa_cells_per_step")  # (the band cells of the step the counters saw)
    old = json.load(open(os.path.join(dst, "instr.json")))
    cells = cells or old["cells_per_step"]
    ij = {"kernel": kname, "valu_wave_instr_per_64_cells": round(tot["SQ_INSTS_VALU"] * 64 / cells, 1),
          "salu_wave_instr_per_64_cells": round(tot["SQ_INSTS_SALU"] * 64 / cells, 1),
          "lds_wave_instr_per_64_cells": round(tot["SQ_INSTS_LDS"] * 64 / cells, 1), "cells_per_step": cells,
          "sq_insts_valu_per_step": tot["SQ_INSTS_VALU"], "sq_insts_salu_per_step": tot["SQ_INSTS_SALU"],
          "peak_Gwaveinst_per_s": old["peak_Gwaveinst_per_s"],
          "remain_rule": 1 if (bi.get("config") or {}).get("poa_remain_rule") == "first-out-edge" else 0,
          "source": f"rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... --kernel-trace --kernel-include-regex k_poa_dp -- python3 bench.py --steps 1 --warmup 0 "
                    f"--cpu-sample 0 (tests/collect_profiles.sh {tag}; raw rows in profiles/{rnd}_{tag}_pmc_insts.txt); peak: tests/microbench/valu_issue.hip, "
                    "profiles/r02_valu_issue_microbench.txt (v_max_i32 / VOP3 / DPP / SDWA / v_pk_* at 6 waves per SIMD: 0.53-0.60 T/s)"}
    json.dump(ij, open(os.path.join(dst, "instr.json"), "w"), indent=1)
    print("instr", ij["valu_wave_instr_per_64_cells"], ij["salu_wave_instr_per_64_cells"])
