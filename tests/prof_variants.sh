#!/bin/bash
# Same-box comparison of several builds of libvga_hip.so (build_ab/lib<NAME>.so): for each, a short config-3 bench
# (reads/s, DP busy time) and one PMC pass (SQ_INSTS_VALU / SALU per 64 band cells of the POA DP kernel).
# usage (GPU box): bash tests/prof_variants.sh <tag> NAME[:ENV=VAL[,ENV=VAL]] ...      env: READS (4000), STEPS (3), PMC (1)
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
OUT=$REPO/gpurun_out/var_$TAG
mkdir -p $OUT
READS=${READS:-4000}; STEPS=${STEPS:-3}
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  [[ "$spec" == *:* ]] && envs=${spec#*:}
  (
    export VGA_LIB=$REPO/build_ab/lib$name.so
    [ -f "$VGA_LIB" ] || { echo "missing $VGA_LIB"; exit 1; }
    IFS=',' read -ra kvs <<< "$envs"; for kv in "${kvs[@]}"; do [ -n "$kv" ] && export "$kv"; done
    lbl=$(echo "$spec" | tr ':,=' '___')
    timeout -k 10 300 python3 $REPO/bench.py $BENCH_ARGS --reads $READS --steps $STEPS --warmup 1 --cpu-sample 0 > $OUT/bench_$lbl.json 2> $OUT/bench_$lbl.err || { echo "$spec: bench failed"; tail -3 $OUT/bench_$lbl.err; exit 1; }
    if [ "${PMC:-1}" = "1" ]; then
      VGA_BENCH_NO_OTHER_RULE=1 timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/pmc_$lbl -o pmc --output-format csv -- python3 $REPO/bench.py $BENCH_ARGS --reads ${PMC_READS:-2000} --steps 1 --warmup 0 --cpu-sample 0 > $OUT/pmcbench_$lbl.json 2> $OUT/pmc_$lbl.err || { echo "$spec: pmc failed"; tail -3 $OUT/pmc_$lbl.err; }
    fi
    python3 - "$spec" $OUT/bench_$lbl.json $OUT/pmcbench_$lbl.json $OUT/pmc_$lbl <<'PY'
import csv, glob, json, sys, collections
spec, bj, pj, pd = sys.argv[1:5]
b = json.loads(open(bj).read().strip().splitlines()[-1])
line = "%-28s reads/s %8.1f  step %7.1f ms  DP busy %7.1f ms" % (spec, b["value"], b["ms_per_step"], b["kernels_busy_ms_per_step"].get("poa_band_dp", 0))
try:
    p = json.loads(open(pj).read().strip().splitlines()[-1])
    tot = collections.Counter()
    for f in glob.glob(pd + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
    cells = p["per_step"]["poa_cells"]
    line += "  | per 64 cells: VALU %6.1f SALU %6.1f LDS %5.1f" % (tot["SQ_INSTS_VALU"] * 64 / cells, tot["SQ_INSTS_SALU"] * 64 / cells, tot["SQ_INSTS_LDS"] * 64 / cells)
    if tot["SQ_WAVE_CYCLES"]:
        line += "  valu-active/wave-cycles %.3f wait-any %.3f" % (tot["SQ_ACTIVE_INST_VALU"] / tot["SQ_WAVE_CYCLES"], tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"])
except Exception as e:
    line += "  (no pmc: %s)" % e
print(line)
open(pd.rsplit("/", 1)[0] + "/summary.txt", "a").write(line + "\n")
PY
  )
done
