#!/bin/bash
# Collects a round's evidence on the GPU box into gpurun_out/prof_<tag>/ (copy the summaries into profiles/ afterwards with
# tests/summarize_profiles.py):
#   1. bench.py plain (the driver's command), 2. the same under rocprofv3 --kernel-trace --stats,
#   3./4. PMC passes FETCH_SIZE and WRITE_SIZE (separate runs, kernel-trace only), one bench step each,
#   5. a PMC pass with the instruction counters (SQ_INSTS_*), one bench step.
# usage (on the GPU box, via gpurun):  bash tests/collect_profiles.sh v7 [--workload config4|config5] [--quick]
#   --workload: the same passes on another workload of bench.py (not the driver's line); --quick: skips the FETCH / WRITE passes
set -e
TAG=${1:-vX}; shift || true
WL=""; QUICK=0
while [ $# -gt 0 ]; do
  case "$1" in
    --workload) WL="--workload $2"; shift 2;;
    --quick) QUICK=1; shift;;
    *) echo "unknown argument $1"; exit 2;;
  esac
done
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the counter passes look at the timed step only (bench.py otherwise adds two steps under the other remain rule)
VGA_TRACE=${VGA_TRACE:-0} timeout -k 10 500 python3 $REPO/bench.py $WL $BENCH_EXTRA > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"; tail -c 600 $OUT/bench.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt --output-format csv -- python3 $REPO/bench.py $WL $BENCH_EXTRA --cpu-sample 0 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
echo "kernel trace done"
if [ $QUICK = 0 ]; then
for c in FETCH_SIZE WRITE_SIZE; do
  VGA_BENCH_NO_OTHER_RULE=1 timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/pmc_$c -o pmc --output-format csv -- python3 $REPO/bench.py $WL $BENCH_EXTRA --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_pmc_$c.json 2> $OUT/pmc_$c.err
  echo "pmc $c done"
done
fi
# instruction counters of the same step (the issue-rate fraction in bench.py's roofline.valu comes from these)
VGA_BENCH_NO_OTHER_RULE=1 timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_LDS_BANK_CONFLICT --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/pmc_INSTS -o pmc --output-format csv -- python3 $REPO/bench.py $WL $BENCH_EXTRA --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_pmc_INSTS.json 2> $OUT/pmc_INSTS.err
echo "pmc INSTS done"
find $OUT -name "*.csv" | head -20
