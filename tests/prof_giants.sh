#!/bin/bash
# Same-box comparison of launch shapes for config 4's very long problems (tests/prof_giants.sh <tag> [reads]):
# workgroup size of their launch and raised issue priority.  Prints reads/s, step time and the slowest problem of each run.
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
TAG=${1:-g}; READS=${2:-10000}
OUT=$REPO/gpurun_out/giants_$TAG
mkdir -p $OUT
for spec in ${SPECS:-base:VGA_POA_GIANT_PRIO=0 prio:VGA_POA_GIANT_PRIO=1} $EXTRA_SPECS; do
  name=${spec%%:*}; envs=${spec#*:}
  (
    IFS=',' read -ra kvs <<< "$envs"; for kv in "${kvs[@]}"; do [ -n "$kv" ] && export "$kv"; done
    VGA_TRACE=1 timeout -k 10 400 python3 $REPO/bench.py --workload config4 --reads $READS --steps 2 --warmup 1 --cpu-sample 0 > $OUT/$name.json 2> $OUT/$name.err || { echo "$name failed"; tail -3 $OUT/$name.err; exit 1; }
    python3 - $name $OUT/$name.json <<'PY'
import json, sys
b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-10s reads/s %8.1f  step %7.1f ms  DP busy %7.1f ms  frac %.4f" % (sys.argv[1], b["value"], b["ms_per_step"], b["kernels_busy_ms_per_step"].get("poa_band_dp", 0), b["roofline"]["frac"]))
PY
    grep "slowest problem" $OUT/$name.err | sort -t: -k3 -n | awk '{print "    " $0}' | sort -k5 -n -r | head -3
  )
done
