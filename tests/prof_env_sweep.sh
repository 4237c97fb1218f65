#!/bin/bash
# same-box sweep of one environment variable over bench.py (config 3): bash tests/prof_env_sweep.sh <tag> VAR v1 v2 ...
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$1; VAR=$2; shift 2
mkdir -p $OUT
for v in "$@"; do
  env $VAR=$v timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_${VAR}_$v.json 2> $OUT/bench_${VAR}_$v.err || { echo "FAILED $VAR=$v"; tail -3 $OUT/bench_${VAR}_$v.err; exit 1; }
  python3 -c "
import json
d=json.loads(open('$OUT/bench_${VAR}_$v.json').read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; dp busy', d['kernels_busy_ms_per_step'].get('poa_band_dp'))"
done
