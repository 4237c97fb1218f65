#!/bin/bash
# same-box A/B on config 3: k_poa_dp_w1 (VGA_POA_W1=1) vs k_poa_dp_t4
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-w1ab}
mkdir -p $OUT
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; dp busy', d['kernels_busy_ms_per_step'].get('poa_band_dp'), 'aligned', d['per_step']['aligned'])"; }
VGA_POA_W1=1 VGA_TRACE=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_w1.json 2> $OUT/bench_w1.err || { tail -5 $OUT/bench_w1.err; exit 1; }
grep -m2 "poa: launch" $OUT/bench_w1.err; grep -m2 "handed back\|classic" $OUT/bench_w1.err
show $OUT/bench_w1.json
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_t4.json 2> $OUT/bench_t4.err || exit 1
show $OUT/bench_t4.json
