#!/bin/bash
# same-box A/B of the POA kernels on config 3: k_poa_dp_t4 (default) vs k_poa_dp_pk (variants build: make -C rs-vgaligner_amd/csrc variants)
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
print('$1', d['value'], 'reads/s', d['ms_per_step'], 'ms/step; dp busy', d['kernels_busy_ms_per_step'].get('poa_band_dp'))"; }
VGA_TRACE=1 timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_t4.json 2> $OUT/bench_t4.err || exit 1
grep -m2 "launch" $OUT/bench_t4.err
show $OUT/bench_t4.json
VGA_LIB=$GRAFT_REPO_ROOT/rs-vgaligner_amd/libvga_hip_variants.so VGA_POA_KERNEL=pk timeout -k 10 400 python bench.py --steps 3 --warmup 1 --cpu-sample 0 > $OUT/bench_pk.json 2> $OUT/bench_pk.err || exit 1
show $OUT/bench_pk.json
