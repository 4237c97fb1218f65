#!/bin/bash
# Fixed costs of one `vgaligner map` process (diagnostics, GPU box): dynamic loading, HIP start, pinned host memory, exit.
# usage: bash tests/prof_startup.sh <tag>
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/startup_${1:-x}
mkdir -p $OUT
EXE=$REPO/rs-vgaligner_amd/vgaligner
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 $REPO/tests/microbench/pinned_time.hip -o /tmp/pinned_time 2> $OUT/build.err || { cat $OUT/build.err; exit 1; }
for i in 1 2; do /tmp/pinned_time 400 ; done > $OUT/pinned.txt 2>&1
cat $OUT/pinned.txt
( time $EXE ) > /dev/null 2> $OUT/usage.txt; grep real $OUT/usage.txt
[ "$ONLY_MICRO" = "1" ] && exit 0
LD_DEBUG=statistics $EXE > /dev/null 2> $OUT/ld_stats.txt
grep -E "total startup|relocation|load" $OUT/ld_stats.txt | head -8
VGA_TRACE=1 python3 $REPO/tests/prof_e2e_cli.py 16 > $OUT/e2e_16.json 2> $OUT/e2e_16.err
cat $OUT/e2e_16.json
grep -E "vgh-trace|map: " $OUT/e2e_16.err
VGA_TRACE=1 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/e2e_10k.json 2> $OUT/e2e_10k.err
cat $OUT/e2e_10k.json
grep -E "vgh-trace|map: |align: " $OUT/e2e_10k.err
VGA_NO_FAST_EXIT=1 VGA_TRACE=1 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/e2e_10k_slow_exit.json 2> $OUT/e2e_10k_slow_exit.err
cat $OUT/e2e_10k_slow_exit.json
grep -E "GAF files written|contexts destroyed|done " $OUT/e2e_10k_slow_exit.err
