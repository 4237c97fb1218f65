cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/exit_${1:-x}
mkdir -p $OUT
VGA_POOL_FILL=0.4 VGA_TRACE=1 timeout -k 10 120 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/fill04.json 2> $OUT/fill04.err
grep -E "free list empty|segment .* listed|poa: pool|sub-batch" $OUT/fill04.err | cut -c1-200 | head -60
VGA_TRACE=1 timeout -k 10 120 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/dflt.json 2> $OUT/dflt.err
grep -E "free list empty|poa: pool|dp \+ trace" $OUT/dflt.err | cut -c1-200 | head
cat $OUT/fill04.json $OUT/dflt.json | cut -c1-260
