cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/exit_${1:-x}
mkdir -p $OUT
for v in a b; do
VGA_TRACE=1 timeout -k 10 120 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/dflt_$v.json 2> $OUT/dflt_$v.err
done
grep -E "vgh-trace|align: |poa: pool|dp \+ trace" $OUT/dflt_b.err | cut -c1-250
cut -c1-250 $OUT/dflt_a.json $OUT/dflt_b.json
VGA_POOL_FILL=0.4 VGA_TRACE=1 timeout -k 10 120 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/fill04.json 2> $OUT/fill04.err
grep -E "free list empty|dp \+ trace|chunk pool" $OUT/fill04.err | cut -c1-200 | head
cut -c1-200 $OUT/fill04.json
cd $REPO && python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-300 $OUT/bench.json
