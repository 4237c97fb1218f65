cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/exit_${1:-x}
mkdir -p $OUT
for v in a; do
VGA_TRACE=1 timeout -k 10 120 python3 $REPO/tests/prof_e2e_cli.py 10000 > $OUT/dflt_$v.json 2> $OUT/dflt_$v.err
grep -E "vgh-trace" $OUT/dflt_$v.err | cut -c1-150
cut -c1-250 $OUT/dflt_$v.json
done
