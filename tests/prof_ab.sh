# diagnostic: A/B of an environment switch over the bench workloads.  usage: VAR=VGA_POA_ARENAS VALS="0 100000" bash tests/prof_ab.sh
cd $GRAFT_REPO_ROOT
for w in ${WORKLOADS:-config3 config4 config5}; do for v in $VALS; do
echo "== $w $VAR=$v"
env $VAR=$v timeout -k 10 300 python bench.py --workload $w --cpu-sample 0 --steps ${STEPS:-2} --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_busy_ms_per_step'].get('poa_band_dp'), d['kernels_ms_per_step'].get('poa_total'))" || exit 1
done; done
