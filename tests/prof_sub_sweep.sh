# diagnostic: throughput against the sub-batch size cap (VGA_POA_SUB), per workload
cd $GRAFT_REPO_ROOT
for w in ${WORKLOADS:-config5 config3}; do for sub in ${SUBS:-1024 2048 4096 100000}; do
echo "== $w sub $sub"
VGA_POA_SUB=$sub timeout -k 10 300 python bench.py --workload $w --cpu-sample 0 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_busy_ms_per_step'].get('poa_band_dp'), d['kernels_ms_per_step'].get('poa_total'))" || exit 1
done; done
