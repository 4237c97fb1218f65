cd $GRAFT_REPO_ROOT
for n in 5000 7000 9000 10000 12000 15000 20000; do
echo "== reads $n"
timeout -k 10 400 python bench.py --reads $n --cpu-sample 0 --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_busy_ms_per_step'].get('poa_band_dp'), d['kernels_ms_per_step'].get('poa_total'))" || exit 1
done
