# same-box A/B of environment settings: SETS="name1:VAR=V,VAR2=V name2:..." (name "base" = no variables)
cd $GRAFT_REPO_ROOT
for rep in 1 2 ${REPS:-}; do for set in $SETS; do
name=${set%%:*}; vars=${set#*:}; [ "$vars" = "$set" ] && vars=""
echo "== $name"
env $(echo $vars | tr ',' ' ') timeout -k 10 300 python bench.py --workload ${WORKLOAD:-config3} --cpu-sample 0 --steps ${STEPS:-3} --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_busy_ms_per_step'].get('poa_band_dp'), d['kernels_ms_per_step'].get('poa_total'))" || exit 1
done; done
