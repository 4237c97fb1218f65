# same-box A/B of two builds of libvga_hip.so (build_ab/libA.so = baseline, build_ab/libB.so = candidate)
cd $GRAFT_REPO_ROOT
cp rs-vgaligner_amd/libvga_hip.so /tmp/lib_keep.so
for v in ${ORDER:-A B A B A B}; do
cp build_ab/lib$v.so rs-vgaligner_amd/libvga_hip.so
echo "== $v"
timeout -k 10 300 python bench.py --workload ${WORKLOAD:-config3} --cpu-sample 0 --steps ${STEPS:-3} --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernels_busy_ms_per_step'].get('poa_band_dp'), d['kernels_busy_ms_per_step'].get('poa_total'))" || exit 1
done
cp /tmp/lib_keep.so rs-vgaligner_amd/libvga_hip.so
