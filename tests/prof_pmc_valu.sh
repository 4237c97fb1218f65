cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/pmc_valu
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/a -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_a.json 2> $OUT/a.err
echo done a
timeout -k 10 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/b -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_b.json 2> $OUT/b.err
echo done b
