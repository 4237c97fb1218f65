cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
for v in "X=1" "VGA_POA_ARENAS=0"; do
  for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA_WRREQ_STALL_sum" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" "TCC_EA_WRREQ_sum TCC_EA_RDREQ_sum TCC_WRITEBACK_sum"; do
    out=$REPO/gpurun_out/pmcmem_${v//=/_}_$(echo $set | tr ' ' '_' | cut -c1-20)
    env $v timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace --kernel-include-regex "k_poa_dp" -d $out -o pmc --output-format csv -- python3 $REPO/bench.py --workload config5 --reads 10000 --steps 1 --warmup 0 --cpu-sample 0 > /dev/null 2> $out.err || { echo "fail $v $set"; tail -2 $out.err; continue; }
    python3 - "$v" $out <<'PY'
import csv, glob, sys, collections
v, d = sys.argv[1:3]
tot = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
print(v, {k: "%.3e" % x for k, x in tot.items()})
PY
  done
done
