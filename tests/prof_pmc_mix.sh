# instruction-mix counters of the POA DP kernel for one bench step
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
OUT=$REPO/gpurun_out/pmc_mix_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_SENDMSG SQ_ACTIVE_INST_MISC SQ_INSTS_VSKIPPED --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/a -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_a.json 2> $OUT/a.err
python3 - <<PY
import csv, glob, collections
tot = collections.Counter()
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(tot.items()): print("%-24s %.4e" % (k, v))
PY
