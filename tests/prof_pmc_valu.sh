# instruction counters of the POA DP kernel for one bench step (10 000 reads of config 3); optional env is passed through
# usage (GPU box): bash tests/prof_pmc_valu.sh <tag> [VAR=value ...]
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
OUT=$REPO/gpurun_out/pmc_valu_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH SQ_THREAD_CYCLES_VALU --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/a -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_a.json 2> $OUT/a.err
echo done a
timeout -k 10 500 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/b -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_b.json 2> $OUT/b.err
echo done b
python3 - <<PY
import csv, glob, collections
for sub in ("a", "b"):
    tot = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in sorted(tot.items()): print("%-24s %.4e" % (k, v))
PY
