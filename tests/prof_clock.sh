# effective shader clock of the POA DP kernel: GRBM_GUI_ACTIVE (sum over 8 XCDs) / 8 / kernel duration, per dispatch
# usage (GPU box): bash tests/prof_clock.sh <tag> [VAR=value ...]
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
OUT=$REPO/gpurun_out/clock_$TAG
mkdir -p $OUT
timeout -k 10 500 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/a -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $OUT/bench_a.json 2> $OUT/a.err
python3 - <<PY
import csv, glob
dur = {}
for f in glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = dur.get(r["Dispatch_Id"])
        if d: print(r["Kernel_Name"][:28], "dispatch", r["Dispatch_Id"], "%.1f ms" % (d * 1e3), "clock %.3f GHz" % (float(r["Counter_Value"]) / 8 / d / 1e9))
PY
