#!/bin/bash
# per-row / per-step instruction cost of the POA DP kernel from synthetic full-width rows (tests/prof_rowcost.py)
# usage (GPU box): bash tests/prof_rowcost.sh <tag> [ENV=VAL ...]        env VGA_LIB selects the build
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
TAG=${1:-x}; shift
for kv in "$@"; do export "$kv"; done
OUT=$REPO/gpurun_out/rowcost_$TAG
mkdir -p $OUT
for spec in "1023 0" "2047 0" "4095 0" "2047 1" "2047 4" "1500 0" "300 0"; do
  set -- $spec
  lbl=q$1_n$2
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --kernel-include-regex "k_poa_dp" -d $OUT/$lbl -o pmc --output-format csv -- python3 $REPO/tests/prof_rowcost.py $1 $2 > $OUT/$lbl.txt 2> $OUT/$lbl.err || { echo "$lbl failed"; tail -3 $OUT/$lbl.err; continue; }
  python3 - $OUT/$lbl $OUT/$lbl.txt <<'PY'
import csv, glob, sys, collections
d, t = sys.argv[1:3]
line = [l for l in open(t) if l.startswith("ROWCOST")][-1].split()
Q, rows, cells = int(line[2]), int(line[6]), int(line[8])
tot = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
print("%s node_len %s: per row: VALU %.0f SALU %.0f LDS %.0f | per 64 cells: VALU %.1f SALU %.1f  (rows %d, %d columns, dp %s ms)" % (
    "Q=%d" % Q, line[4], tot["SQ_INSTS_VALU"] / rows, tot["SQ_INSTS_SALU"] / rows, tot["SQ_INSTS_LDS"] / rows,
    tot["SQ_INSTS_VALU"] * 64 / cells, tot["SQ_INSTS_SALU"] * 64 / cells, rows, Q + 1, line[-1]))
PY
done
