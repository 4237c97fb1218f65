#!/bin/bash
# What the exit of `vgaligner map` costs (diagnostics, GPU box): with and without the alignment pass, with the allocator tuned or not
cd /tmp && export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
OUT=$REPO/gpurun_out/exit_${1:-x}
mkdir -p $OUT
run() {  # label, env..., -- extra flags
  lbl=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    VGA_TRACE=1 python3 $REPO/tests/prof_e2e_cli.py 10000 "$@" > $OUT/$lbl.json 2> $OUT/$lbl.err )
  python3 - $lbl $OUT/$lbl.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s wall %.2f s  reads/s %7.1f  before main %.3f  after done %.3f" % (sys.argv[1], j["map_s"], j["aligned_reads_per_s_end_to_end"], j["before_main_s"], j["after_done_s"]))
PY
}
run default --
run default_again --
run untuned_malloc VGA_TUNE_MALLOC=0 --
run pool_fill_04 VGA_POOL_FILL=0.4 --
run chains_only E2E_NO_ALIGN=1 --
