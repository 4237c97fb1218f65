#!/bin/bash
# usage: prof_sweep.sh <reads> ; runs prof_small.py under several kernel configurations (diagnostics only)
N=${1:-2000}
run() { echo "== $*"; env "$@" VGA_TRACE=1 timeout -k 10 300 python3 tests/prof_small.py $N 10000 2 2>&1 | grep -a "launch \|poa_band_dp\|poa_value\|poa_traceback" | cut -c1-400 | tail -4; }
run VGA_POA_KERNEL=full
run VGA_X=1
run VGA_POA_NT=128
run VGA_POA_NT=192
run VGA_POA_NT=320
run VGA_POA_WINDOW=8192
run VGA_POA_WINDOW=8192 VGA_POA_NT=256
run VGA_POA_WINDOW=2048
run VGA_POA_WINDOW=2048 VGA_POA_NT=256
