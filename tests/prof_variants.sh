# the GPU parity suite under the diagnostic configurations of the POA engine (each must pass)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | cut -c1-200; }
run VGA_POA_ARENAS=0
run VGA_POA_ARENAS=300
run VGA_POA_TB=wave
run VGA_POA_TB=lane
run VGA_POOL_BYTES=300000000
run VGA_POA_H16=1
run VGA_POA_KERNEL=full
run VGA_POA_WINDOW=256
run VGA_POA_KERNEL=512
run VGA_POA_KERNEL=generic
run VGA_POA_SLOTS=1
run VGA_POA_SUB=7
run VGA_POA_KERNEL=unpacked
run VGA_MAP_CHAIN=old
