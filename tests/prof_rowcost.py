"""Synthetic POA problems that isolate the per-row and per-step instruction cost of the DP kernel (diagnostics, GPU box):
unbanded alignment of a random query of Q bases to a linear graph of N bases makes every row exactly Q + 1 columns wide.
    python3 tests/prof_rowcost.py Q [node_len] [N] [problems]      (run under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU)
Prints rows, cells and kernel time; tests/prof_rowcost.sh turns several Q into per-row / per-step instruction counts."""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

Q = int(sys.argv[1])
node_len = int(sys.argv[2]) if len(sys.argv) > 2 else 0
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1500
nprob = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
p = ge.load_package()
rng = random.Random(7)
g = "".join(rng.choice("ACGT") for _ in range(N))
nodes = [g] if node_len <= 0 else [g[i:i + node_len] for i in range(0, N, node_len)]
edges = [(i, i + 1) for i in range(len(nodes) - 1)]
probs = []
for _ in range(nprob):
    q = "".join(rng.choice("ACGT") for _ in range(Q))
    probs.append((nodes, edges, q))
ctx = p.Context(0)
pp = p.default_poa_params()
pp.wb = -1
t0 = time.time()
out = ctx.poa_batch(probs, pp)
dt = time.time() - t0
cells = int(out.n_cells.sum())
rows = int(out.n_rows.sum())
kt = {k["name"]: k for k in ctx.kernel_times()}
print("ROWCOST Q %d node_len %d rows %d cells %d wall %.3f s dp_ms %.1f" % (Q, node_len, rows, cells, dt, kt.get("poa_band_dp", {}).get("busy_ms", 0.0)))
