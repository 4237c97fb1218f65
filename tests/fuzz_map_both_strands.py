"""one-off fuzz (not collected by pytest): only_forward = 0 mapping, GPU == oracle on reads of both strands"""
import os, sys, time, tempfile
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import DATA, pkg, upload_oracle_index, compare_map
from oracle import oracle_py as o
o.build()
p = pkg()
ctx = p.Context(0)
comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
rc = lambda s: "".join(comp[c] for c in reversed(s))
d = tempfile.mkdtemp()
hla = os.path.join(d, "hla19.gfa"); p.readsim.config4_graph(DATA, hla)
t0 = time.time()
for gfa, n, L in ((os.path.join(DATA, "DRB1-3123.gfa"), 300, 1500), (hla, 120, 3000), (os.path.join(DATA, "DRB1-3123.gfa"), 40, 10000)):
    ix = o.Index(o.Graph.from_gfa(gfa), 11)
    upload_oracle_index(ctx, ix)
    reads = p.readsim.simulate_reads(gfa, n, L, 0.03, 0.03, 0.04, seed=99)
    seqs = [r.seq for r in reads] + [rc(r.seq) for r in reads]
    mp = p.default_map_params(); mp.only_forward = 0
    mo = ctx.batch(seqs).map(mp)
    compare_map(o, ix, mo, seqs, only_forward=False)
    print(os.path.basename(gfa), n, L, "both strands ok", round(time.time() - t0, 1), flush=True)
