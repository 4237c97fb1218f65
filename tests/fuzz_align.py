"""one-off fuzz (not collected by pytest): GPU == oracle on many mid-size reads and random POA problems"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from helpers import DATA, pkg, upload_oracle_index
import test_gpu_parity as T
from oracle import oracle_py as o

o.build()
p = pkg()
ctx = p.Context(0)
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed_off = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # other reads on every run: python tests/fuzz_align.py 6000 1000
t0 = time.time()
for gfa, seeds in ((os.path.join(DATA, "DRB1-3123.gfa"), (101, 102, 103)),):
    ix = o.Index(o.Graph.from_gfa(gfa), 11)
    upload_oracle_index(ctx, ix)
    for sd in seeds:
        L = {101: 1200, 102: 2500, 103: 4000}[sd]
        reads = p.readsim.simulate_reads(gfa, n_reads // 3, L, 0.04, 0.05, 0.06, seed=sd + seed_off)  # indel-heavy: the band's edges move a lot
        T._check_align(o, ctx, ix, reads)
        print("reads of", L, "ok", round(time.time() - t0, 1), "s", flush=True)
if n_reads >= 600:
    import tempfile
    d = tempfile.mkdtemp()
    hla = os.path.join(d, "hla19.gfa"); p.readsim.config4_graph(DATA, hla)
    syn = os.path.join(d, "syn.gfa"); p.readsim.synth_pangenome(syn, 80000, seed=5)
    for gfa in (hla, syn, os.path.join(DATA, "DRB1-3123.gfa")):
        ix = o.Index(o.Graph.from_gfa(gfa), 11)
        upload_oracle_index(ctx, ix)
        T._check_align(o, ctx, ix, p.readsim.simulate_reads(gfa, 40, 10000, 0.03, 0.03, 0.04, seed=7 + seed_off))
        T._check_align(o, ctx, ix, p.readsim.simulate_reads(gfa, 200, 700, 0.08, 0.06, 0.06, seed=8 + seed_off))
        print(os.path.basename(gfa), "10 kbp + noisy short reads ok", round(time.time() - t0, 1), "s", flush=True)
rng = random.Random(77 + seed_off)
probs = [T._rand_problem(rng, rng.randint(5, 60), 12) for _ in range(400)]
T._check_poa(o, ctx, probs)
print("400 random POA problems ok", flush=True)

# chimeric / rearranged reads: a stretch of random bases, a large deletion or a large duplication inside a real read --
# chains that cover part of the read, long extensions, rows that span the whole query, alignments with negative scores
if n_reads >= 600:
    gfa = os.path.join(DATA, "DRB1-3123.gfa")
    ix = o.Index(o.Graph.from_gfa(gfa), 11)
    upload_oracle_index(ctx, ix)
    base = p.readsim.simulate_reads(gfa, n_reads // 6, 3000, 0.02, 0.02, 0.03, seed=55 + seed_off)
    odd = []
    for i, r in enumerate(base):
        s = r.seq
        a = rng.randint(100, len(s) - 1200)
        k = i % 4
        if k == 0: s = s[:a] + "".join(rng.choice("ACGT") for _ in range(rng.randint(300, 1500))) + s[a + 800:]
        elif k == 1: s = s[:a] + s[a + rng.randint(400, 1000):]
        elif k == 2: s = s[:a + 600] + s[a:a + 600] + s[a + 600:]
        else: s = "".join(rng.choice("ACGT") for _ in range(rng.randint(200, 900))) + s[a:]
        import dataclasses
        odd.append(dataclasses.replace(r, seq=s))
    T._check_align(o, ctx, ix, odd)
    print(len(odd), "chimeric / rearranged reads ok", round(time.time() - t0, 1), "s", flush=True)
