"""The GAF node-path agreement metric (experiments-snakemake/gafcompare.py:27-78 restated) and the truth GAF of
simulated reads."""
import os

from helpers import DATA, pkg


def test_jaccard_rules():
    g = pkg().gafcompare
    assert g.signed_path(">12<7>100") == [12, -7, 100]
    assert g.jaccard([1, 2, 3], [1, 2, 3]) == 1.0  # exact path
    # ranges [min, max): mine 10..20, ref 15..30 -> intersection range(15, 20) = 5, union range(10, 30) = 20
    assert g.jaccard([10, 20], [15, 30]) == 5 / 20
    assert g.jaccard([10, 12], [40, 50]) == 0.0  # disjoint: empty intersection range
    assert g.jaccard([5], [6]) == 0.0 and g.jaccard([5, 9], [5, 9, 7]) == 1.0  # same extremes, different lists
    assert g.jaccard([], [3, 4]) == 0.0  # placeholder record ('*'): scored 0 (the reference script raises)
    mine = "r1\t10\t0\t10\t+\t>1>2>3\t9\t0\t9\t0\t9\t255\tx\nr1\t10\t0\t10\t+\t>9\t9\t0\t9\t0\t9\t255\tx\nr3\t5\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n"
    ref = "r1\t10\t0\t10\t+\t>1>2>3\t9\t0\t9\t0\t9\t255\tt\nr2\t10\t0\t10\t+\t>4\t9\t0\t9\t0\t9\t255\tt\nr3\t5\t0\t5\t+\t>7>8\t5\t0\t5\t0\t5\t255\tt\n"
    r = g.compare(mine, ref)
    assert (r["matching_reads"], r["total_ref_reads"]) == (2, 3) and r["jaccard"] == [1.0, 0.0]  # first r1 record counts


def test_truth_gaf_of_simulated_reads():
    p = pkg()
    gfa = os.path.join(DATA, "DRB1-3123.gfa")
    reads = p.readsim.simulate_reads(gfa, 5, 400, 0.0, 0.0, 0.0, seed=3)
    segs, paths = p.readsim.parse_gfa_paths(gfa)
    by_name = dict(paths)
    for ln, r in zip(p.readsim.truth_gaf(gfa, reads).splitlines(), reads):
        f = ln.split("\t")
        assert f[0] == r.name and len(f) == 13
        ids = [int(x) for x in f[5].replace(">", " ").split()]
        seq = "".join(segs[i] for i in ids)
        assert r.seq in seq  # error-free read: contained in the concatenation of its truth nodes ...
        assert len(seq) - len(r.seq) < len(segs[ids[0]]) + len(segs[ids[-1]])  # ... with no spare node at either end
        steps = [n for n, _ in by_name[r.path]]
        k = steps.index(ids[0])
        assert steps[k:k + len(ids)] == ids
