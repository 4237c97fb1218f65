"""The POA oracle (oracle/og_poa.c) against an independent textbook formulation.

abPOA itself is not available (parity with it is unpinned), so what CAN be pinned is that the oracle's
score is the true optimum of the model it claims to implement: global alignment of the query to a source-to-sink path
of the graph, match / mismatch, and the convex gap cost g(k) = min(o1 + k e1, o2 + k e2) per gap run.  The check
below shares nothing with the oracle's formulation (no affine E / F states, no graph recurrence, no band): it
enumerates every source-to-sink path of a small graph and aligns the query to each path's sequence with the
general-gap dynamic programme of Waterman, Smith and Beyer (every gap length tried explicitly)."""
import random

import numpy as np
import pytest


def wsb_global(a: str, b: str, m, x, g):
    """max score of a global alignment of a and b, gap run of length k costs g(k); O(|a| |b| (|a| + |b|))"""
    la, lb = len(a), len(b)
    D = np.full((la + 1, lb + 1), -10**9, dtype=np.int64)
    D[0, 0] = 0
    gk = np.array([0] + [g(k) for k in range(1, max(la, lb) + 1)], dtype=np.int64)
    for i in range(la + 1):
        for j in range(lb + 1):
            if i == 0 and j == 0:
                continue
            best = -10**9
            if i and j:
                best = D[i - 1, j - 1] + (m if a[i - 1] == b[j - 1] else -x)
            if i:
                best = max(best, int((D[:i, j] - gk[i:0:-1]).max()))
            if j:
                best = max(best, int((D[i, :j] - gk[j:0:-1]).max()))
            D[i, j] = best
    return int(D[la, lb])


def all_paths(n, edges):
    out = {}
    for s, d in edges:
        out.setdefault(s, []).append(d)
    has_in = {d for _, d in edges}
    res = []

    def walk(v, acc):
        if v not in out:
            res.append(acc + [v])
            return
        for w in out[v]:
            walk(w, acc + [v])
    for v in range(n):
        if v not in has_in:
            walk(v, [])
    return res


def rand_graph(rng, n):
    nodes = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4))) for _ in range(n)]
    edges = sorted({(a, b) for a in range(n) for b in range(a + 1, min(n, a + 3)) if rng.random() < 0.6})
    return nodes, edges


@pytest.mark.parametrize("pen", [(2, 4, 4, 2, 24, 1), (1, 3, 2, 2, 10, 1), (2, 2, 1, 3, 6, 1)])
def test_oracle_score_is_the_optimum_of_the_convex_gap_model(oracle, pen):
    m, x, o1, e1, o2, e2 = pen
    g = lambda k: min(o1 + k * e1, o2 + k * e2)
    rng = random.Random(hash(pen) & 0xffff)
    p = oracle.default_poa_params()
    p.match, p.mismatch, p.gap_open1, p.gap_ext1, p.gap_open2, p.gap_ext2 = pen
    p.wb = -1  # no band: the oracle must find the global optimum
    checked = 0
    for _ in range(60):
        nodes, edges = rand_graph(rng, rng.randint(1, 7))
        paths = all_paths(len(nodes), edges)
        q = "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 14)))
        if rng.random() < 0.5:  # a noisy copy of one path: long matches, realistic gaps
            src = "".join(nodes[v] for v in rng.choice(paths))
            q = "".join(c if rng.random() > 0.2 else rng.choice("ACGT") for c in src)[: 16] or "A"
            if rng.random() < 0.5 and len(q) > 6:
                cut = rng.randint(1, len(q) - 4)
                q = q[:cut] + q[cut + rng.randint(1, 3):]
        want = max(wsb_global("".join(nodes[v] for v in path), q, m, x, g) for path in paths)
        r = oracle.poa_align(nodes, edges, q, p)
        assert r.ok and r.best_score == want, (nodes, edges, q, r.best_score, want, r.cigar)
        checked += 1
    assert checked == 60


def test_banded_oracle_equals_unbanded_on_well_behaved_reads(oracle):
    """the adaptive band (b = 10, f = 0.01) is wide enough for a low-error read: same optimum as without a band"""
    rng = random.Random(4)
    for _ in range(20):
        nodes = ["".join(rng.choice("ACGT") for _ in range(rng.randint(2, 9))) for _ in range(12)]
        edges = [(i, i + 1) for i in range(11)] + [(i, i + 2) for i in range(0, 10, 3)]
        path, v = [], 0
        while v < 12:
            path.append(v)
            v += 2 if (v % 3 == 0 and v + 2 < 12 and rng.random() < 0.5) else 1
        q = "".join(c if rng.random() > 0.03 else rng.choice("ACGT") for c in "".join(nodes[v] for v in path))
        pb, pu = oracle.default_poa_params(), oracle.default_poa_params()
        pu.wb = -1
        assert oracle.poa_align(nodes, edges, q, pb).best_score == oracle.poa_align(nodes, edges, q, pu).best_score
