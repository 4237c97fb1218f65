"""The HIP path against the committed golden vectors, through the C ABI and the product's own C++ host (index
builder, map_reads, GAF writer).  Nothing here touches the oracle: the expected values are data."""
import hashlib
import json
import os
import struct
import sys

import numpy as np
import pytest

from helpers import pkg

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden as mg  # noqa: E402


def _hex(x):
    return struct.pack("<d", float(x)).hex()


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    return mg.golden_inputs(pkg(), str(tmp_path_factory.mktemp("golden")))


def test_hip_hot_path_reproduces_golden(cases):
    p = pkg()
    want = json.load(open(os.path.join(GOLDEN, "hot_path.json")))
    ctx = p.Context(0)
    for name, (gfa, k, reads) in cases.items():
        w = want[name]
        hi = p.HostIndex.build_from_gfa(gfa, k)
        hi.upload(ctx)
        names, seqs = [r[0] for r in reads], [r[1] for r in reads]
        cg, ag, n_al = hi.map_reads(ctx, names, seqs, also_align=True)
        assert cg == w["chains_gaf"], name + ": chains GAF"
        assert ag == w["alignments_gaf"], name + ": alignments GAF"
        b = ctx.batch(seqs)
        mo = b.map()
        for r, g in enumerate(w["map"]):
            a0, a1 = int(mo.anchor_off[r]), int(mo.anchor_off[r + 1])
            assert a1 - a0 == g["n_anchors"], f"{name} read {r}: anchors"
            rows = [(int(mo.anchor_id[i]), int(mo.query_begin[i]), int(mo.target_begin[i]), int(mo.target_end[i]),
                     _hex(mo.max_chain_score[i]), int(mo.best_pred_id[i])) for i in range(a0, a1)]
            blob = b"".join(("%d,%d,%d,%d,%s,%s;" % t).encode() for t in rows)
            assert [list(t) for t in rows[:3]] == g["first_anchors"], f"{name} read {r}: first anchors"
            assert hashlib.sha256(blob).hexdigest() == g["anchors_sha256"], f"{name} read {r}: sorted anchors / f(i) / predecessors"
            assert _hex(mo.curr_max[r]) == g["curr_max"]
            assert [[bool(ph), list(ch)] for ph, ch in mo.chains_of(r)] == g["chains"], f"{name} read {r}: chains"
        al = b.align(mo)
        assert (al.poa_rows, al.poa_cells) == (w["poa_rows"], w["poa_cells"]), name
    ctx.close()


def test_hip_poa_reproduces_golden():
    p = pkg()
    want = json.load(open(os.path.join(GOLDEN, "poa.json")))
    ctx = p.Context(0)
    out = ctx.poa_batch([(w["nodes"], [tuple(e) for e in w["edges"]], w["query"]) for w in want])
    for i, w in enumerate(want):
        assert bool(out.ok[i]) == w["ok"]
        s, e = int(out.path_off[i]), int(out.path_off[i + 1])
        got = {"best_score": int(out.best_score[i]), "cigar": out.cigar[i], "cs": out.cs[i], "abpoa_nodes": out.abpoa_nodes[s:e].tolist(),
               "graph_nodes": out.graph_nodes[s:e].tolist(), "aln_start_offset": int(out.aln_start_offset[i]),
               "aln_end_offset": int(out.aln_end_offset[i]), "n_aligned_bases": int(out.n_aligned_bases[i]),
               "n_rows": int(out.n_rows[i]), "n_cells": int(out.n_cells[i])}
        assert got == {k: w[k] for k in got}, f"problem {i}"
    ctx.close()
