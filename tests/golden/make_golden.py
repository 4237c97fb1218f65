#!/usr/bin/env python3
"""Generates tests/golden/*.json: expected outputs of the hot path on small seeded inputs.

The reference (Rust + abPOA) cannot be built or run in this environment, so the vectors are
produced by the CPU oracle (oracle/, a restatement of the reference algorithm that is itself pinned by the
reference's own known-answer tests, tests/test_oracle_reference_vectors.py).  They freeze today's answers: the CPU
suite checks that the oracle still reproduces them, the GPU suite checks that the HIP path reproduces them through the
C ABI -- so a change that moved both the oracle and the kernels the same way would still be caught.

    python tests/golden/make_golden.py        (rewrites the fixtures; inputs are regenerated from fixed seeds)
"""
import hashlib
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DATA = os.path.join(HERE, "data")


def golden_inputs(pkg, tmp_dir):
    """name -> (gfa path, k, [(read name, sequence)]): the seeded inputs, shared with the tests"""
    rs = pkg.readsim
    drb1 = os.path.join(DATA, "DRB1-3123.gfa")
    syn = os.path.join(tmp_dir, "syn20k.gfa")
    rs.synth_pangenome(syn, 20000, seed=79)
    hla = os.path.join(tmp_dir, "hla19.gfa")
    rs.config4_graph(DATA, hla)
    single = [ln.strip() for ln in open(os.path.join(DATA, "single-read-test.fa")) if ln.strip()]
    named = lambda reads: [(r.name, r.seq) for r in reads]
    return {
        "config1_test_gfa": (os.path.join(DATA, "test.gfa"), 11, [(single[0][1:], single[1])]),
        "drb1_600bp_ont": (drb1, 11, named(rs.simulate_reads(drb1, 4, 600, 0.03, 0.03, 0.04, seed=5))),
        "drb1_150bp": (drb1, 11, named(rs.simulate_reads(drb1, 6, 150, 0.01, 0.0, 0.0, seed=6))
                       + [("poly_a", "A" * 150), ("with_n", "ACGTN" * 30), ("short", "ACGTACG")]),
        "drb1_2500bp_ont": (drb1, 11, named(rs.simulate_reads(drb1, 2, 2500, 0.03, 0.03, 0.04, seed=7))),
        "hla19_1200bp_ont": (hla, 11, named(rs.simulate_reads(hla, 6, 1200, 0.03, 0.03, 0.04, seed=8))),
        "syn20k_1500bp_ont": (syn, 11, named(rs.simulate_reads(syn, 3, 1500, 0.03, 0.03, 0.04, seed=9))),
    }


def golden_poa_problems():
    """seeded create_align_safe(nodes, edges, query) problems: (nodes, edges, query)"""
    rng = random.Random(2024)
    out = []
    for t in range(12):
        n = rng.randint(1, 14)
        nodes = ["".join(rng.choice("ACGT") for _ in range(rng.randint(1, 9))) for _ in range(n)]
        edges = sorted({(a, b) for a in range(n) for b in range(a + 1, min(n, a + 4)) if rng.random() < 0.5})
        path, v = [], 0
        while True:
            path.append(v)
            nxt = [b for a, b in edges if a == v]
            if not nxt:
                break
            v = rng.choice(nxt)
        q = list("".join(nodes[v] for v in path))
        for i in range(len(q)):
            x = rng.random()
            if x < 0.06:
                q[i] = rng.choice("ACGT")
            elif x < 0.10:
                q[i] += rng.choice("ACGT") * rng.randint(1, 3)
            elif x < 0.14:
                q[i] = ""
        out.append((nodes, [list(e) for e in edges], "".join(q) or "A"))
    out.append((["ACGT"], [], "ACGT"))
    out.append((["AC", "GT", "NN"], [[0, 1], [1, 2]], "ACNGTNN"))
    return out


def f64_hex(x):
    import struct
    return struct.pack("<d", x).hex()


def oracle_map_record(o, ix, seq):
    ref = o.chain_anchors(ix, seq, 50, 1000, 3)
    sa = ref.sorted_anchors
    pos_of = {a.id: i for i, a in enumerate(sa)}
    blob = b"".join(("%d,%d,%d,%d,%s,%s;" % (a.id, a.query_begin, a.target_begin[1], a.target_end[1], f64_hex(a.max_chain_score),
                                             a.best_predecessor_id)).encode() for a in sa)
    return {"n_anchors": len(sa), "anchors_sha256": hashlib.sha256(blob).hexdigest(), "curr_max": f64_hex(ref.curr_max),
            "first_anchors": [[a.id, a.query_begin, a.target_begin[1], a.target_end[1], f64_hex(a.max_chain_score), a.best_predecessor_id]
                              for a in sa[:3]],
            "chains": [[bool(ph), [pos_of[a.id] for a in ch]] for ph, ch in zip(ref.is_placeholder, ref.chains)]}


def main():
    import tempfile

    import __graft_entry__ as ge
    from oracle import oracle_py as o

    o.build()
    pkg = ge.load_package()
    with tempfile.TemporaryDirectory() as tmp:
        cases = {}
        for name, (gfa, k, reads) in golden_inputs(pkg, tmp).items():
            ix = o.Index(o.Graph.from_gfa(gfa), k)
            names, seqs = [r[0] for r in reads], [r[1] for r in reads]
            cg, ag, st = o.map_reads(ix, names, seqs)
            cases[name] = {"k": k, "reads_sha256": hashlib.sha256("\n".join(seqs).encode()).hexdigest(),
                           "map": [oracle_map_record(o, ix, s) for s in seqs], "chains_gaf": cg, "alignments_gaf": ag,
                           "poa_rows": st["poa_rows"], "poa_cells": st["poa_cells"]}
        json.dump(cases, open(os.path.join(HERE, "hot_path.json"), "w"), indent=1, sort_keys=True)
        poa = []
        for nodes, edges, q in golden_poa_problems():
            r = o.poa_align(nodes, [tuple(e) for e in edges], q, None)
            poa.append({"nodes": nodes, "edges": edges, "query": q, "ok": bool(r.ok), "best_score": r.best_score, "cigar": r.cigar,
                        "cs": r.cs_string, "abpoa_nodes": list(r.abpoa_nodes), "graph_nodes": list(r.graph_nodes),
                        "aln_start_offset": r.aln_start_offset, "aln_end_offset": r.aln_end_offset,
                        "n_aligned_bases": r.n_aligned_bases, "n_rows": r.n_rows, "n_cells": r.n_cells})
        json.dump(poa, open(os.path.join(HERE, "poa.json"), "w"), indent=1, sort_keys=True)
    print("wrote hot_path.json (%d cases) and poa.json (%d problems)" % (len(cases), len(poa)))


if __name__ == "__main__":
    main()
