"""The committed golden vectors (tests/golden/*.json, written by tests/golden/make_golden.py) against the CPU oracle:
the inputs regenerate bit for bit from their seeds and the oracle still gives the frozen answers."""
import hashlib
import json
import os
import sys

import pytest

from helpers import pkg

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import make_golden as mg  # noqa: E402


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    return mg.golden_inputs(pkg(), str(tmp_path_factory.mktemp("golden")))


def test_golden_inputs_regenerate(cases):
    want = json.load(open(os.path.join(GOLDEN, "hot_path.json")))
    assert sorted(want) == sorted(cases)
    for name, (gfa, k, reads) in cases.items():
        assert hashlib.sha256("\n".join(s for _, s in reads).encode()).hexdigest() == want[name]["reads_sha256"], name


def test_oracle_reproduces_golden_hot_path(oracle, cases):
    want = json.load(open(os.path.join(GOLDEN, "hot_path.json")))
    for name, (gfa, k, reads) in cases.items():
        ix = oracle.Index(oracle.Graph.from_gfa(gfa), k)
        names, seqs = [r[0] for r in reads], [r[1] for r in reads]
        cg, ag, st = oracle.map_reads(ix, names, seqs)
        w = want[name]
        assert cg == w["chains_gaf"] and ag == w["alignments_gaf"], name
        assert (st["poa_rows"], st["poa_cells"]) == (w["poa_rows"], w["poa_cells"])
        assert [mg.oracle_map_record(oracle, ix, s) for s in seqs] == w["map"], name


def test_config1_golden_is_the_reference_placeholder_line():
    """0 anchors => one placeholder chain => this exact line in both GAF files"""
    w = json.load(open(os.path.join(GOLDEN, "hot_path.json")))["config1_test_gfa"]
    line = "seq0\t31\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n"
    assert w["chains_gaf"] == line and w["alignments_gaf"] == line


def test_oracle_reproduces_golden_poa(oracle):
    for i, w in enumerate(json.load(open(os.path.join(GOLDEN, "poa.json")))):
        r = oracle.poa_align(w["nodes"], [tuple(e) for e in w["edges"]], w["query"], None)
        got = {"ok": bool(r.ok), "best_score": r.best_score, "cigar": r.cigar, "cs": r.cs_string, "abpoa_nodes": list(r.abpoa_nodes),
               "graph_nodes": list(r.graph_nodes), "aln_start_offset": r.aln_start_offset, "aln_end_offset": r.aln_end_offset,
               "n_aligned_bases": r.n_aligned_bases, "n_rows": r.n_rows, "n_cells": r.n_cells}
        assert got == {k: w[k] for k in got}, f"problem {i}"
