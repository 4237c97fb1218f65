"""The C++ host (Index::build, FASTA/FASTQ reader, index file) checked against the oracle and the
reference's fixtures.  Two independently written builders (C oracle, C++ product) must agree bit for bit."""
import os

import numpy as np
import pytest

from helpers import DATA, oracle_index_arrays, pkg


@pytest.mark.parametrize("gfa,k", [("test.gfa", 11), ("test.gfa", 3), ("DRB1-3123.gfa", 11), ("DRB1-3123.gfa", 7)])
def test_host_index_equals_oracle_index(oracle, gfa, k):
    path = os.path.join(DATA, gfa)
    hi = pkg().HostIndex.build_from_gfa(path, k)
    got = hi.arrays()
    want = oracle_index_arrays(oracle.Index(oracle.Graph.from_gfa(path), k))
    assert got["k"] == want["k"] and got["seq_fwd"] == want["seq_fwd"]
    for key in ("node_seq_idx", "node_edge_idx", "node_edges_to", "edges", "kmer_starts"):
        assert np.array_equal(got[key], np.asarray(want[key], dtype=np.uint64)), key
    assert got["kmer_keys"] == want["kmer_keys"]
    for f in ("start", "end", "start_orient", "end_orient"):
        assert np.array_equal(got["kmer_pos_table"][f], want["kmer_pos_table"][f]), f


def _same_index(oracle, path, k):
    got = pkg().HostIndex.build_from_gfa(path, k).arrays()
    want = oracle_index_arrays(oracle.Index(oracle.Graph.from_gfa(path), k))
    assert got["k"] == want["k"] and got["seq_fwd"] == want["seq_fwd"] and got["kmer_keys"] == want["kmer_keys"]
    for key in ("node_seq_idx", "node_edge_idx", "node_edges_to", "edges", "kmer_starts"):
        assert np.array_equal(got[key], np.asarray(want[key], dtype=np.uint64)), key
    for f in ("start", "end", "start_orient", "end_orient"):
        assert np.array_equal(got["kmer_pos_table"][f], want["kmer_pos_table"][f]), f


def test_host_index_equals_oracle_index_config4_and_config5_graphs(oracle, config4_gfa, config5_small_gfa):
    """the merged HLA graph (config #4) and the synthetic pangenome generator (config #5, at 60 kbp)"""
    _same_index(oracle, config4_gfa, 11)
    _same_index(oracle, config5_small_gfa, 11)


def test_graph_generators_are_deterministic_and_topological(tmp_path, config4_gfa):
    rs = pkg().readsim
    a, b = str(tmp_path / "a.gfa"), str(tmp_path / "b.gfa")
    na = rs.synth_pangenome(a, 20000, seed=5)
    assert na == rs.synth_pangenome(b, 20000, seed=5) and open(a).read() == open(b).read()
    for path in (a, config4_gfa):
        segs, paths = rs.parse_gfa_paths(path)
        assert sorted(segs) == list(range(1, len(segs) + 1))
        edges, mixed, back = set(), 0, 0
        for ln in open(path):
            if ln.startswith("L\t"):
                f = ln.split("\t")
                if f[2] != f[4]:
                    mixed += 1
                    continue
                assert f[2] == "+"  # reversed pairs were turned around
                back += int(f[3]) <= int(f[1])
                edges.add((int(f[1]), int(f[3])))
        # the synthetic pangenome is a DAG in id order; of the merged HLA loci three stay cyclic after sorting and two are not
        # strand-consistent (readsim.HLA_CONFIG4): 62 back edges incl. 30 self loops, 17 mixed links, out of 33 177
        assert (mixed, back) == ((0, 0) if path == a else (17, 62)), (mixed, back)
        for name, steps in paths:
            if any(rev for _, rev in steps):
                continue  # paths with reverse steps are skipped by the read sampler
            assert all((x[0], y[0]) in edges for x, y in zip(steps, steps[1:])), name
    segs, paths = rs.parse_gfa_paths(a)
    assert len(paths) == 16 and 19000 < sum(len(s) for s in segs.values()) < 21500


def test_host_index_small_graph_vectors(tmp_path, oracle):
    """src/index.rs:761-824 (linearisation, NodeRefs) and 1109-1129 (ACT -> F0..F3) through the C++ builder"""
    gfa = tmp_path / "simple.gfa"
    gfa.write_text("H\tVN:Z:1.0\nS\t1\tA\nS\t2\tCT\nS\t3\tGA\nS\t4\tGCA\nL\t1\t+\t2\t+\t0M\nL\t1\t+\t3\t+\t0M\n"
                   "L\t2\t+\t4\t+\t0M\nL\t3\t+\t4\t+\t0M\n")
    a = pkg().HostIndex.build_from_gfa(str(gfa), 3).arrays()
    assert a["seq_fwd"] == b"ACTGAGCA"
    assert a["node_seq_idx"].tolist() == [0, 1, 3, 5, 8]
    assert a["node_edge_idx"].tolist() == [0, 2, 4, 6, 8]
    assert a["node_edges_to"].tolist() == [0, 1, 1, 2, 0]
    keys = [a["kmer_keys"][i:i + 3] for i in range(0, len(a["kmer_keys"]), 3)]
    s = int(a["kmer_starts"][keys.index(b"ACT")])
    t = a["kmer_pos_table"][s]
    assert (int(t["start_orient"]), int(t["start"]), int(t["end_orient"]), int(t["end"])) == (0, 0, 0, 3)
    assert int(a["kmer_pos_table"][s + 1]["start"]) == 2**64 - 1  # delimiter


def test_host_index_store_load_roundtrip(tmp_path):
    p = pkg()
    hi = p.HostIndex.build_from_gfa(os.path.join(DATA, "test.gfa"), 11)
    f = str(tmp_path / "t.idx")
    hi.store(f)
    a, b = hi.arrays(), p.HostIndex.load(f).arrays()
    for key in a:
        if key == "kmer_pos_table":
            for fld in ("start", "end", "start_orient", "end_orient"):
                assert np.array_equal(a[key][fld], b[key][fld])
        elif isinstance(a[key], np.ndarray):
            assert np.array_equal(a[key], b[key])
        else:
            assert a[key] == b[key]


def test_host_errors_mirror_reference_panics(tmp_path):
    p = pkg()
    with pytest.raises(p.hostlib.HostError):  # kmer.rs:828 unwrap() on an empty k-mer list
        p.HostIndex.build_from_gfa(os.path.join(DATA, "test.gfa"), 100)
    bad = tmp_path / "ids.gfa"
    bad.write_text("S\t2\tACGT\nS\t5\tAC\nL\t2\t+\t5\t+\t0M\n")
    with pytest.raises(p.hostlib.HostError):  # ids must be 1..n
        p.HostIndex.build_from_gfa(str(bad), 3)
    with pytest.raises(p.hostlib.HostError):
        p.HostIndex.load(str(bad))
    with pytest.raises(p.hostlib.HostError):  # io.rs:86 "Unrecognized file type"
        p.hostlib.read_seqs_from_file(str(bad))


def test_host_reader_matches_reference_fixtures():
    """src/io.rs:267-308"""
    r = pkg().hostlib.read_seqs_from_file
    assert r(os.path.join(DATA, "single-read-test.fa")) == [("seq0", "AAAAACGTTAAATTTGGCATCGTAGCAAAAA")]
    assert r(os.path.join(DATA, "multiple-read-test.fa")) == [("seq0", "AAAAACGTTAAATTTGGCATCGTAGCAAAAA"),
                                                                ("seq1", "TTTCGTTAAATTTGGCATCGTAGCTTT")]
    assert len(r(os.path.join(DATA, "test.fq"))) == 1


def test_host_reader_multiline_fasta(tmp_path):
    """src/io.rs:100-122: every sequence line is a read; later lines get the name suffixed by a counter"""
    f = tmp_path / "m.fa"
    f.write_text(">a b\nACGT\nTTTT\n\nGG\n>c\nAA\n")
    assert pkg().hostlib.read_seqs_from_file(str(f)) == [("a b", "ACGT"), ("a b1", "TTTT"), ("a b2", "GG"), ("c", "AA")]


def test_validation_records(tmp_path):
    """src/validate.rs: parse_nodes_from_path_matching (tests at validate.rs:227-240: ">1<2>3" -> [1, 2, 3],
    ">10<20" -> [10, 20], "*" -> []) and ValidationRecord::to_string, through the C++ host"""
    import json

    p = pkg()
    gfa = tmp_path / "g.gfa"
    gfa.write_text("H\tVN:Z:1.0\n" + "".join("S\t%d\t%s\n" % (i + 1, s) for i, s in enumerate(["AAC", "ACG", "T"] + ["G"] * 17))
                   + "".join("L\t%d\t+\t%d\t+\t0M\n" % (i, i + 1) for i in range(1, 20)))
    hi = p.HostIndex.build_from_gfa(str(gfa), 3)
    rec = lambda path, notes="as:i:-30 cs:Z::3,cg:Z:3M": "r\t3\t0\t3\t+\t%s\t3\t0\t3\t0\t3\t255\t%s\n" % (path, notes)
    got = hi.validation_records(rec(">1<2>3"), ["r", "r"], ["AAC", "TTT"])
    assert got == 'r\ncg:Z:3M\nAAC\n[1, 2, 3]\n["AAC", "ACG", "T"]\n\n'  # the first read named r; '<' parses to the bare id
    assert hi.validation_records(rec(">10<20"), ["r"], ["AAC"]).splitlines()[3] == "[10, 20]"
    # last id < first id: the sequences are reverse-complemented (validate.rs:50-53,113-124)
    assert hi.validation_records(rec("<2<1"), ["r"], ["AAC"]).splitlines()[3:5] == ["[2, 1]", '["CGT", "GTT"]']
    assert hi.validation_records("r\t3\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n", ["r"], ["AAC"]) == "r\nNOT ALIGNED\nAAC\n[]\n[]\n\n"
    with pytest.raises(p.hostlib.HostError):  # the reference unwraps the read lookup
        hi.validation_records(rec(">1"), ["x"], ["AAC"])
    # the golden DRB1 alignments: one record per line, the node sequences spell the path
    want = json.load(open(os.path.join(os.path.dirname(DATA), "hot_path.json")))["drb1_600bp_ont"]
    drb1 = p.HostIndex.build_from_gfa(os.path.join(DATA, "DRB1-3123.gfa"), 11)
    arr = drb1.arrays()
    lines = want["alignments_gaf"].splitlines()
    names = [ln.split("\t")[0] for ln in lines]
    out = drb1.validation_records(want["alignments_gaf"], names, ["ACGT"] * len(names)).split("\n\n")
    assert len(out) == len(lines) + 1 and out[-1] == ""
    for ln, r in zip(lines, out):
        f, v = ln.split("\t"), r.split("\n")
        ids = [int(x) for x in f[5].replace(">", " ").split()]
        st = arr["node_seq_idx"]
        assert v[0] == f[0] and v[1] == "cg:Z:" + f[12].split(",cg:Z:")[1] and v[2] == "ACGT"
        assert v[3] == "[" + ", ".join(map(str, ids)) + "]"
        assert v[4] == "[" + ", ".join('"%s"' % arr["seq_fwd"][int(st[i - 1]):int(st[i])].decode() for i in ids) + "]"


def test_toposort_gfa_keeps_every_path_sequence_and_orders_the_acyclic_loci(tmp_path):
    """readsim.toposort_gfa, the stand-in for `odgi sort` (reference README.md:24-28), on the 20 HLA-zoo graphs: the sequence
    every path spells is unchanged, ids are 1..n, links between reversed handles are turned around, and the loci that have
    a topological order get one (no back edge)."""
    rs = pkg().readsim
    out = str(tmp_path / "s.gfa")
    cyclic = {}
    for name in rs.HLA_ALL:
        src = os.path.join(DATA, name + ".gfa")
        st = rs.toposort_gfa(src, out)
        s0, p0 = rs.parse_gfa_paths(src)
        s1, p1 = rs.parse_gfa_paths(out)
        assert sorted(s1) == list(range(1, len(s0) + 1)) and st["nodes"] == len(s0)
        assert [n for n, _ in p0] == [n for n, _ in p1]
        for (_, a), (_, b) in zip(p0, p1):
            assert rs.path_sequence(s0, a) == rs.path_sequence(s1, b)
        assert not any(ln.startswith("L\t") and ln.split("\t")[2] == "-" and ln.split("\t")[4] == "-" for ln in open(out))
        if st["back_edges"] or st["mixed_links"]:
            cyclic[name.split("/")[-1]] = (st["back_edges"], st["mixed_links"])
        # sorting a sorted graph changes nothing (where the orientation is determined at all: strand-consistent components)
        if st["mixed_links"] == 0:
            again = str(tmp_path / "s2.gfa")
            st2 = rs.toposort_gfa(out, again)
            assert st2["flipped"] == 0 and open(again).read() == open(out).read(), name
    assert cyclic == {"5-B3106": (22, 0), "7-MICB-4277": (94, 0), "8-C3107": (31, 0), "10-F-3134": (1, 0), "15-H-3136-spoa": (1, 0),
                      "16-DQB1-3119-spoa": (7, 0), "17-DRB1-3123-smooth": (0, 15), "20-C3107-smooth": (0, 2)}, cyclic


def test_alignment_record_writer(oracle):
    """vgh::gaf_from_alignment (host/vgh_map.cpp) writes GAFAlignment::to_string (src/align.rs:746-760) of generate_alignment's
    record (src/align.rs:1145-1167) through raw pointers into a buffer sized from an upper bound: the text must be the plain
    formatting, for short and long ids, either orientation, an empty path, a prefix already in the buffer, and a placeholder."""
    import random
    H = pkg().hostlib
    rng = random.Random(5)

    def expect(name, L, handles, plen, ps, pe, blk, cs, cg):
        path = "".join(("<" if h & 1 else ">") + str(h >> 1) for h in handles)
        return "\t".join([name, str(L), "0", str(L), "+", path, str(plen), str(ps), str(pe), "0", str(blk), "255", "as:i:-30 " + cs + ",cg:Z:" + cg]) + "\n"

    cases = [
        ("r1", 12, [2, 4, 7], 3, 0, 5, 12, "cs:Z::12", "12M"),
        ("read with spaces", 1, [(2 ** 63 - 1) << 1 | 1], 1, 0, 1, 1, "cs:Z:*ac", "1M"),           # a 19-digit id, reverse
        ("e", 0, [], 0, 0, 0, 0, "cs:Z:", ""),                                                        # nothing aligned to anything
        ("big", 10 ** 7, [rng.randrange(1, 10 ** 9) << 1 | rng.randrange(2) for _ in range(5000)], 5000, 3, 9, 10 ** 7,
         "cs:Z:" + ":7*ag" * 4000, "7M1X" * 4000),
    ]
    for prefix in ("", "earlier record\n" * 3):
        for name, L, hs, plen, ps, pe, blk, cs, cg in cases:
            got = H.gaf_alignment_record(name, L, True, hs, plen, ps, pe, blk, cs, cg, prefix=prefix)
            assert got == prefix + expect(name, L, hs, plen, ps, pe, blk, cs, cg)
    # not aligned: the placeholder of src/align.rs:1012-1028
    assert H.gaf_alignment_record("nope", 77, False, [], 0, 0, 0, 0, "", "", prefix="x\n") == "x\nnope\t77\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n"
