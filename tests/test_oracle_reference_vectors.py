"""Pins the CPU oracle against every known-answer vector the reference's own unit tests hold for
the map -> chain -> align path.  Each test names the reference test it
restates (paths relative to the reference checkout)."""
import math

import pytest

F, R = 0, 1


def simple_graph(o):
    """src/index.rs:654-678 / src/chain.rs:715-739: 1:A -> {2:CT, 3:GA} -> 4:GCA"""
    return o.Graph.from_nodes_edges([(1, "A"), (2, "CT"), (3, "GA"), (4, "GCA")], [(1, 2), (1, 3), (2, 4), (3, 4)])


def H(o, i, rev=False):
    return o.pack(i, rev)


# ---------------------------------------------------------------- src/index.rs
def test_forward_creation(oracle):
    """src/index.rs:761-824 test_forward_creation"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert ix.seq_length == 8
    assert ix.seq_fwd == "ACTGAGCA"
    assert ix.seq_bv() == [1, 1, 0, 1, 0, 1, 0, 0, 1]
    assert ix.node_ref() == [(0, 0, 0), (1, 2, 1), (3, 4, 1), (5, 6, 2), (8, 8, 0)]


def test_simple_path(oracle):
    """src/index.rs:843-890 test_simple_path"""
    g = oracle.Graph.from_nodes_edges([(1, "ACG"), (2, "TTT"), (3, "CA")], [(1, 2), (2, 3)])
    ix = oracle.Index(g, 3)
    assert ix.seq_length == 8 and ix.seq_fwd == "ACGTTTCA"
    nr = ix.node_ref()
    assert nr[1] == (3, 1, 1) and nr[2] == (6, 3, 1)
    assert g.generate_kmers_count(3) == 12


def test_kmers_graph_generation(oracle):
    """src/index.rs:827-840 test_kmers_graph_generation"""
    g = simple_graph(oracle)
    assert g.generate_kmers_count(3) == 14
    assert g.generate_kmers_count(6) == 4
    assert g.generate_kmers_count(100) == 0


def test_compare_sequential_parallel_graphkmer(oracle):
    """src/index.rs:1246-1258: generate_kmers == generate_kmers_parallel on N-free graphs"""
    g = simple_graph(oracle)
    assert oracle.Index(g, 3).n_graph_kmers == g.generate_kmers_count(3)
    g2 = oracle.Graph.from_nodes_edges([(1, "GAT"), (2, "T"), (3, "A"), (4, "CA")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    assert oracle.Index(g2, 3).n_graph_kmers == g2.generate_kmers_count(3)


def test_revcomp(oracle):
    """src/dna.rs:46-50 test_revcomp + seq_rev of the simple graph (src/index.rs:651-653 drawing)"""
    g = oracle.Graph.from_nodes_edges([(1, "ATGC")], [])
    assert oracle.Index(g, 2).seq_rev == "GCAT"
    assert oracle.Index(simple_graph(oracle), 3).seq_rev == "TGCTCAGT"


def test_table_consistency(oracle):
    """src/index.rs:966-1075 test_table: first/last base of every hit agree with the k-mer"""
    ix = oracle.Index(simple_graph(oracle), 3)
    for kmer in ix.kmer_keys():
        hits = ix.find_positions_for_query_kmer(kmer)
        assert hits
        for (so, sp), (eo, ep) in hits:
            ref = ix.seq_fwd if so == F else ix.seq_rev
            sub = ref[sp:ep]
            assert sub[0] == kmer[0] and sub[-1] == kmer[2]


def test_index_access(oracle):
    """src/index.rs:1109-1129 test_index_access"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert ix.find_positions_for_query_kmer("ACT") == [((F, 0), (F, 3))]


def test_index_access_2(oracle):
    """src/index.rs:1132-1170 test_index_access_2"""
    g = oracle.Graph.from_nodes_edges([(1, "TTT"), (2, "AAA")], [(1, 2)])
    ix = oracle.Index(g, 3)
    assert ix.find_positions_for_query_kmer("TTT") == [((F, 0), (F, 3)), ((R, 0), (R, 3))]


def test_index_access_nodes(oracle):
    """src/index.rs:1219-1243 test_index_access_nodes"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert ix.node_id_from_seqpos(F, 0) == 1
    assert ix.node_id_from_seqpos(F, 2) == 2
    assert ix.node_id_from_seqpos(R, 0) == 4


def test_edges_from_handle(oracle):
    """src/index.rs:1261-1284 test_edges_from_handle"""
    ix = oracle.Index(simple_graph(oracle), 3)
    h = [H(oracle, i) for i in (1, 2, 3, 4)]
    assert ix.edges_from_handle(h[0]) == [h[1], h[2]]
    assert ix.edges_from_handle(h[1]) == [h[0], h[3]]
    assert ix.edges_from_handle(h[2]) == [h[0], h[3]]
    assert ix.edges_from_handle(h[3]) == [h[1], h[2]]


def test_index_incoming_outgoing_edges(oracle):
    """src/index.rs:1287-1367 test_index_incoming_outgoing_edges (forward and reverse handles)"""
    o = oracle
    ix = o.Index(simple_graph(o), 3)
    h = [H(o, i) for i in (1, 2, 3, 4)]
    fl = lambda x: x ^ 1
    assert ix.incoming_edges_from_handle(h[0]) == []
    assert ix.outgoing_edges_from_handle(h[0]) == [h[1], h[2]]
    assert ix.incoming_edges_from_handle(h[1]) == [h[0]]
    assert ix.outgoing_edges_from_handle(h[1]) == [h[3]]
    assert ix.incoming_edges_from_handle(h[2]) == [h[0]]
    assert ix.outgoing_edges_from_handle(h[2]) == [h[3]]
    assert ix.incoming_edges_from_handle(h[3]) == [h[1], h[2]]
    assert ix.outgoing_edges_from_handle(h[3]) == []
    assert ix.incoming_edges_from_handle(fl(h[0])) == [fl(h[2]), fl(h[1])]
    assert ix.outgoing_edges_from_handle(fl(h[0])) == []
    assert ix.incoming_edges_from_handle(fl(h[3])) == []
    assert ix.outgoing_edges_from_handle(fl(h[3])) == [fl(h[2]), fl(h[1])]
    assert ix.incoming_edges_from_handle(fl(h[1])) == [fl(h[3])]
    assert ix.outgoing_edges_from_handle(fl(h[1])) == [fl(h[0])]


def test_seq_from_handle(oracle):
    """src/index.rs:1395-1422 test_seq_from_handle"""
    g = simple_graph(oracle)
    ix = oracle.Index(g, 3)
    for i in (1, 2, 3, 4):
        for rev in (False, True):
            h = H(oracle, i, rev)
            assert ix.seq_from_handle(h) == g.sequence(h)
    assert g.sequence(H(oracle, 4, True)) == "TGC"


def test_handle_from_seqpos(oracle):
    """src/index.rs:1425-1443 test_handle_from_seqpos"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert ix.handle_from_seqpos(F, 0) == H(oracle, 1)
    assert ix.handle_from_seqpos(R, 0) == H(oracle, 4, True)


def test_reverse_handles(oracle):
    """src/index.rs:1446-1476 test_reverse_handles"""
    o = oracle
    g = o.Graph.from_nodes_edges([(1, "AAA"), (2, "TTT"), (3, "CCC"), (4, "GGG")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    ix = o.Index(g, 3)
    for i in (1, 2, 3, 4):
        rev = H(o, i, True)
        for (so, sp), _ in ix.find_positions_for_query_kmer(g.sequence(rev)):
            got = ix.handle_from_seqpos(so, sp)
            if got & 1:
                assert got == rev


def test_wrong_index(oracle):
    """src/index.rs:1491-1631 test_wrong_index: fwd/rev node id at every node start"""
    g = oracle.Graph.from_nodes_edges(
        [(1, "AAAAAAA"), (2, "TTT"), (3, "CCC"), (4, "GGGGGGG"), (5, "GGG"), (6, "CCC"), (7, "TTTTTTT")],
        [(1, 2), (1, 3), (2, 4), (3, 4), (4, 5), (4, 6), (5, 7), (6, 7)],
    )
    ix = oracle.Index(g, 11)
    nr = ix.node_ref()
    for i in range(len(nr) - 1):
        assert ix.node_id_from_seqpos(F, nr[i][0]) == i + 1
        assert ix.node_id_from_seqpos(R, nr[i][0]) == ix.n_nodes - i


def test_inverse_rank(oracle):
    """src/index.rs:1634-1650 test_inverse_rank"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert [ix.get_bv_rank(i) for i in range(8)] == [1, 2, 2, 3, 3, 4, 4, 4]
    assert [ix.get_bv_inverse_rank(i) for i in range(8)] == [1, 1, 1, 2, 2, 3, 3, 4]


def test_index_returns_same_positions(oracle):
    """src/index.rs:1653-1666: select(id) == node_ref[id-1].seq_idx"""
    ix = oracle.Index(simple_graph(oracle), 3)
    nr = ix.node_ref()
    for i in (1, 2, 3, 4):
        assert ix.get_bv_select(i) == nr[i - 1][0]


def test_index_contains_multinode_kmers(oracle):
    """src/index.rs:1669-1732: a k-mer's span in linear coordinates can exceed k"""
    o = oracle
    ix = o.Index(simple_graph(o), 5)
    assert ix.find_positions_for_query_kmer("ACTGC")
    assert ix.find_positions_for_query_kmer("CTGCA")
    g2 = o.Graph.from_nodes_edges([(1, "ACG"), (2, "C"), (3, "G"), (4, "TTTTT")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    ix2 = o.Index(g2, 5)
    p = ix2.find_positions_for_query_kmer("ACGGT")[0]
    assert (p[0][1], p[1][1]) == (0, 6)
    p = ix2.find_positions_for_query_kmer("GCTTT")[0]
    assert (p[0][1], p[1][1]) == (2, 8)
    p = ix2.find_positions_for_query_kmer("CTTTT")[0]
    assert (p[0][1], p[1][1]) == (3, 9)
    g3 = o.Graph.from_nodes_edges(
        [(1, "ACG"), (2, "C"), (3, "G"), (4, "TTTTT"), (5, "TA"), (6, "CG"), (7, "TTT")],
        [(1, 2), (1, 3), (2, 4), (3, 4), (4, 5), (4, 6), (5, 7), (6, 7)],
    )
    p = o.Index(g3, 5).find_positions_for_query_kmer("TTCGT")[0]
    assert (p[0][1], p[1][1]) == (8, 15)


# ---------------------------------------------------------------- src/kmer.rs
def test_seqorient_seqpos_ordering(oracle):
    """src/kmer.rs:942-983: Forward < Reverse, (orient, position) lexicographic -- observed through the
    sorted positions of a k-mer present on both strands"""
    g = oracle.Graph.from_nodes_edges([(1, "TTT"), (2, "AAA")], [(1, 2)])
    hits = oracle.Index(g, 3).find_positions_for_query_kmer("AAA")
    assert hits == sorted(hits)
    assert [h[0][0] for h in hits] == [F, R]


# ---------------------------------------------------------------- src/io.rs
def test_read_fasta_fastq(oracle, data_dir):
    """src/io.rs:267-308 test_read_fasta_single_read / test_read_fasta_headers / test_read_fastq"""
    s = oracle.read_seqs_from_file(f"{data_dir}/single-read-test.fa")
    assert s == [("seq0", "AAAAACGTTAAATTTGGCATCGTAGCAAAAA")]
    m = oracle.read_seqs_from_file(f"{data_dir}/multiple-read-test.fa")
    assert m == [("seq0", "AAAAACGTTAAATTTGGCATCGTAGCAAAAA"), ("seq1", "TTTCGTTAAATTTGGCATCGTAGCTTT")]
    assert len(oracle.read_seqs_from_file(f"{data_dir}/test.fq")) == 1


def test_split_into_kmers(oracle):
    """src/io.rs:310-334 test_split_ok/greater/lesser, observed as anchors per query k-mer"""
    g = oracle.Graph.from_nodes_edges([(1, "AAACTG")], [])
    ix = oracle.Index(g, 3)
    a = ix.anchors_for_query("AAACTG")
    assert [(x.query_begin, x.query_end) for x in a] == [(0, 3), (1, 4), (2, 5), (3, 6)]
    assert ix.anchors_for_query("AA") == []
    ix4 = oracle.Index(oracle.Graph.from_nodes_edges([(1, "AAAA")], []), 4)
    assert ix4.anchors_for_query("AAA") == []


# ---------------------------------------------------------------- src/chain.rs
def test_anchors_found(oracle):
    """src/chain.rs:742-753 anchors_found"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert len(ix.anchors_for_query("ACTGCA", True)) == 4
    assert len(ix.anchors_for_query("AGAGC", True)) == 3


def test_anchors_found_2(oracle):
    """src/chain.rs:756-777 anchors_found_2"""
    g = oracle.Graph.from_nodes_edges(
        [(1, "AAAAAAAAAAA"), (2, "C"), (3, "G"), (4, "TTTTTTTTTTTT")], [(1, 2), (1, 3), (2, 4), (3, 4)]
    )
    ix = oracle.Index(g, 11)
    assert len(ix.anchors_for_query("AAAAACTTTTTT", True)) == 2


def test_simple_anchors(oracle):
    """src/chain.rs:806-823 test_simple_anchors"""
    ix = oracle.Index(oracle.Graph.from_nodes_edges([(1, "ACT")], []), 3)
    a = ix.anchors_for_query("ACT", False)
    assert len(a) == 1
    assert (a[0].query_begin, a[0].query_end, a[0].target_begin, a[0].target_end) == (0, 3, (F, 0), (F, 3))
    assert a[0].max_chain_score == 3.0 and a[0].best_predecessor_id == -1


def test_simple_anchors_reverse(oracle):
    """src/chain.rs:826-859 test_simple_anchors_reverse"""
    o = oracle
    g = o.Graph.from_nodes_edges([(1, "AAA"), (2, "CCC"), (3, "GGG"), (4, "AAA")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    ix = o.Index(g, 3)
    a = ix.anchors_for_query("TTT", False)
    assert len(a) == 2
    h0 = ix.handle_from_seqpos(*a[0].target_begin)
    assert h0 == H(o, 4, True)
    assert h0 == ix.handle_from_seqpos(a[0].target_end[0], a[0].target_end[1] - 1)
    h1 = ix.handle_from_seqpos(*a[1].target_begin)
    assert h1 == H(o, 1, True)
    assert h1 == ix.handle_from_seqpos(a[1].target_end[0], a[1].target_end[1] - 1)


def test_simple_anchors_reverse_2(oracle):
    """src/chain.rs:862-888 test_simple_anchors_reverse_2"""
    o = oracle
    g = o.Graph.from_nodes_edges([(1, "AAA"), (2, "CCC"), (3, "GGG"), (4, "AAA")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    ix = o.Index(g, 9)
    a = ix.anchors_for_query("TTTCCCTTT", False)
    assert len(a) == 1
    assert ix.handle_from_seqpos(*a[0].target_begin) == H(o, 4, True)
    assert ix.handle_from_seqpos(a[0].target_end[0], a[0].target_end[1] - 1) == H(o, 1, True)


def test_anchors_and_no_anchors(oracle):
    """src/chain.rs:891-918 test_anchors / test_no_anchors / test_no_anchors_2"""
    ix = oracle.Index(simple_graph(oracle), 3)
    assert len(ix.anchors_for_query("ACTGCA", False)) >= 4
    assert ix.anchors_for_query("AAATTT", False) == []
    assert ix.anchors_for_query("", False) == []


def test_score_anchors(oracle):
    """src/chain.rs:1001-1035 test_score_anchors: equal target_end.position => -f64::MAX"""
    A = oracle.AnchorT
    a = A(36, 35, 46, (F, 3907), (F, 3918), 31.397, -1)
    b = A(51, 49, 60, (F, 3906), (F, 3918), 49.0, -1)
    assert oracle.score_anchor(a, b, 11, 100) == -1.7976931348623157e308


def test_chains_smoke(oracle, data_dir):
    """src/chain.rs:921-976 test_chains / test_chains_2 (the reference only asserts non-empty)"""
    ix = oracle.Index(simple_graph(oracle), 3)
    r = oracle.chain_anchors(ix, "ACTGCA", 50, 1000, 1, only_forward=False)
    assert len(r.chains) >= 1
    g = oracle.Graph.from_gfa(f"{data_dir}/test.gfa")
    ix2 = oracle.Index(g, 11)
    r2 = oracle.chain_anchors(ix2, ix2.seq_fwd, 50, 1000, 2, only_forward=False)
    assert r2.sorted_anchors
    assert len(r2.chains) >= 1


def test_anchors_found_single_node(oracle):
    """src/chain.rs:780-803 anchors_found_single_node (smoke: must not crash, min_anchors=0)"""
    ix = oracle.Index(oracle.Graph.from_nodes_edges([(1, "AAATTAAA")], []), 3)
    r = oracle.chain_anchors(ix, "AAAAAA", 100, 100, 0)
    assert len(r.chains) >= 1


# ---------------------------------------------------------------- src/align.rs, src/map.rs
def test_to_string_placeholder(oracle):
    """src/align.rs:1204-1231 test_to_string_placeholder"""
    assert oracle.gaf_placeholder("Read1", 6) == "Read1\t6\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n"


def test_config1_test_gfa(oracle, data_dir):
    """src/map.rs:243-259 test_map_no_alignment (BASELINE config #1): test.gfa + single-read-test.fa at
    k=11 share no forward 11-mer, so both GAFs hold one placeholder record."""
    g = oracle.Graph.from_gfa(f"{data_dir}/test.gfa")
    assert g.n_nodes() == 19
    ix = oracle.Index(g, 11)
    assert ix.seq_length == 57
    reads = oracle.read_seqs_from_file(f"{data_dir}/single-read-test.fa")
    cg, ag, st = oracle.map_reads(ix, [r[0] for r in reads], [r[1] for r in reads])
    exp = "seq0\t31\t*\t*\t*\t*\t*\t*\t*\t*\t*\t0\t*\n"
    assert cg == exp and ag == exp
    assert st["n_placeholder_reads"] == 1
