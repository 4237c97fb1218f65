"""GPU parity tests: every call goes through the C ABI of libvga_hip.so and is compared with the CPU
oracle on the same inputs -- bit-exact for anchors, sort order, f(i) (f64 bit patterns), predecessors,
chain membership, POA score / CIGAR / cs / node path."""
import os
import random

import numpy as np
import pytest

from helpers import DATA, compare_map, pkg, run_smoke, upload_oracle_index

pytestmark = pytest.mark.gpu

DRB1 = os.path.join(DATA, "DRB1-3123.gfa")


@pytest.fixture(scope="module")
def ctx():
    p = pkg()
    c = p.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def drb1(oracle):
    g = oracle.Graph.from_gfa(DRB1)
    return g, oracle.Index(g, 11)


def simple_graph(o):
    return o.Graph.from_nodes_edges([(1, "A"), (2, "CT"), (3, "GA"), (4, "GCA")], [(1, 2), (1, 3), (2, 4), (3, 4)])


def test_smoke():
    run_smoke()


def test_anchor_vectors_of_the_reference(oracle, ctx):
    """src/chain.rs:742-777 anchors_found / anchors_found_2 through the device probe"""
    ix = oracle.Index(simple_graph(oracle), 3)
    upload_oracle_index(ctx, ix)
    seqs = ["ACTGCA", "AGAGC", "AAATTT", "", "AC"]
    mo = ctx.batch(seqs).map(pkg().default_map_params())
    counts = np.diff(mo.anchor_off).tolist()
    assert counts == [4, 3, 0, 0, 0]
    p = pkg().default_map_params()
    p.chain_min_n_anchors = 1
    compare_map(oracle, ix, ctx.batch(seqs).map(p), seqs, min_anchors=1)
    g2 = oracle.Graph.from_nodes_edges([(1, "AAAAAAAAAAA"), (2, "C"), (3, "G"), (4, "TTTTTTTTTTTT")],
                                       [(1, 2), (1, 3), (2, 4), (3, 4)])
    ix2 = oracle.Index(g2, 11)
    upload_oracle_index(ctx, ix2)
    mo2 = ctx.batch(["AAAAACTTTTTT"]).map()
    assert int(mo2.n_anchors) == 2
    compare_map(oracle, ix2, mo2, ["AAAAACTTTTTT"])


def test_reverse_strand_anchor_vectors_of_the_reference(oracle, ctx, drb1):
    """vga_map_params.only_forward = 0, i.e. anchors_for_query(.., false): the reference's own vectors for it
    (src/chain.rs:806-976: test_simple_anchors, test_simple_anchors_reverse, _reverse_2, test_anchors, test_no_anchors(_2),
    test_chains, test_chains_2) through the device probe / sort / chaining, and GPU == oracle on real reads.  Bit 31 of a
    target coordinate is its orientation.  vga_align_batch refuses chains that hold reverse-strand anchors."""
    p = pkg()
    mp = p.default_map_params()
    mp.only_forward = 0
    mp.chain_min_n_anchors = 1
    REV = 1 << 31
    # test_simple_anchors
    ix = oracle.Index(oracle.Graph.from_nodes_edges([(1, "ACT")], []), 3)
    upload_oracle_index(ctx, ix)
    mo = ctx.batch(["ACT"]).map(mp)
    assert int(mo.n_anchors) == 1
    assert (int(mo.query_begin[0]), int(mo.target_begin[0]), int(mo.target_end[0])) == (0, 0, 3)
    # test_simple_anchors_reverse / _reverse_2: AAA - (CCC | GGG) - AAA, reads from the reverse strand
    diamond = oracle.Graph.from_nodes_edges([(1, "AAA"), (2, "CCC"), (3, "GGG"), (4, "AAA")], [(1, 2), (1, 3), (2, 4), (3, 4)])
    ix = oracle.Index(diamond, 3)
    upload_oracle_index(ctx, ix)
    mo = ctx.batch(["TTT"]).map(mp)
    assert int(mo.n_anchors) == 2 and all(int(x) & REV for x in mo.target_begin) and all(int(x) & REV for x in mo.target_end)
    by_id = sorted(zip(mo.anchor_id.tolist(), mo.target_begin.tolist(), mo.target_end.tolist()))
    nodes = [(ix.handle_from_seqpos(1, tb & ~REV), ix.handle_from_seqpos(1, (te & ~REV) - 1)) for _, tb, te in by_id]
    assert nodes == [(4 * 2 + 1, 4 * 2 + 1), (1 * 2 + 1, 1 * 2 + 1)]  # anchor 0 on node 4 reversed, anchor 1 on node 1 reversed
    compare_map(oracle, ix, mo, ["TTT"], min_anchors=1, only_forward=False)
    ix9 = oracle.Index(diamond, 9)
    upload_oracle_index(ctx, ix9)
    mo = ctx.batch(["TTTCCCTTT"]).map(mp)
    assert int(mo.n_anchors) == 1
    assert ix9.handle_from_seqpos(1, int(mo.target_begin[0]) & ~REV) == 4 * 2 + 1
    assert ix9.handle_from_seqpos(1, (int(mo.target_end[0]) & ~REV) - 1) == 1 * 2 + 1
    # test_anchors / test_no_anchors / test_no_anchors_2 / test_chains
    ix = oracle.Index(simple_graph(oracle), 3)
    upload_oracle_index(ctx, ix)
    seqs = ["ACTGCA", "AAATTT", "", "TGCAGT"]
    mo = ctx.batch(seqs).map(mp)
    counts = np.diff(mo.anchor_off).tolist()
    assert counts[0] >= 4 and counts[1] == 0 and counts[2] == 0
    compare_map(oracle, ix, mo, seqs, min_anchors=1, only_forward=False)
    assert not mo.chains_of(0)[0][0]
    # test_chains_2: the whole linearisation of test.gfa as a read, min 2 anchors
    g = oracle.Graph.from_gfa(os.path.join(DATA, "test.gfa"))
    ixt = oracle.Index(g, 11)
    upload_oracle_index(ctx, ixt)
    mp.chain_min_n_anchors = 2
    mo = ctx.batch([ixt.seq_fwd]).map(mp)
    assert int(mo.n_anchors) > 0 and not mo.chains_of(0)[0][0]
    compare_map(oracle, ixt, mo, [ixt.seq_fwd], min_anchors=2, only_forward=False)
    # real reads, both strands: forward reads and their reverse complements against DRB1-3123
    _, ixd = drb1
    upload_oracle_index(ctx, ixd)
    mp.chain_min_n_anchors = 3
    reads = p.readsim.config2_reads(DRB1, 12) + p.readsim.simulate_reads(DRB1, 3, 1200, 0.03, 0.03, 0.04, seed=21)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    seqs = [r.seq for r in reads] + ["".join(comp[c] for c in reversed(r.seq)) for r in reads[:8]]
    b = ctx.batch(seqs)
    mo = b.map(mp)
    compare_map(oracle, ixd, mo, seqs, only_forward=False)
    assert any(int(x) & REV for x in mo.target_end)
    with pytest.raises(p.VgaError) as e:  # RangeOrient::Reverse / Both is not built
        b.align(mo)
    assert e.value.code == -4


def test_map_config1_placeholder(oracle, ctx):
    """BASELINE config #1: test.gfa + single-read-test.fa, k=11 -> one placeholder chain"""
    g = oracle.Graph.from_gfa(os.path.join(DATA, "test.gfa"))
    ix = oracle.Index(g, 11)
    upload_oracle_index(ctx, ix)
    reads = oracle.read_seqs_from_file(os.path.join(DATA, "single-read-test.fa"))
    b = ctx.batch([r[1] for r in reads])
    mo = b.map()
    assert mo.chains_of(0) == [(True, [])]
    al = b.align(mo)
    assert al.aligned.tolist() == [0]
    # the whole linearisation as a read does chain (src/chain.rs:946-976 test_chains_2)
    p = pkg().default_map_params()
    p.chain_min_n_anchors = 2
    compare_map(oracle, ix, ctx.batch([ix.seq_fwd]).map(p), [ix.seq_fwd], min_anchors=2)


def test_map_config2_sample(oracle, ctx, drb1):
    """BASELINE config #2 (sample): 150 bp reads, 1 % substitutions, map-only"""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.config2_reads(DRB1, 300)
    seqs = [r.seq for r in reads]
    compare_map(oracle, ix, ctx.batch(seqs).map(), seqs)


def test_map_config3_sample_and_edge_reads(oracle, ctx, drb1):
    """BASELINE config #3 (sample): 10 kbp ONT-profile reads + ragged / degenerate reads in one batch"""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.config3_reads(DRB1, 12)
    seqs = [r.seq for r in reads]
    seqs += ["", "ACGT", "N" * 50, seqs[0][:11], seqs[1][:300].replace("A", "N", 3), "ACGTACGTACGTACGTAAAAAAAAAAAAAAAAAAAAAA"]
    compare_map(oracle, ix, ctx.batch(seqs).map(), seqs)


def test_chaining_argmax_in_f64_matches_the_integer_one(oracle, ctx, drb1, monkeypatch):
    """K3 takes a step's argmax on round(1000 score) as 32-bit integers when a read's scores fit (every read of the workloads);
    reads with more than ~195 000 anchors keep the f64 reduction.  VGA_CHAIN_F64=1 sends every read through that path: the same
    f(i) bits, predecessors and chains (src/chain.rs:398-450)."""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.config3_reads(DRB1, 8)
    seqs = [r.seq for r in reads] + [r.seq for r in pkg().readsim.config2_reads(DRB1, 100)] + ["", "ACGT"]
    monkeypatch.setenv("VGA_CHAIN_F64", "1")
    compare_map(oracle, ix, ctx.batch(seqs).map(), seqs)
    monkeypatch.delenv("VGA_CHAIN_F64")
    compare_map(oracle, ix, ctx.batch(seqs).map(), seqs)


def test_map_parameters(oracle, ctx, drb1):
    """non-default bandwidth / max_gap / min_anchors follow the oracle too"""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.simulate_reads(DRB1, 20, 1500, 0.03, 0.03, 0.04, seed=3)
    seqs = [r.seq for r in reads]
    for bw, mg, ma in ((10, 50, 2), (64, 5000, 5), (1, 1000, 1)):
        p = pkg().default_map_params()
        p.bandwidth, p.max_gap, p.chain_min_n_anchors = bw, mg, ma
        compare_map(oracle, ix, ctx.batch(seqs).map(p), seqs, bw, mg, ma)


def _rand_problem(rng, n_nodes, max_len, qlen_scale=1.0, alphabet="ACGT"):
    nodes = ["".join(rng.choice(alphabet) for _ in range(rng.randint(1, max_len))) for _ in range(n_nodes)]
    edges = []
    for v in range(1, n_nodes):
        srcs = {v - 1} if rng.random() < 0.8 else set()
        for _ in range(rng.randint(0, 2)):
            srcs.add(rng.randint(max(0, v - 6), v - 1))
        for s in sorted(srcs):
            edges.append((s, v))
    # walk a random path to make the query, then mutate it
    path, v = [], 0
    out = {}
    for s, d in edges:
        out.setdefault(s, []).append(d)
    while True:
        path.append(v)
        if v not in out:
            break
        v = rng.choice(out[v])
    q = "".join(nodes[v] for v in path)
    q = q[: max(1, int(len(q) * qlen_scale))]
    ql = list(q)
    for i in range(len(ql)):
        x = rng.random()
        if x < 0.05:
            ql[i] = rng.choice("ACGT")
        elif x < 0.08:
            ql[i] = ql[i] + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 4)))
        elif x < 0.11:
            ql[i] = ""
    q = "".join(ql) or "A"
    return nodes, edges, q


def _check_poa(oracle, ctx, problems, params=None, oparams=None):
    out = ctx.poa_batch(problems, params)
    for i, (nodes, edges, q) in enumerate(problems):
        ref = oracle.poa_align(nodes, edges, q, oparams)
        assert bool(out.ok[i]) == ref.ok, f"problem {i}: ok differs"
        assert int(out.n_rows[i]) == ref.n_rows and int(out.n_cells[i]) == ref.n_cells, f"problem {i}: band cells differ"
        if not ref.ok:
            continue
        assert int(out.best_score[i]) == ref.best_score, f"problem {i}: score {out.best_score[i]} vs {ref.best_score}"
        assert out.cigar[i] == ref.cigar, f"problem {i}: CIGAR differs"
        assert out.cs[i] == ref.cs_string, f"problem {i}: cs differs"
        s, e = int(out.path_off[i]), int(out.path_off[i + 1])
        assert out.abpoa_nodes[s:e].tolist() == ref.abpoa_nodes
        assert out.graph_nodes[s:e].tolist() == ref.graph_nodes
        assert (int(out.aln_start_offset[i]), int(out.aln_end_offset[i]), int(out.n_aligned_bases[i])) == (
            ref.aln_start_offset, ref.aln_end_offset, ref.n_aligned_bases)


def test_poa_random_small_graphs(oracle, ctx):
    rng = random.Random(1234)
    problems = [_rand_problem(rng, rng.randint(1, 40), 6) for _ in range(150)]
    problems += [_rand_problem(rng, rng.randint(20, 120), 12, qlen_scale=s) for s in (0.3, 0.6, 1.0, 1.0) for _ in range(10)]
    problems += [_rand_problem(rng, 8, 5, alphabet="ACGTN") for _ in range(20)]
    problems += [(["A"], [], "A"), (["ACGT"], [], "T"), (["A", "C", "G"], [(0, 2), (1, 2)], "CG"),
                 (["AC", "GT"], [], "ACGTACGT")]
    _check_poa(oracle, ctx, problems)


def test_poa_wide_rows_and_unbanded(oracle, ctx):
    """rows wider than one 256-lane step, long nodes, and wb < 0 (no banding)"""
    rng = random.Random(99)
    problems = [_rand_problem(rng, 60, 60) for _ in range(6)]
    _check_poa(oracle, ctx, problems)
    pp, op = pkg().default_poa_params(), oracle.default_poa_params()
    pp.wb = -1
    op.wb = -1
    _check_poa(oracle, ctx, problems[:3] + [_rand_problem(rng, 12, 8) for _ in range(20)], pp, op)
    pp2, op2 = pkg().default_poa_params(), oracle.default_poa_params()
    for p_ in (pp2, op2):
        p_.match, p_.mismatch, p_.gap_open1, p_.gap_ext1, p_.gap_open2, p_.gap_ext2, p_.wb, p_.wf = 1, 3, 2, 2, 10, 1, 3, 0.05
    _check_poa(oracle, ctx, [_rand_problem(rng, 30, 8) for _ in range(30)], pp2, op2)


def _check_align(oracle, ctx, ix, reads, remain_rule=None):
    seqs = [r.seq for r in reads]
    b = ctx.batch(seqs)
    mo = b.map()
    compare_map(oracle, ix, mo, seqs)
    pp, omp = pkg().default_poa_params(), oracle.default_map_params()
    if remain_rule is not None:
        pp.remain_rule = remain_rule
        omp.poa.remain_rule = remain_rule
    al = b.align(mo, params=pp)
    cg, ag, st = oracle.map_reads(ix, [r.name for r in reads], seqs, omp)
    lines = ag.splitlines()
    assert len(lines) == len(seqs)
    for r in range(len(seqs)):
        f = lines[r].split("\t")
        if f[5] == "*":
            assert not al.aligned[r]
            continue
        assert al.aligned[r], f"read {r} aligned on the CPU only"
        hs = al.path_handles[int(al.path_off[r]):int(al.path_off[r + 1])].tolist()
        assert "".join((">" if not (h & 1) else "<") + str(h >> 1) for h in hs) == f[5], f"read {r}: node path"
        assert f[12] == "as:i:-30 " + al.cs[r] + ",cg:Z:" + al.cigar[r], f"read {r}: cs / CIGAR"
        assert (int(f[6]), int(f[7]), int(f[8]), int(f[10])) == (
            int(al.path_length[r]), int(al.path_start[r]), int(al.path_end[r]), int(al.block_length[r]))
    assert al.poa_cells == st["poa_cells"] and al.poa_rows == st["poa_rows"]
    return al


def test_align_prepare_only_moves_the_allocation(oracle, drb1):
    """vga_align_prepare (include/vga_hip.h): the traceback memory of the first vga_align_batch call starts to be allocated
    early, on a thread of its own.  On a context of its own -- first call, then again while the workspace exists, with sizes
    that are nothing like the reads', and right before the context goes -- the alignments stay the oracle's."""
    _, ix = drb1
    c = pkg().Context(0)
    try:
        upload_oracle_index(c, ix)
        reads = pkg().readsim.simulate_reads(DRB1, 10, 2500, 0.03, 0.03, 0.04, seed=23)
        c.align_prepare(len(reads), 2600)
        _check_align(oracle, c, ix, reads)
        c.align_prepare(3, 40)          # (smaller than what exists: nothing to do)
        c.align_prepare(100000, 60000)  # (far more than the reads need: the call after it still fits)
        _check_align(oracle, c, ix, reads[:4])
        c.align_prepare(0, 0)
        c.align_prepare(5000, 9000)     # (still allocating when the context is destroyed)
    finally:
        c.close()


@pytest.mark.parametrize("rule", [0, 1], ids=["longest-path", "first-out-edge"])
def test_remain_rule_of_the_adaptive_band(oracle, ctx, drb1, config4_gfa, monkeypatch, rule):
    """vga_poa_params.remain_rule: which path `remain` (the diagonal term of the adaptive band) follows -- the longest path to
    the sink, or the first out-edge in edge-list order (abPOA's heaviest-out-edge rule on unit weights; oracle/og_poa.c names
    it an open choice).  Both through vga_poa_batch (node tables built by poa_prepare on the host) and through
    vga_align_batch (built by k_sg_emit on the device, and by the host threads with VGA_SUBGRAPH=host), against the oracle."""
    rng = random.Random(31 + rule)
    pp, op = pkg().default_poa_params(), oracle.default_poa_params()
    pp.remain_rule = rule
    op.remain_rule = rule
    problems = [_rand_problem(rng, rng.randint(2, 40), 8) for _ in range(80)] + [_rand_problem(rng, 60, 60) for _ in range(3)]
    # two source nodes, two sinks, arms of very different length: the two rules give different bands here
    problems.append((["ACGTACGTACGTACGTACGTACGTACGTACGT", "TT", "G" * 40, "ACGTAC", "CC"], [(0, 1), (0, 2), (1, 3), (2, 3), (2, 4)],
                     "ACGTACGTACGTACGTACGTACGTACGTACGTTTACGTAC"))
    _check_poa(oracle, ctx, problems, pp, op)
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.simulate_reads(DRB1, 12, 2500, 0.03, 0.03, 0.04, seed=17) + pkg().readsim.config3_reads(DRB1, 2)
    al = _check_align(oracle, ctx, ix, reads, remain_rule=rule)
    monkeypatch.setenv("VGA_SUBGRAPH", "host")
    al_h = _check_align(oracle, ctx, ix, reads[:6], remain_rule=rule)
    monkeypatch.delenv("VGA_SUBGRAPH")
    assert al_h.cigar == al.cigar[:6]
    ix4 = oracle.Index(oracle.Graph.from_gfa(config4_gfa), 11)
    upload_oracle_index(ctx, ix4)
    _check_align(oracle, ctx, ix4, pkg().readsim.simulate_reads(config4_gfa, 5, 3000, 0.03, 0.03, 0.04, seed=5), remain_rule=rule)


def test_align_short_reads(oracle, ctx, drb1):
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    _check_align(oracle, ctx, ix, pkg().readsim.config2_reads(DRB1, 60))


def test_align_config3_sample(oracle, ctx, drb1):
    """BASELINE config #3 (sample): 10 kbp ONT-profile reads, --also-align"""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    _check_align(oracle, ctx, ix, pkg().readsim.simulate_reads(DRB1, 10, 2500, 0.03, 0.03, 0.04, seed=11))
    _check_align(oracle, ctx, ix, pkg().readsim.config3_reads(DRB1, 3))


def test_text_arena_too_small_for_some_problems_falls_back_to_their_operations(oracle, ctx, drb1, monkeypatch):
    """k_poa_text claims a problem's place in the launch's text arena with one atomic add; a problem that finds the arena full is
    flagged and its operations come back instead, encoded on the host -- in one call beside problems whose strings came from the
    device (VGA_POA_TEXT_ARENA pins a small arena).  Records equal the oracle's either way (src/align.rs:1096-1168)."""
    gfa, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.config3_reads(DRB1, 6)
    monkeypatch.setenv("VGA_POA_TEXT_ARENA", "30000")  # (a 10 kbp read's cs + CIGAR + path take ~18 KB: one or two fit)
    al = _check_align(oracle, ctx, ix, reads)
    assert sum(al.aligned) == 6


def test_value_row_ring_keeps_rows_of_unequal_width(oracle, ctx, drb1):
    """Regression (round 2): node-end value rows live in a per-problem ring.  As a byte ring that wrapped whenever a row did
    not fit, a run of rows of unequal width could overwrite the row written two slots earlier while a sibling allele still
    had to read it -- read 1324 of this set (a poor alignment, score -296, whose rows span the whole query) came back
    with a garbage score and a different band.  The ring now has fixed slots of one worst-case row."""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.simulate_reads(DRB1, 2000, 2500, 0.03, 0.03, 0.04, seed=4242)
    al = _check_align(oracle, ctx, ix, reads[1316:1332])
    assert int(al.best_score[8]) == -296


def test_align_config4_merged_hla_sample(oracle, ctx, config4_gfa):
    """BASELINE config #4 (sample): reads from several loci of the merged HLA graph, full path length where the
    locus is shorter than 10 kbp; includes the one-node DRB5 locus (12.9 kbp in a single node)"""
    ix = oracle.Index(oracle.Graph.from_gfa(config4_gfa), 11)
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.config3_reads(config4_gfa, 24)
    seen, pick = set(), []
    for r in reads:  # one read per locus present in the draw
        if r.path.split("_")[0] not in seen:
            seen.add(r.path.split("_")[0])
            pick.append(r)
    assert len(pick) >= 5
    _check_align(oracle, ctx, ix, pick)


def test_subgraphs_from_host_threads_match_the_device_kernels(oracle, ctx, drb1, config4_gfa, monkeypatch):
    """VGA_SUBGRAPH=host: find_range_chain / extend_range_chain_2 / find_nodes_edges_for_abpoa and the POA node tables on host
    threads (the round-1 path) instead of k_sg_mark / k_sg_emit -- same records as the oracle and as the device path, on
    DRB1-3123 and on the merged HLA graph (whose loci keep links between opposite strands)"""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.simulate_reads(DRB1, 6, 2500, 0.03, 0.03, 0.04, seed=14) + pkg().readsim.config2_reads(DRB1, 30)
    dev = _check_align(oracle, ctx, ix, reads)
    monkeypatch.setenv("VGA_SUBGRAPH", "host")
    host = _check_align(oracle, ctx, ix, reads)
    assert dev.cigar == host.cigar and dev.path_handles.tolist() == host.path_handles.tolist() and dev.poa_cells == host.poa_cells
    ix4 = oracle.Index(oracle.Graph.from_gfa(config4_gfa), 11)
    upload_oracle_index(ctx, ix4)
    reads4 = pkg().readsim.config3_reads(config4_gfa, 28)
    host4 = _check_align(oracle, ctx, ix4, reads4)
    monkeypatch.delenv("VGA_SUBGRAPH")
    dev4 = _check_align(oracle, ctx, ix4, reads4)
    assert dev4.cigar == host4.cigar and dev4.path_handles.tolist() == host4.path_handles.tolist()


def test_align_config5_synthetic_pangenome_sample(oracle, ctx, config5_small_gfa):
    """BASELINE config #5 generator (SNP bubble / 100 bp, indel bubble / 1 kbp, 32 bp nodes) at 60 kbp"""
    ix = oracle.Index(oracle.Graph.from_gfa(config5_small_gfa), 11)
    upload_oracle_index(ctx, ix)
    _check_align(oracle, ctx, ix, pkg().readsim.config3_reads(config5_small_gfa, 6))
    _check_align(oracle, ctx, ix, pkg().readsim.config2_reads(config5_small_gfa, 40))


def test_map_without_dp_arrays_feeds_the_same_alignments(oracle, ctx, drb1):
    """vga_map_params.emit_dp = 0: ids / f(i) / predecessors stay on the GPU, chains and alignments are unchanged"""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    reads = pkg().readsim.config2_reads(DRB1, 40)
    b = ctx.batch([r.seq for r in reads])
    full = b.map()
    mp = pkg().default_map_params()
    mp.emit_dp = 0
    lean = b.map(mp)
    assert lean.anchor_id is None and lean.max_chain_score is None and lean.best_pred_id is None
    assert lean.query_begin.tolist() == full.query_begin.tolist() and lean.target_end.tolist() == full.target_end.tolist()
    assert lean.chain_anchor_idx.tolist() == full.chain_anchor_idx.tolist()
    a1, a2 = b.align(full), b.align(lean)
    assert a1.cigar == a2.cigar and a1.cs == a2.cs and a1.path_handles.tolist() == a2.path_handles.tolist()


# the diagnostic VGA_POA_KERNEL=unpacked / full configurations keep whole rows in LDS: long queries are refused there
_long_ok = pytest.mark.skipif(any(k in os.environ.get("VGA_POA_KERNEL", "") for k in ("unpacked", "full")),
                              reason="forced unpacked kernel / full LDS array: ~22 kbp / ~35 kbp limit")


@_long_ok
def test_poa_mixed_problem_sizes_and_unbanded_long_query(oracle, ctx, drb1):
    """random problems of very different sizes in one batch, and an unbanded 33 kbp query against a tiny graph (H falls below
    -30 000 along the first row)."""
    rng = random.Random(4321)
    problems = [_rand_problem(rng, rng.randint(1, 40), 6) for _ in range(60)]
    problems += [_rand_problem(rng, rng.randint(20, 120), 12, qlen_scale=s) for s in (0.3, 1.0) for _ in range(8)]
    problems += [_rand_problem(rng, 60, 60) for _ in range(3)]
    _check_poa(oracle, ctx, problems)
    pp, op = pkg().default_poa_params(), oracle.default_poa_params()
    pp.wb = -1
    op.wb = -1
    long_q = "".join(rng.choice("ACGT") for _ in range(33000))
    _check_poa(oracle, ctx, [(["ACGT", "TTGA"], [(0, 1)], long_q), (["ACGTACGT"], [], "ACGTTCGT")], pp, op)
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    _check_align(oracle, ctx, ix, pkg().readsim.config3_reads(DRB1, 3))


def test_poa_large_gap_penalties_use_the_unpacked_kernel(oracle, ctx):
    """open+extend of the two gap pieces need more than 8 bits together -> k_poa_dp_lds (2-byte gap deltas)"""
    rng = random.Random(7)
    pp, op = pkg().default_poa_params(), oracle.default_poa_params()
    for p_ in (pp, op):
        p_.gap_open1, p_.gap_ext1, p_.gap_open2, p_.gap_ext2 = 6, 3, 200, 1
    _check_poa(oracle, ctx, [_rand_problem(rng, 25, 10) for _ in range(25)] + [_rand_problem(rng, 60, 60) for _ in range(3)], pp, op)


def test_unsupported_inputs_fail_loudly(oracle, ctx, drb1):
    p = pkg()
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    mp = p.default_map_params()
    mp.bandwidth = 100
    with pytest.raises(p.VgaError) as e:
        ctx.batch(["ACGT" * 10]).map(mp)
    assert e.value.code == -4
    with pytest.raises(p.VgaError):  # edge with src >= dst
        ctx.poa_batch([(["AC", "GT"], [(1, 0)], "ACGT")])
    # a query whose column codes no longer fit the LDS next to the row window
    with pytest.raises(p.VgaError) as e:
        ctx.poa_batch([(["ACGT"], [], "A" * 400000)])
    assert e.value.code == -4


@_long_ok
def test_poa_long_query_beyond_the_lds_window(oracle, ctx):
    """60 kbp query: 15x the 4096-column LDS window, band 2 x 610 + 1 (banded), tiny graph + a longer one"""
    rng = random.Random(5)
    q = "".join(rng.choice("ACGT") for _ in range(60000))
    g = [q[i:i + 500] for i in range(0, 3000, 500)]
    _check_poa(oracle, ctx, [(g, [(i, i + 1) for i in range(len(g) - 1)], q[:3000] + q[40000:40500]),
                             (["ACGT", "TTGA"], [(0, 1)], q)])


@_long_ok
def test_poa_query_longer_than_a_pool_chunk_of_scratch(oracle, ctx):
    """140 kbp queries: beyond ~131 kbp the two wide-row scratch rows (12 B per column), a ring of worst-case value
    rows and a multi-predecessor direction row with its three planes no longer fit one 1 MiB pool chunk each; the allocator
    of k_poa_dp_t4 takes whole chunks for them.  Tiny graphs keep the oracle cheap: every row spans the whole query."""
    rng = random.Random(99)
    q = "".join(rng.choice("ACGT") for _ in range(140000))
    bubble = (["ACGT", "TT", "GA", "CCAT"], [(0, 1), (0, 2), (1, 3), (2, 3)], q[:70000] + "ACGTTTCCAT" + q[70000:])
    _check_poa(oracle, ctx, [(["ACGT", "TTGA"], [(0, 1)], q), bubble])


@_long_ok
def test_poa_230_kbp_query_with_the_long_problem_launch_shape(oracle, ctx, monkeypatch):
    """very long problems are launched with 1 024 threads and an 8 192-column LDS window; beyond ~228 kbp the query's
    column codes leave no room for that window.  The launch then halves the window (and only steps the workgroup through
    instantiated sizes) instead of failing the call.  VGA_POA_WINDOW / VGA_POA_NT pin the shape poa_feed::klass selects."""
    monkeypatch.setenv("VGA_POA_WINDOW", "8192")
    monkeypatch.setenv("VGA_POA_NT", "1024")
    rng = random.Random(230)
    q = "".join(rng.choice("ACGT") for _ in range(230000))
    g = [q[100000 + i:100000 + i + 7] for i in range(0, 28, 7)]
    _check_poa(oracle, ctx, [(g, [(0, 1), (1, 2), (2, 3), (0, 2)], q), (["ACGT", "TTGA"], [(0, 1)], q[:229000])])


def _fallback_problems(rng):
    return [_rand_problem(rng, rng.randint(1, 40), 6) for _ in range(30)] + [_rand_problem(rng, 60, 60) for _ in range(2)]


@pytest.mark.parametrize("env", [{"VGA_POA_ARENAS": "0"}, {"VGA_POA_ARENAS": "0", "VGA_POA_SLOTS": "1"}, {"VGA_POA_KERNEL": "unpacked"},
                                 {"VGA_POA_KERNEL": "t4"}, {"VGA_POA_KERNEL": "t4", "VGA_POA_ARENAS": "0"}, {"VGA_POA_KERNEL": "generic"},
                                 {"VGA_POA_TB": "wave"}, {"VGA_POOL_BYTES": "300000000"}, {"VGA_POA_SUB": "7"}, {"VGA_POA_WINDOW": "256"},
                                 {"VGA_POA_NT": "512"}, {"VGA_POA_NT": "1024"}, {"VGA_POA_NT": "512", "VGA_POA_ARENAS": "0"},
                                 {"VGA_SG_SPLIT": "1"}, {"VGA_SG_SPLIT": "2", "VGA_POA_SUB": "2", "VGA_POOL_BYTES": "300000000", "VGA_POA_ARENAS": "0"},
                                 {"VGA_POA_KERNEL": "t6"}, {"VGA_POA_KERNEL": "t6,generic"}, {"VGA_POA_KERNEL": "t5"},
                                 {"VGA_POA_TEXT": "host"}, {"VGA_POA_TEXT_MEMCPY": "1"},
                                 {"VGA_POA_KERNEL": "t7"}, {"VGA_POA_KERNEL": "t7,generic"}, {"VGA_POA_KERNEL": "t7", "VGA_POA_T7_NT": "128", "VGA_POA_T7_WINDOW": "1024"},
                                 {"VGA_POA_KERNEL": "t7", "VGA_POA_T7_NT": "1024"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_poa_paths_the_library_can_fall_back_to(oracle, ctx, drb1, monkeypatch, env):
    """the configurations poa_run selects by itself when it has to -- classic pool instead of arenas (problems too large for
    an arena), k_poa_dp_lds (gap penalties beyond the byte range of k_poa_dp_t4), a traceback kernel of its own, a pool so
    small that sub-batches are re-queued, tiny sub-batches, a narrow LDS window (HBM detour of wide rows), the workgroup sizes of large launches (512) and of
    very long problems (1 024), the one-wave-per-problem kernel of narrow bands on every launch (k_poa_dp_t6: what does not fit its
    window comes back and runs in k_poa_dp_t5) and switched off, cs / CIGAR / node paths from host threads instead of k_poa_text (and
    its text fetched by hipMemcpy instead of the copy kernel), the subgraph store
    in two parts (the second prepared beside the first launch) with launches and re-runs that draw on both -- forced
    through their environment switches and held to the same parity"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = random.Random(31)
    _check_poa(oracle, ctx, _fallback_problems(rng))
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    n = 2 if env.get("VGA_POA_KERNEL") == "unpacked" else (6 if "VGA_SG_SPLIT" in env else 3)
    _check_align(oracle, ctx, ix, pkg().readsim.simulate_reads(DRB1, n, 2500, 0.03, 0.03, 0.04, seed=12))


def test_chunk_pool_that_starts_far_too_small(oracle, drb1, monkeypatch):
    """The traceback pool of chunk-pool mode under shortage: it starts at a fraction of what the resident problems write
    (VGA_POOL_FILL) and can only grow in small segments (VGA_POOL_SEG; at most 80 of them), so workgroups find the free lists
    empty, raise the shortage flag, wait for the keeper thread to list new segments, and some give their problem up -- to a second
    chunk-mode pass and then to the classic pass.  Whatever route a problem took, its alignment is the oracle's."""
    monkeypatch.setenv("VGA_POOL_FILL", "0.002")
    monkeypatch.setenv("VGA_POOL_SEG", str(32 << 20))
    _, ix = drb1
    c = pkg().Context(0)
    try:
        upload_oracle_index(c, ix)
        reads = pkg().readsim.simulate_reads(DRB1, 320, 2500, 0.03, 0.03, 0.04, seed=41)
        _check_align(oracle, c, ix, reads)
    finally:
        c.close()


def test_batch_outlives_its_context(drb1, oracle):
    """a vga_batch handle destroyed after vga_ctx_destroy (the order a garbage collector picks) must not touch the
    freed context, and must not disturb the next context of the process"""
    p = pkg()
    _, ix = drb1
    seqs = [r.seq for r in p.readsim.config2_reads(DRB1, 8)]
    c1 = p.Context(0)
    upload_oracle_index(c1, ix)
    b1 = c1.batch(seqs)
    n1 = int(b1.align(b1.map()).aligned.sum())
    c1.close()
    with pytest.raises(p.VgaError) as e:
        b1.map()
    assert e.value.code == -1  # VGA_ERR_ARG: detached batch
    b1.close()
    c2 = p.Context(0)
    upload_oracle_index(c2, ix)
    b2 = c2.batch(seqs)
    assert int(b2.align(b2.map()).aligned.sum()) == n1
    b2.close()
    c2.close()


def test_two_contexts_driven_from_two_threads(drb1):
    """one context per thread on the same device, concurrently: same records as a sequential run"""
    import threading

    p = pkg()
    _, ix = drb1
    seqs = [r.seq for r in p.readsim.simulate_reads(DRB1, 24, 1500, 0.03, 0.03, 0.04, seed=17)]
    out = {}

    def work(tag, n_rounds):
        c = p.Context(0)
        upload_oracle_index(c, ix)
        b = c.batch(seqs)
        for _ in range(n_rounds):
            al = b.align(b.map())
        out[tag] = (al.cigar, al.cs, al.path_handles.tolist(), al.best_score.tolist())
        b.close()
        c.close()

    work("seq", 1)
    th = [threading.Thread(target=work, args=(t, 3)) for t in ("a", "b")]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert out["a"] == out["seq"] and out["b"] == out["seq"]


def test_align_best_of_n_chains(oracle, ctx, drb1):
    """align_best_n > 1 (src/align.rs:34-55): every read's first min(n, len) chains are aligned and the record with
    the longest path_length wins (stable, None < Some).  Only chains that end on the maximal score are reported
    (src/chain.rs:455-558), so several chains per read need ties: error-free reads made of the same piece two or three
    times -- the copies cannot chain with each other (the target would have to go backwards) and score the same."""
    _, ix = drb1
    upload_oracle_index(ctx, ix)
    p = pkg()
    src = p.readsim.simulate_reads(DRB1, 6, 700, 0.0, 0.0, 0.0, seed=23)
    seqs = [r.seq[:500] + r.seq[:500] for r in src] + [src[0].seq[:300] * 3, src[1].seq]
    names = [f"dup{i}" for i in range(len(seqs))]
    b = ctx.batch(seqs)
    mo = b.map()
    compare_map(oracle, ix, mo, seqs)
    assert max(len(mo.chains_of(r)) for r in range(len(seqs))) >= 2, "the test needs reads with several chains"
    for best_n in (1, 2, 5):
        al = b.align(mo, best_n=best_n)
        mp = oracle.default_map_params()
        mp.align_best_n = best_n
        _, ag, st = oracle.map_reads(ix, names, seqs, mp)
        for r, ln in enumerate(ag.splitlines()):
            f = ln.split("\t")
            assert (f[5] != "*") == bool(al.aligned[r])
            if f[5] == "*":
                continue
            hs = al.path_handles[int(al.path_off[r]):int(al.path_off[r + 1])].tolist()
            assert "".join((">" if not (h & 1) else "<") + str(h >> 1) for h in hs) == f[5], f"best_n {best_n} read {r}: node path"
            assert f[12] == "as:i:-30 " + al.cs[r] + ",cg:Z:" + al.cigar[r], f"best_n {best_n} read {r}: cs / CIGAR"
            assert (int(f[6]), int(f[7]), int(f[8]), int(f[10])) == (
                int(al.path_length[r]), int(al.path_start[r]), int(al.path_end[r]), int(al.block_length[r]))
        assert al.poa_cells == st["poa_cells"], f"best_n {best_n}: the same chains were aligned"


def test_alignment_fields_come_from_the_device_and_settings_are_per_context(oracle, drb1):
    """K4c (k_poa_text): vga_align_batch ships cs / CIGAR / deduplicated node paths as text -- far fewer bytes than the raw
    traceback operations (5 B per alignment column) -- and both routes give the oracle's records.  The same call runs on a
    context with its own pool share and host thread count (vga_ctx_set_pool_fraction / vga_ctx_set_host_threads, ABI 6), which
    reject values outside their ranges."""
    _, ix = drb1
    c = pkg().Context(0)
    try:
        c.set_pool_fraction(0.25)
        c.set_host_threads(2)
        for bad in (0.0, 1.5, -1.0):
            with pytest.raises(pkg().VgaError):
                c.set_pool_fraction(bad)
        with pytest.raises(pkg().VgaError):
            c.set_host_threads(5000)
        upload_oracle_index(c, ix)
        reads = pkg().readsim.simulate_reads(DRB1, 24, 6000, 0.03, 0.03, 0.04, seed=91)
        _check_align(oracle, c, ix, reads)
        seqs = [r.seq for r in reads]
        b = c.batch(seqs)
        dev = b.map_align_raw()
        os.environ["VGA_POA_TEXT"] = "host"
        try:
            host = b.map_align_raw()
        finally:
            del os.environ["VGA_POA_TEXT"]
        assert dev["aligned"] == host["aligned"] == len(reads)
        assert dev["cigar_bytes"] == host["cigar_bytes"] and dev["path_bases"] == host["path_bases"]
        # text: a few bytes per alignment event; operations: 5 bytes per column (+ the gathered node sequences)
        assert dev["result_bytes"] * 4 < host["result_bytes"], (dev["result_bytes"], host["result_bytes"])
    finally:
        c.close()
