"""Shared helpers of the test-suite (test infrastructure: may use the oracle)."""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "tests", "golden", "data")


def pkg():
    import __graft_entry__ as ge

    return ge.load_package()


def oracle_index_arrays(ix):
    """Oracle index -> the host arrays of vga_index_desc."""
    from oracle import oracle_py as o

    p = pkg()
    L = o.lib()
    n = ix.n_kmer_pos
    raw = np.ctypeslib.as_array(C.cast(L.og_index_kmer_pos_table(ix.h), C.POINTER(C.c_uint8)), shape=(n * 32,))
    src = raw.view(np.dtype({"names": ["so", "sp", "eo", "ep"], "formats": ["u1", "<u8", "u1", "<u8"],
                             "offsets": [0, 8, 16, 24], "itemsize": 32}))
    tab = np.zeros(n, dtype=p.binding.KMERPOS_DTYPE)
    tab["start"], tab["end"], tab["start_orient"], tab["end_orient"] = src["sp"], src["ep"], src["so"], src["eo"]
    nr = ix.node_ref()
    return dict(
        k=ix.k, seq_fwd=ix.seq_fwd.encode(),
        node_seq_idx=[x[0] for x in nr], node_edge_idx=[x[1] for x in nr], node_edges_to=[x[2] for x in nr],
        edges=ix.edges(), kmer_keys="".join(ix.kmer_keys()).encode(), kmer_starts=ix.kmer_starts(), kmer_pos_table=tab,
    )


def upload_oracle_index(ctx, ix):
    ctx.upload_index(**oracle_index_arrays(ix))


def compare_map(o, ix, mo, seqs, bandwidth=50, max_gap=1000, min_anchors=3, only_forward=True):
    """GPU vga_map_batch output vs the oracle, read by read, bit for bit.  With only_forward=False the GPU carries the
    orientation of a target coordinate in its bit 31 (include/vga_hip.h)."""
    for r, s in enumerate(seqs):
        ref = o.chain_anchors(ix, s, bandwidth, max_gap, min_anchors, only_forward=only_forward)
        a0, a1 = int(mo.anchor_off[r]), int(mo.anchor_off[r + 1])
        sa = ref.sorted_anchors
        assert a1 - a0 == len(sa), f"read {r}: {a1 - a0} anchors on the GPU, {len(sa)} in the oracle"
        assert mo.anchor_id[a0:a1].tolist() == [x.id for x in sa], f"read {r}: sorted anchor ids differ"
        assert mo.query_begin[a0:a1].tolist() == [x.query_begin for x in sa]
        assert mo.target_begin[a0:a1].tolist() == [x.target_begin[1] | (x.target_begin[0] << 31) for x in sa]
        assert mo.target_end[a0:a1].tolist() == [x.target_end[1] | (x.target_end[0] << 31) for x in sa]
        gf = mo.max_chain_score[a0:a1]
        of = np.array([x.max_chain_score for x in sa], dtype=np.float64)
        assert gf.view(np.uint64).tolist() == of.view(np.uint64).tolist(), f"read {r}: f(i) bit patterns differ"
        assert mo.best_pred_id[a0:a1].tolist() == [x.best_predecessor_id for x in sa], f"read {r}: predecessors differ"
        assert np.float64(mo.curr_max[r]).view(np.uint64) == np.float64(ref.curr_max).view(np.uint64)
        pos_of_id = {x.id: i for i, x in enumerate(sa)}
        want = [(ph, [pos_of_id[x.id] for x in ch]) for ph, ch in zip(ref.is_placeholder, ref.chains)]
        assert mo.chains_of(r) == want, f"read {r}: chains differ"


def run_smoke():
    from oracle import oracle_py as o

    p = pkg()
    gfa = os.path.join(DATA, "DRB1-3123.gfa")
    g = o.Graph.from_gfa(gfa)
    ix = o.Index(g, 11)
    reads = p.readsim.simulate_reads(gfa, 4, 600, 0.03, 0.03, 0.04, seed=5)
    seqs = [r.seq for r in reads]
    ctx = p.Context(0)
    upload_oracle_index(ctx, ix)
    b = ctx.batch(seqs)
    mo = b.map()
    compare_map(o, ix, mo, seqs)
    al = b.align(mo)
    _, ag, _ = o.map_reads(ix, [r.name for r in reads], seqs)
    lines = ag.splitlines()
    for r in range(len(seqs)):
        f = lines[r].split("\t")
        if f[5] == "*":
            assert not al.aligned[r]
            continue
        assert al.aligned[r]
        path = "".join((">" if not (h & 1) else "<") + str(h >> 1)
                       for h in al.path_handles[int(al.path_off[r]):int(al.path_off[r + 1])].tolist())
        assert path == f[5], f"read {r}: node path differs"
        assert f[12] == "as:i:-30 " + al.cs[r] + ",cg:Z:" + al.cigar[r], f"read {r}: cs/CIGAR differ"
        assert (int(f[6]), int(f[7]), int(f[8]), int(f[10])) == (
            int(al.path_length[r]), int(al.path_start[r]), int(al.path_end[r]), int(al.block_length[r]))
    print("smoke ok:", len(seqs), "reads,", mo.n_anchors, "anchors,", al.poa_cells, "POA cells")
