"""ctypes binding of libvga_hip.so (the C ABI declared in include/vga_hip.h).

This is plumbing for tests, bench.py and the Python drivers; the product is the shared library.
There is no CPU fallback: if the library is missing, or no gfx950 device is visible, every entry
point raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VGA_LIB") or os.path.join(_HERE, "libvga_hip.so")  # VGA_LIB: another build of the library (same-box A/B runs)

VGA_OK = 0
ERR_NAMES = {-1: "VGA_ERR_ARG", -2: "VGA_ERR_HIP", -3: "VGA_ERR_NOMEM", -4: "VGA_ERR_UNSUPPORTED",
             -5: "VGA_ERR_NO_INDEX", -6: "VGA_ERR_NO_DEVICE", -7: "VGA_ERR_POOL"}

# every symbol include/vga_hip.h declares
ABI_SYMBOLS = [
    "vga_ctx_create", "vga_ctx_destroy", "vga_last_error", "vga_ctx_synchronize", "vga_abi_version",
    "vga_index_upload", "vga_batch_create", "vga_batch_destroy", "vga_map_default_params", "vga_map_batch",
    "vga_map_result_free", "vga_poa_default_params", "vga_poa_result_free", "vga_poa_batch", "vga_align_batch",
    "vga_align_result_free", "vga_last_kernel_times", "vga_chain_paths_text", "vga_chain_text_free",
    "vga_align_prepare", "vga_ctx_set_pool_fraction", "vga_ctx_set_host_threads",
]


class VgaError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class KmerPos(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("start_orient", C.c_uint8), ("end_orient", C.c_uint8)]


KMERPOS_DTYPE = np.dtype({"names": ["start", "end", "start_orient", "end_orient"],
                          "formats": ["<u8", "<u8", "u1", "u1"], "offsets": [0, 8, 16, 17], "itemsize": C.sizeof(KmerPos)})


class IndexDesc(C.Structure):
    _fields_ = [
        ("kmer_length", C.c_uint32), ("seq_length", C.c_uint64), ("seq_fwd", C.c_char_p), ("n_nodes", C.c_uint64),
        ("node_seq_idx", C.POINTER(C.c_uint64)), ("node_edge_idx", C.POINTER(C.c_uint64)),
        ("node_edges_to", C.POINTER(C.c_uint64)), ("n_edges", C.c_uint64), ("edges", C.POINTER(C.c_uint64)),
        ("n_kmers", C.c_uint64), ("kmer_keys", C.c_char_p), ("kmer_starts", C.POINTER(C.c_uint64)),
        ("n_kmer_pos", C.c_uint64), ("kmer_pos_table", C.POINTER(KmerPos)),
    ]


class MapParams(C.Structure):
    _fields_ = [("bandwidth", C.c_uint32), ("max_gap", C.c_uint64), ("chain_min_n_anchors", C.c_uint32),
                ("only_forward", C.c_int), ("emit_dp", C.c_int)]


class PoaParams(C.Structure):
    _fields_ = [("match", C.c_int32), ("mismatch", C.c_int32), ("gap_open1", C.c_int32), ("gap_ext1", C.c_int32),
                ("gap_open2", C.c_int32), ("gap_ext2", C.c_int32), ("wb", C.c_int32), ("remain_rule", C.c_int32), ("wf", C.c_double)]


_P = C.POINTER


class ChainText(C.Structure):
    _fields_ = [("n_chains", C.c_uint64), ("text_off", C.POINTER(C.c_uint64)), ("text", C.POINTER(C.c_char)), ("ms_total", C.c_float)]


class MapResult(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64), ("n_anchors", C.c_uint64), ("anchor_off", _P(C.c_uint64)), ("anchor_id", _P(C.c_uint32)),
        ("query_begin", _P(C.c_uint32)), ("target_begin", _P(C.c_uint32)), ("target_end", _P(C.c_uint32)),
        ("max_chain_score", _P(C.c_double)), ("best_pred_id", _P(C.c_int32)), ("curr_max", _P(C.c_double)),
        ("n_chains", C.c_uint64), ("chain_off", _P(C.c_uint64)), ("chain_placeholder", _P(C.c_uint8)),
        ("chain_anchor_off", _P(C.c_uint64)), ("chain_anchor_idx", _P(C.c_uint32)),
        ("ms_probe", C.c_float), ("ms_sort", C.c_float), ("ms_chain", C.c_float), ("ms_total", C.c_float),
        ("n_hits", C.c_uint64),
    ]


class PoaResult(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("ok", _P(C.c_uint8)), ("best_score", _P(C.c_int32)), ("path_off", _P(C.c_uint64)),
        ("abpoa_nodes", _P(C.c_uint32)), ("graph_nodes", _P(C.c_uint32)), ("aln_start_offset", _P(C.c_uint32)),
        ("aln_end_offset", _P(C.c_uint32)), ("n_aligned_bases", _P(C.c_uint32)), ("cigar_off", _P(C.c_uint64)),
        ("cigar", _P(C.c_char)), ("cs_off", _P(C.c_uint64)), ("cs", _P(C.c_char)), ("n_rows", _P(C.c_uint64)),
        ("n_cells", _P(C.c_uint64)), ("n_value_cells", _P(C.c_uint64)), ("ms_dp", C.c_float), ("ms_traceback", C.c_float), ("ms_total", C.c_float),
    ]


class AlignResult(C.Structure):
    _fields_ = [
        ("n_reads", C.c_uint64), ("aligned", _P(C.c_uint8)), ("path_off", _P(C.c_uint64)), ("path_handles", _P(C.c_uint64)),
        ("path_length", _P(C.c_uint32)), ("path_start", _P(C.c_uint32)), ("path_end", _P(C.c_uint32)),
        ("block_length", _P(C.c_uint32)), ("best_score", _P(C.c_int32)), ("cigar_off", _P(C.c_uint64)),
        ("cigar", _P(C.c_char)), ("cs_off", _P(C.c_uint64)), ("cs", _P(C.c_char)),
        ("poa_rows", C.c_uint64), ("poa_cells", C.c_uint64), ("poa_value_cells", C.c_uint64), ("poa_problems", C.c_uint64),
        ("ms_subgraph", C.c_float), ("ms_dp", C.c_float), ("ms_traceback", C.c_float), ("ms_total", C.c_float),
        ("result_bytes", C.c_uint64),
    ]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_float), ("launches", C.c_uint32), ("algorithmic_bytes", C.c_uint64),
                ("busy_ms", C.c_float), ("reserved", C.c_uint32)]


_lib = None


def load_library():
    """dlopen libvga_hip.so and declare the prototypes.  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.vga_ctx_create.argtypes = [C.c_int, _P(vp)]
    L.vga_ctx_destroy.argtypes = [vp]
    L.vga_last_error.argtypes = [vp]
    L.vga_last_error.restype = C.c_char_p
    L.vga_ctx_synchronize.argtypes = [vp]
    L.vga_index_upload.argtypes = [vp, _P(IndexDesc)]
    L.vga_batch_create.argtypes = [vp, C.c_char_p, _P(C.c_uint64), C.c_uint64, _P(vp)]
    L.vga_batch_destroy.argtypes = [vp]
    L.vga_map_default_params.argtypes = [_P(MapParams)]
    L.vga_map_batch.argtypes = [vp, _P(MapParams), _P(_P(MapResult))]
    L.vga_map_result_free.argtypes = [_P(MapResult)]
    L.vga_poa_default_params.argtypes = [_P(PoaParams)]
    L.vga_poa_result_free.argtypes = [_P(PoaResult)]
    L.vga_poa_batch.argtypes = [vp, C.c_uint64, _P(C.c_uint64), _P(C.c_uint64), C.c_char_p, _P(C.c_uint64),
                                _P(C.c_uint32), _P(C.c_uint32), _P(C.c_uint64), C.c_char_p, _P(PoaParams),
                                _P(_P(PoaResult))]
    L.vga_align_batch.argtypes = [vp, _P(MapResult), C.c_uint32, _P(PoaParams), _P(_P(AlignResult))]
    L.vga_align_result_free.argtypes = [_P(AlignResult)]
    L.vga_last_kernel_times.argtypes = [vp, _P(KernelTime), C.c_int]
    L.vga_chain_paths_text.argtypes = [vp, _P(MapResult), _P(_P(ChainText))]
    L.vga_align_prepare.argtypes = [vp, C.c_uint64, C.c_uint32]
    L.vga_align_prepare.restype = C.c_int
    L.vga_ctx_set_pool_fraction.argtypes = [vp, C.c_double]
    L.vga_ctx_set_pool_fraction.restype = C.c_int
    L.vga_ctx_set_host_threads.argtypes = [vp, C.c_uint32]
    L.vga_ctx_set_host_threads.restype = C.c_int
    L.vga_chain_text_free.argtypes = [_P(ChainText)]
    _lib = L
    return L


def _np(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def _u64p(a):
    return a.ctypes.data_as(_P(C.c_uint64))


def _u32p(a):
    return a.ctypes.data_as(_P(C.c_uint32))


def default_map_params() -> MapParams:
    p = MapParams()
    load_library().vga_map_default_params(C.byref(p))
    return p


def default_poa_params() -> PoaParams:
    p = PoaParams()
    load_library().vga_poa_default_params(C.byref(p))
    return p


class MapOut:
    """Host copy of a vga_map_result (numpy arrays) that also keeps the C object alive for vga_align_batch."""

    def __init__(self, L, ptr):
        self._L, self._ptr = L, ptr
        r = ptr.contents
        R, A, nc = int(r.n_reads), int(r.n_anchors), int(r.n_chains)
        self.n_reads, self.n_anchors, self.n_chains = R, A, nc
        self.anchor_off = _np(r.anchor_off, R + 1, np.uint64)
        self.anchor_id = _np(r.anchor_id, A, np.uint32) if r.anchor_id else None  # None with emit_dp = 0
        self.query_begin = _np(r.query_begin, A, np.uint32)
        self.target_begin = _np(r.target_begin, A, np.uint32)
        self.target_end = _np(r.target_end, A, np.uint32)
        self.max_chain_score = _np(r.max_chain_score, A, np.float64) if r.max_chain_score else None
        self.best_pred_id = _np(r.best_pred_id, A, np.int32) if r.best_pred_id else None
        self.curr_max = _np(r.curr_max, R, np.float64)
        self.chain_off = _np(r.chain_off, R + 1, np.uint64)
        self.chain_placeholder = _np(r.chain_placeholder, nc, np.uint8)
        self.chain_anchor_off = _np(r.chain_anchor_off, nc + 1, np.uint64)
        self.chain_anchor_idx = _np(r.chain_anchor_idx, int(self.chain_anchor_off[-1]) if nc else 0, np.uint32)
        self.ms = {"probe": r.ms_probe, "sort": r.ms_sort, "chain": r.ms_chain, "total": r.ms_total}
        self.n_hits = int(r.n_hits)

    def chains_of(self, read: int):
        """[(is_placeholder, [sorted-anchor index, ...]), ...] for one read"""
        out = []
        for c in range(int(self.chain_off[read]), int(self.chain_off[read + 1])):
            s, e = int(self.chain_anchor_off[c]), int(self.chain_anchor_off[c + 1])
            out.append((bool(self.chain_placeholder[c]), self.chain_anchor_idx[s:e].tolist()))
        return out

    def close(self):
        if self._ptr is not None:
            self._L.vga_map_result_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PoaOut:
    def __init__(self, L, ptr):
        r = ptr.contents
        n = int(r.n)
        self.n = n
        self.ok = _np(r.ok, n, np.uint8)
        self.best_score = _np(r.best_score, n, np.int32)
        self.path_off = _np(r.path_off, n + 1, np.uint64)
        tp = int(self.path_off[-1])
        self.abpoa_nodes = _np(r.abpoa_nodes, tp, np.uint32)
        self.graph_nodes = _np(r.graph_nodes, tp, np.uint32)
        self.aln_start_offset = _np(r.aln_start_offset, n, np.uint32)
        self.aln_end_offset = _np(r.aln_end_offset, n, np.uint32)
        self.n_aligned_bases = _np(r.n_aligned_bases, n, np.uint32)
        self.cigar_off = _np(r.cigar_off, n + 1, np.uint64)
        self.cs_off = _np(r.cs_off, n + 1, np.uint64)
        cig = C.string_at(r.cigar, int(self.cigar_off[-1])) if n else b""
        cs = C.string_at(r.cs, int(self.cs_off[-1])) if n else b""
        self.cigar = [cig[int(self.cigar_off[i]):int(self.cigar_off[i + 1]) - 1].decode() for i in range(n)]
        self.cs = [cs[int(self.cs_off[i]):int(self.cs_off[i + 1]) - 1].decode() for i in range(n)]
        self.n_rows = _np(r.n_rows, n, np.uint64)
        self.n_cells = _np(r.n_cells, n, np.uint64)
        self.n_value_cells = _np(r.n_value_cells, n, np.uint64)
        self.ms = {"dp": r.ms_dp, "traceback": r.ms_traceback, "total": r.ms_total}
        L.vga_poa_result_free(ptr)


class AlignOut:
    def __init__(self, L, ptr):
        r = ptr.contents
        R = int(r.n_reads)
        self.n_reads = R
        self.aligned = _np(r.aligned, R, np.uint8)
        self.path_off = _np(r.path_off, R + 1, np.uint64)
        self.path_handles = _np(r.path_handles, int(self.path_off[-1]), np.uint64)
        self.path_length = _np(r.path_length, R, np.uint32)
        self.path_start = _np(r.path_start, R, np.uint32)
        self.path_end = _np(r.path_end, R, np.uint32)
        self.block_length = _np(r.block_length, R, np.uint32)
        self.best_score = _np(r.best_score, R, np.int32)
        self.cigar_off = _np(r.cigar_off, R + 1, np.uint64)
        self.cs_off = _np(r.cs_off, R + 1, np.uint64)
        cig = C.string_at(r.cigar, int(self.cigar_off[-1])) if R else b""
        cs = C.string_at(r.cs, int(self.cs_off[-1])) if R else b""
        self.cigar = [cig[int(self.cigar_off[i]):int(self.cigar_off[i + 1]) - 1].decode() for i in range(R)]
        self.cs = [cs[int(self.cs_off[i]):int(self.cs_off[i + 1]) - 1].decode() for i in range(R)]
        self.poa_rows, self.poa_cells, self.poa_problems = int(r.poa_rows), int(r.poa_cells), int(r.poa_problems)
        self.poa_value_cells = int(r.poa_value_cells)
        self.ms = {"subgraph": r.ms_subgraph, "dp": r.ms_dp, "traceback": r.ms_traceback, "total": r.ms_total}
        L.vga_align_result_free(ptr)


class Batch:
    def __init__(self, ctx: "Context", seqs: Sequence[str]):
        self.ctx = ctx
        L = ctx.L
        self.seqs = list(seqs)
        self._concat = "".join(self.seqs).encode()
        off = np.zeros(len(self.seqs) + 1, dtype=np.uint64)
        if self.seqs:
            off[1:] = np.cumsum([len(s) for s in self.seqs], dtype=np.uint64)
        self._off = off
        h = C.c_void_p()
        ctx._check(L.vga_batch_create(ctx.h, self._concat, _u64p(off), len(self.seqs), C.byref(h)))
        self.h = h

    def map(self, params: Optional[MapParams] = None) -> MapOut:
        L = self.ctx.L
        p = params or default_map_params()
        out = _P(MapResult)()
        self.ctx._check(L.vga_map_batch(self.h, C.byref(p), C.byref(out)))
        return MapOut(L, out)

    def align(self, chains: MapOut, best_n: int = 1, params: Optional[PoaParams] = None) -> AlignOut:
        L = self.ctx.L
        p = params or default_poa_params()
        out = _P(AlignResult)()
        self.ctx._check(L.vga_align_batch(self.h, chains._ptr, best_n, C.byref(p), C.byref(out)))
        return AlignOut(L, out)

    def map_raw(self, map_params: Optional[MapParams] = None) -> dict:
        """One map-only pass (anchors + chains; BASELINE config #2) without numpy conversion.  Counters only."""
        L = self.ctx.L
        mp = map_params
        if mp is None:
            mp = default_map_params()
            mp.emit_dp = 0  # the chains GAF reads coordinates and chain membership only
        m = _P(MapResult)()
        self.ctx._check(L.vga_map_batch(self.h, C.byref(mp), C.byref(m)))
        try:
            q = m.contents
            R = int(q.n_reads)
            ph = np.ctypeslib.as_array(q.chain_placeholder, shape=(int(q.n_chains),)) if int(q.n_chains) else np.zeros(0, np.uint8)
            co = np.ctypeslib.as_array(q.chain_off, shape=(R + 1,)) if R else np.zeros(1, np.uint64)
            # a read is mapped when its first chain is a real one
            mapped = int((ph[co[:-1].astype(np.int64)] == 0).sum()) if R else 0
            return dict(n_reads=R, aligned=mapped, n_anchors=int(q.n_anchors), n_hits=int(q.n_hits), n_chains=int(q.n_chains),
                        ms_map=float(q.ms_total), ms_probe=float(q.ms_probe), ms_sort=float(q.ms_sort), ms_chain=float(q.ms_chain),
                        kernels=self.ctx.kernel_times())
        finally:
            L.vga_map_result_free(m)

    def map_align_raw(self, map_params: Optional[MapParams] = None, best_n: int = 1,
                      poa_params: Optional[PoaParams] = None) -> dict:
        """One pass of the whole hot path without converting the results to numpy (bench.py's timed step).
        Returns counters only."""
        import time as _t
        L = self.ctx.L
        mp = map_params
        if mp is None:
            mp = default_map_params()
            mp.emit_dp = 0  # nothing downstream of the chains reads ids / f(i) / predecessors
        pp = poa_params or default_poa_params()
        m = _P(MapResult)()
        t0 = _t.perf_counter()
        self.ctx._check(L.vga_map_batch(self.h, C.byref(mp), C.byref(m)))
        t1 = _t.perf_counter()
        kt_map = self.ctx.kernel_times()
        try:
            a = _P(AlignResult)()
            self.ctx._check(L.vga_align_batch(self.h, m, best_n, C.byref(pp), C.byref(a)))
            t2 = _t.perf_counter()
            kt_aln = self.ctx.kernel_times()
            r, q = a.contents, m.contents
            R = int(r.n_reads)
            aligned = int(np.ctypeslib.as_array(r.aligned, shape=(R,)).sum()) if R else 0
            out = dict(n_reads=R, aligned=aligned, n_anchors=int(q.n_anchors), n_hits=int(q.n_hits),
                       poa_rows=int(r.poa_rows), poa_cells=int(r.poa_cells), poa_value_cells=int(r.poa_value_cells),
                       poa_problems=int(r.poa_problems), path_bases=int(np.ctypeslib.as_array(r.path_length, shape=(R,)).sum()) if R else 0,
                       cigar_bytes=int(r.cigar_off[R]) if R else 0, result_bytes=int(r.result_bytes),
                       ms_map=float(q.ms_total), ms_probe=float(q.ms_probe), ms_sort=float(q.ms_sort), ms_chain=float(q.ms_chain),
                       ms_align=float(r.ms_total), ms_subgraph=float(r.ms_subgraph), ms_dp_summed_launches=float(r.ms_dp),
                       ms_traceback=float(r.ms_traceback), kernels=kt_map + kt_aln)
            t3 = _t.perf_counter()
            L.vga_align_result_free(a)
        finally:
            L.vga_map_result_free(m)
        t4 = _t.perf_counter()
        out["ms_wall_map_call"] = (t1 - t0) * 1e3
        out["ms_wall_align_call"] = (t2 - t1) * 1e3
        out["ms_wall_free"] = (t4 - t3) * 1e3
        return out

    def close(self):
        if self.h:
            self.ctx.L.vga_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One vga_ctx (one GPU)."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        rc = self.L.vga_ctx_create(device, C.byref(h))
        if rc != VGA_OK:
            raise VgaError(rc, f"vga_ctx_create(device={device}) failed: no usable MI355X; this library has no CPU path")
        self.h = h
        self._keep = None

    def _check(self, rc: int):
        if rc != VGA_OK:
            raise VgaError(rc, self.L.vga_last_error(self.h).decode())

    def set_pool_fraction(self, fraction: float):
        """share of the device's free memory this context's traceback pool may take (contexts that share a GPU)"""
        self._check(self.L.vga_ctx_set_pool_fraction(self.h, float(fraction)))

    def set_host_threads(self, n: int):
        """host threads this context's calls fan out to (0: the default)"""
        self._check(self.L.vga_ctx_set_host_threads(self.h, int(n)))

    def upload_index(self, k: int, seq_fwd: bytes, node_seq_idx, node_edge_idx, node_edges_to, edges, kmer_keys: bytes,
                     kmer_starts, kmer_pos_table: np.ndarray):
        """kmer_pos_table: structured array with KMERPOS_DTYPE (records incl. delimiters)."""
        a = [np.ascontiguousarray(x, dtype=np.uint64) for x in (node_seq_idx, node_edge_idx, node_edges_to, edges, kmer_starts)]
        tab = np.ascontiguousarray(kmer_pos_table)
        assert tab.dtype.itemsize == C.sizeof(KmerPos)
        d = IndexDesc()
        d.kmer_length = k
        d.seq_length = len(seq_fwd)
        d.seq_fwd = seq_fwd
        d.n_nodes = len(a[0]) - 1
        d.node_seq_idx, d.node_edge_idx, d.node_edges_to = _u64p(a[0]), _u64p(a[1]), _u64p(a[2])
        d.n_edges = len(a[3])
        d.edges = _u64p(a[3])
        d.n_kmers = len(a[4])
        d.kmer_keys = kmer_keys
        d.kmer_starts = _u64p(a[4])
        d.n_kmer_pos = len(tab)
        d.kmer_pos_table = tab.ctypes.data_as(_P(KmerPos))
        self._check(self.L.vga_index_upload(self.h, C.byref(d)))

    def batch(self, seqs: Sequence[str]) -> Batch:
        return Batch(self, seqs)

    def poa_batch(self, problems, params: Optional[PoaParams] = None) -> PoaOut:
        """problems: [(node strings, [(src, dst), ...], query), ...] -- the create_align_safe arguments."""
        p = params or default_poa_params()
        n = len(problems)
        node_ptr = np.zeros(n + 1, dtype=np.uint64)
        edge_ptr = np.zeros(n + 1, dtype=np.uint64)
        query_off = np.zeros(n + 1, dtype=np.uint64)
        node_off: List[int] = []
        chunks: List[str] = []
        es: List[int] = []
        ed: List[int] = []
        qs: List[str] = []
        pos = 0
        for i, (nodes, edges, query) in enumerate(problems):
            node_ptr[i] = len(node_off)
            for s in nodes:
                node_off.append(pos)
                pos += len(s)
            chunks.extend(nodes)
            edge_ptr[i] = len(es)
            es.extend(e[0] for e in edges)
            ed.extend(e[1] for e in edges)
            query_off[i + 1] = query_off[i] + len(query)
            qs.append(query)
        node_ptr[n] = len(node_off)
        edge_ptr[n] = len(es)
        node_off.append(pos)
        noff = np.asarray(node_off, dtype=np.uint64)
        esa = np.asarray(es if es else [0], dtype=np.uint32)
        eda = np.asarray(ed if ed else [0], dtype=np.uint32)
        out = _P(PoaResult)()
        self._check(self.L.vga_poa_batch(self.h, n, _u64p(node_ptr), _u64p(noff), "".join(chunks).encode(), _u64p(edge_ptr),
                                         _u32p(esa), _u32p(eda), _u64p(query_off), "".join(qs).encode(), C.byref(p),
                                         C.byref(out)))
        return PoaOut(self.L, out)

    def align_prepare(self, n_reads: int, max_read_len: int) -> None:
        """vga_align_prepare: start allocating what the first align_batch of this context will need (returns at once)"""
        self._check(self.L.vga_align_prepare(self.h, int(n_reads), int(max_read_len)))

    def chain_paths_text(self, chains: "MapOut") -> List[bytes]:
        """the path column of every chain's GAF record (vga_chain_paths_text), one bytes object per chain"""
        out = _P(ChainText)()
        self._check(self.L.vga_chain_paths_text(self.h, chains._ptr, C.byref(out)))
        try:
            t = out.contents
            n = int(t.n_chains)
            off = [int(t.text_off[i]) for i in range(n + 1)]
            raw = C.string_at(t.text, off[n]) if off[n] else b""
            return [raw[off[i]:off[i + 1]] for i in range(n)]
        finally:
            self.L.vga_chain_text_free(out)

    def kernel_times(self):
        arr = (KernelTime * 32)()
        n = self.L.vga_last_kernel_times(self.h, arr, 32)
        return [{"name": arr[i].name.decode(), "ms": float(arr[i].ms), "launches": int(arr[i].launches),
                 "algorithmic_bytes": int(arr[i].algorithmic_bytes), "busy_ms": float(arr[i].busy_ms)} for i in range(min(n, 32))]

    def synchronize(self):
        self._check(self.L.vga_ctx_synchronize(self.h))

    def close(self):
        if self.h:
            self.L.vga_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
