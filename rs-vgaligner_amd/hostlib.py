"""ctypes binding of libvga_host.so: the C++ host side (Index::build, read_seqs_from_file, map_reads, GAF).
No oracle here; index arrays come from the product's own C++ builder."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import binding

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvga_host.so")
_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    binding.load_library()  # libvga_hip.so first (libvga_host.so links against it)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run __graft_entry__.build()")
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.vgh_last_error.restype = C.c_char_p
    L.vgh_index_build_from_gfa.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64]
    L.vgh_index_build_from_gfa.restype = vp
    L.vgh_index_load.argtypes = [C.c_char_p]
    L.vgh_index_load.restype = vp
    L.vgh_index_store.argtypes = [vp, C.c_char_p]
    L.vgh_index_free.argtypes = [vp]
    L.vgh_index_desc.argtypes = [vp]
    L.vgh_index_desc.restype = C.POINTER(binding.IndexDesc)
    L.vgh_index_upload.argtypes = [vp, vp]
    L.vgh_map_reads.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint64, C.c_uint64,
                                C.c_int, C.c_uint64, C.c_char_p, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.vgh_validation_records.argtypes = [vp, C.c_char_p, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(vp)]
    L.vgh_read_seqs_from_file.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_char_p))]
    L.vgh_read_seqs_from_file.restype = C.c_int64
    L.vgh_plan_shards.argtypes = [C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                                  C.POINTER(C.c_uint32), C.c_uint64]
    L.vgh_plan_shards.restype = C.c_uint64
    L.vgh_map_reads_multi.argtypes = [vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint64, C.c_uint64, C.c_int, C.c_uint64,
                                      C.POINTER(C.c_int), C.c_uint32, C.c_uint64, C.c_char_p, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]
    L.vgh_gaf_alignment_record.argtypes = [C.c_char_p, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint64), C.c_uint64, C.c_uint32, C.c_uint32,
                                           C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p]
    L.vgh_gaf_alignment_record.restype = vp
    L.vgh_textpath_replay.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_char_p, C.c_uint32, C.c_uint32,
                                      C.POINTER(C.c_double)]
    L.vgh_free.argtypes = [vp]
    _lib = L
    return L


class HostError(RuntimeError):
    pass


class HostIndex:
    """Index::build(HashGraph::from_gfa(path), k, max_furcations, max_degree) done by the C++ host."""

    def __init__(self, handle):
        self.L = load_library()
        self.h = handle

    @classmethod
    def build_from_gfa(cls, gfa: str, k: int, max_furcations: int = 100, max_degree: int = 100) -> "HostIndex":
        L = load_library()
        h = L.vgh_index_build_from_gfa(gfa.encode(), k, max_furcations, max_degree)
        if not h:
            raise HostError(L.vgh_last_error().decode())
        return cls(h)

    @classmethod
    def load(cls, path: str) -> "HostIndex":
        L = load_library()
        h = L.vgh_index_load(path.encode())
        if not h:
            raise HostError(L.vgh_last_error().decode())
        return cls(h)

    def store(self, path: str):
        if self.L.vgh_index_store(self.h, path.encode()) != 0:
            raise HostError(self.L.vgh_last_error().decode())

    def desc(self) -> binding.IndexDesc:
        return self.L.vgh_index_desc(self.h).contents

    def arrays(self) -> dict:
        """numpy copies of every field of the vga_index_desc (for cross-checks)"""
        d = self.desc()
        n, e, nk, np_ = int(d.n_nodes), int(d.n_edges), int(d.n_kmers), int(d.n_kmer_pos)
        k = int(d.kmer_length)
        arr = lambda p, m: np.ctypeslib.as_array(p, shape=(m,)).copy() if m else np.zeros(0, np.uint64)
        tab = np.ctypeslib.as_array(C.cast(d.kmer_pos_table, C.POINTER(C.c_uint8)), shape=(np_ * 24,)).copy().view(binding.KMERPOS_DTYPE)
        return dict(k=k, seq_fwd=C.string_at(d.seq_fwd, int(d.seq_length)), node_seq_idx=arr(d.node_seq_idx, n + 1),
                    node_edge_idx=arr(d.node_edge_idx, n + 1), node_edges_to=arr(d.node_edges_to, n + 1),
                    edges=arr(d.edges, e), kmer_keys=C.string_at(d.kmer_keys, nk * k), kmer_starts=arr(d.kmer_starts, nk),
                    kmer_pos_table=tab)

    def upload(self, ctx: binding.Context):
        ctx._check(self.L.vgh_index_upload(self.h, ctx.h))

    def map_reads(self, ctx: binding.Context, names: Sequence[str], seqs: Sequence[str], max_gap: int = 1000,
                  chain_min_n_anchors: int = 3, also_align: bool = True, align_best_n: int = 1,
                  out_prefix: Optional[str] = None) -> Tuple[str, str, int]:
        """map_reads (src/map.rs:27-216): returns (chains GAF, alignments GAF, aligned reads)"""
        n = len(seqs)
        nm = (C.c_char_p * n)(*[s.encode() for s in names])
        sq = (C.c_char_p * n)(*[s.encode() for s in seqs])
        cg, ag, na = C.c_void_p(), C.c_void_p(), C.c_uint64()
        rc = self.L.vgh_map_reads(ctx.h, self.h, n, nm, sq, max_gap, chain_min_n_anchors, 1 if also_align else 0, align_best_n,
                                  out_prefix.encode() if out_prefix else None, C.byref(cg), C.byref(ag), C.byref(na))
        if rc != 0:
            raise HostError(self.L.vgh_last_error().decode())
        c, a = C.string_at(cg).decode(), C.string_at(ag).decode()
        self.L.vgh_free(cg)
        self.L.vgh_free(ag)
        return c, a, int(na.value)

    def textpath_replay(self, ctx: binding.Context, names: Sequence[str], seqs: Sequence[str], out_prefix: str, repeat: int = 4,
                        n_threads: int = 0) -> dict:
        """diagnostics: one chunk mapped and aligned once, its GAF text put together and appended to two files `repeat` times"""
        n = len(seqs)
        nm = (C.c_char_p * n)(*[s.encode() for s in names])
        sq = (C.c_char_p * n)(*[s.encode() for s in seqs])
        out = (C.c_double * 6)()
        rc = self.L.vgh_textpath_replay(ctx.h, self.h, n, nm, sq, out_prefix.encode(), repeat, n_threads, out)
        if rc != 0:
            raise HostError(self.L.vgh_last_error().decode())
        return {"bytes": int(out[0]), "chains_bytes": int(out[1]), "seconds": out[2], "text_seconds": out[3], "write_seconds": out[4],
                "threads": int(out[5]), "gb_per_s": out[0] / 1e9 / max(out[2], 1e-9)}

    def map_reads_multi(self, names: Sequence[str], seqs: Sequence[str], devices: Sequence[int] = (), chunk_reads: int = 32768,
                        max_gap: int = 1000, chain_min_n_anchors: int = 3, also_align: bool = True, align_best_n: int = 1,
                        out_prefix: Optional[str] = None) -> Tuple[str, str, int, int]:
        """vgh::map_reads_multi: one context + host thread per entry of `devices` (all visible GPUs when empty), contiguous
        base-balanced slices in chunks of chunk_reads; returns (chains GAF, alignments GAF, aligned reads, chunks)"""
        n = len(seqs)
        nm = (C.c_char_p * n)(*[s.encode() for s in names])
        sq = (C.c_char_p * n)(*[s.encode() for s in seqs])
        dv = (C.c_int * max(1, len(devices)))(*devices)
        cg, ag, na, nc = C.c_void_p(), C.c_void_p(), C.c_uint64(), C.c_uint64()
        rc = self.L.vgh_map_reads_multi(self.h, n, nm, sq, max_gap, chain_min_n_anchors, 1 if also_align else 0, align_best_n, dv, len(devices),
                                        chunk_reads, out_prefix.encode() if out_prefix else None, C.byref(cg), C.byref(ag), C.byref(na), C.byref(nc))
        if rc != 0:
            raise HostError(self.L.vgh_last_error().decode())
        c, a = C.string_at(cg).decode(), C.string_at(ag).decode()
        self.L.vgh_free(cg)
        self.L.vgh_free(ag)
        return c, a, int(na.value), int(nc.value)

    def validation_records(self, alignments_gaf: str, names: Sequence[str], seqs: Sequence[str]) -> str:
        """create_validation_records + to_string (src/validate.rs:36-145) for every record of an alignments GAF"""
        n = len(seqs)
        nm = (C.c_char_p * n)(*[s.encode() for s in names])
        sq = (C.c_char_p * n)(*[s.encode() for s in seqs])
        out = C.c_void_p()
        if self.L.vgh_validation_records(self.h, alignments_gaf.encode(), n, nm, sq, C.byref(out)) != 0:
            raise HostError(self.L.vgh_last_error().decode())
        s = C.string_at(out).decode()
        self.L.vgh_free(out)
        return s

    def close(self):
        if self.h:
            self.L.vgh_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_shards(lengths: Sequence[int], n_slots: int, chunk_reads: int) -> List[Tuple[int, int, int]]:
    """vgh::plan_shards: [(begin, end, slot)] in read order"""
    L = load_library()
    n = len(lengths)
    ln = (C.c_uint64 * max(1, n))(*lengths)
    cap = n + n_slots + 1
    b, e, s = (C.c_uint64 * cap)(), (C.c_uint64 * cap)(), (C.c_uint32 * cap)()
    m = L.vgh_plan_shards(ln, n, n_slots, chunk_reads, b, e, s, cap)
    assert m <= cap
    return [(int(b[i]), int(e[i]), int(s[i])) for i in range(m)]


def read_seqs_from_file(path: str) -> List[Tuple[str, str]]:
    L = load_library()
    names, seqs = C.POINTER(C.c_char_p)(), C.POINTER(C.c_char_p)()
    n = L.vgh_read_seqs_from_file(path.encode(), C.byref(names), C.byref(seqs))
    if n < 0:
        raise HostError(L.vgh_last_error().decode())
    out = [(names[i].decode(), seqs[i].decode()) for i in range(n)]
    return out


def gaf_alignment_record(name: str, seq_len: int, aligned: bool, handles: Sequence[int], path_length: int, path_start: int, path_end: int,
                         block_length: int, cs: str, cigar: str, prefix: str = "") -> str:
    """the alignments-GAF record the driver's text threads write for one read (vgh::gaf_from_alignment), appended to prefix"""
    L = load_library()
    hs = (C.c_uint64 * max(1, len(handles)))(*handles)
    p = L.vgh_gaf_alignment_record(prefix.encode(), name.encode(), seq_len, 1 if aligned else 0, hs, len(handles), path_length, path_start,
                                   path_end, block_length, cs.encode(), cigar.encode())
    if not p:
        raise HostError(L.vgh_last_error().decode())
    try:
        return C.string_at(p).decode()
    finally:
        L.vgh_free(p)
