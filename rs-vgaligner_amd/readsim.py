"""Synthetic read generator for the BASELINE.json workloads (host-side tooling, numpy only).

The reference simulates reads with `vg sim -s 77` (experiments-snakemake/Snakefile:32), which is not
available offline.  This module samples reads from the GFA's own P lines instead:
  * a path is drawn uniformly among the paths stored on the forward strand (the mapper only keeps
    forward/forward anchors, src/map.rs:62; a path written with '-' steps is skipped),
  * the start offset is uniform in [0, len(path) - read_len],
  * errors are i.i.d. per template base: substitution / insertion / deletion with the given rates,
    indel lengths geometric(p=0.7) (see _mutate for the exact rule),
  * PRNG: numpy PCG64 seeded with 77.
Config #2: 150 bp, 1 % substitutions.  Config #3: 10 kbp, 3 % sub / 3 % ins / 4 % del ("ONT-like").
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np


def parse_gfa_paths(gfa_path: str) -> Tuple[Dict[int, str], List[Tuple[str, List[Tuple[int, bool]]]]]:
    segs: Dict[int, str] = {}
    paths: List[Tuple[str, List[Tuple[int, bool]]]] = []
    with open(gfa_path) as f:
        for ln in f:
            if ln.startswith("S\t"):
                p = ln.rstrip("\n").split("\t")
                segs[int(p[1])] = p[2]
            elif ln.startswith("P\t"):
                p = ln.rstrip("\n").split("\t")
                steps = [(int(s[:-1]), s[-1] == "-") for s in p[2].split(",") if s]
                paths.append((p[1], steps))
    return segs, paths


_COMP = str.maketrans("ACGTNacgtn", "TGCANtgcan")


def path_sequence(segs: Dict[int, str], steps: List[Tuple[int, bool]]) -> str:
    out = []
    for nid, rev in steps:
        s = segs[nid]
        out.append(s[::-1].translate(_COMP) if rev else s)
    return "".join(out)


@dataclass
class SimRead:
    name: str
    seq: str
    path: str
    offset: int


def _mutate(tpl: np.ndarray, rng: np.random.Generator, sub: float, ins: float, dele: float) -> bytes:
    """Vectorised error model.  Every template base draws one category:
    substitution (a different base), insertion (geometric(0.7) random bases BEFORE the base, the base
    itself kept), deletion start (this base and the following geometric(0.7)-1 bases are dropped), or
    copy.  Events on a base that an earlier deletion removed are dropped with it."""
    L = tpl.shape[0]
    bases = np.frombuffer(b"ACGT", dtype=np.uint8)
    u = rng.random(L)
    is_sub = u < sub
    is_ins = (u >= sub) & (u < sub + ins)
    is_del = (u >= sub + ins) & (u < sub + ins + dele)
    out = tpl.copy()
    ns = int(is_sub.sum())
    if ns:
        # a different base: rotate the 2-bit code by 1..3
        code = np.searchsorted(bases, out[is_sub])
        code = np.where(bases[np.clip(code, 0, 3)] == out[is_sub], code, 0)
        out[is_sub] = bases[(code + rng.integers(1, 4, size=ns)) & 3]
    keep = np.ones(L, dtype=bool)
    nd = int(is_del.sum())
    if nd:
        starts = np.flatnonzero(is_del)
        lens = rng.geometric(0.7, size=nd)
        diff = np.zeros(L + 1, dtype=np.int32)
        np.add.at(diff, starts, 1)
        np.add.at(diff, np.minimum(starts + lens, L), -1)
        keep = np.cumsum(diff[:L]) == 0
    reps = np.ones(L, dtype=np.int64)
    ni = int(is_ins.sum())
    if ni:
        reps[is_ins] += rng.geometric(0.7, size=ni)
    reps[~keep] = 0
    res = np.repeat(out, reps)
    if ni:
        # positions of inserted bases: all but the last copy of each repeated run of an insertion site
        ends = np.cumsum(reps)
        site = np.flatnonzero(is_ins & keep)
        if site.size:
            mask = np.zeros(res.shape[0], dtype=bool)
            for e, n in zip(ends[site], reps[site]):
                mask[e - n:e - 1] = True
            res[mask] = bases[rng.integers(0, 4, size=int(mask.sum()))]
    return res.tobytes()


def simulate_reads(gfa_path: str, n_reads: int, read_len: int, sub: float, ins: float, dele: float,
                   seed: int = 77, forward_only: bool = True) -> List[SimRead]:
    segs, paths = parse_gfa_paths(gfa_path)
    seqs = []
    for name, steps in paths:
        if forward_only and any(rev for _, rev in steps):
            continue
        seqs.append((name, np.frombuffer(path_sequence(segs, steps).encode(), dtype=np.uint8)))
    if not seqs:
        raise ValueError("no forward path in " + gfa_path)
    rng = np.random.Generator(np.random.PCG64(seed))
    reads: List[SimRead] = []
    for r in range(n_reads):
        pi = int(rng.integers(0, len(seqs)))
        name, ps = seqs[pi]
        L = min(read_len, ps.shape[0])
        off = int(rng.integers(0, ps.shape[0] - L + 1))
        tpl = ps[off:off + L]
        if sub == 0 and ins == 0 and dele == 0:
            seq = tpl.tobytes().decode()
        else:
            seq = _mutate(tpl, rng, sub, ins, dele).decode()
        reads.append(SimRead(f"read{r}", seq, name, off))
    return reads


def config2_reads(gfa_path: str, n_reads: int = 1000) -> List[SimRead]:
    return simulate_reads(gfa_path, n_reads, 150, 0.01, 0.0, 0.0, seed=77)


def config3_reads(gfa_path: str, n_reads: int = 10000, read_len: int = 10000) -> List[SimRead]:
    return simulate_reads(gfa_path, n_reads, read_len, 0.03, 0.03, 0.04, seed=77)


def write_fasta(reads: List[SimRead], path: str) -> None:
    """single-line FASTA: the reference emits one read per sequence line (src/io.rs:100-122)"""
    with open(path, "w") as f:
        for r in reads:
            f.write(f">{r.name}\n{r.seq}\n")
