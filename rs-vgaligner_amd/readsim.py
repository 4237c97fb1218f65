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
    template_len: int = 0  # bases of the source path the read was drawn from (before errors)


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
        reads.append(SimRead(f"read{r}", seq, name, off, L))
    return reads


def config2_reads(gfa_path: str, n_reads: int = 1000) -> List[SimRead]:
    return simulate_reads(gfa_path, n_reads, 150, 0.01, 0.0, 0.0, seed=77)


def config3_reads(gfa_path: str, n_reads: int = 10000, read_len: int = 10000) -> List[SimRead]:
    return simulate_reads(gfa_path, n_reads, read_len, 0.03, 0.03, 0.04, seed=77)


def truth_gaf(gfa_path: str, reads: List[SimRead]) -> str:
    """The GAF a perfect aligner would write for simulated reads, as far as the node path goes: one record per read
    whose path field lists the nodes the read's template [offset, offset + template length) overlaps on its source
    path (what `vg sim -a` + `vg convert --gam-to-gaf` provide in the reference's experiments, Snakefile:32-40).
    The template length is not stored; it is bounded by the read's own source window: reads are cut from the path
    at `offset` with the requested length before errors are applied, so the window is recomputed from the path."""
    segs, paths = parse_gfa_paths(gfa_path)
    by_name = {name: steps for name, steps in paths}
    out = []
    for r in reads:
        steps = by_name[r.path]
        lens = [len(segs[n]) for n, _ in steps]
        total = sum(lens)
        tpl = min(getattr(r, "template_len", 0) or len(r.seq), total - r.offset)
        lo, hi = r.offset, r.offset + tpl
        pos, nodes = 0, []
        for (nid, rev), ln in zip(steps, lens):
            if pos < hi and pos + ln > lo:
                nodes.append(("<" if rev else ">") + str(nid))
            pos += ln
            if pos >= hi:
                break
        out.append("\t".join([r.name, str(len(r.seq)), "0", str(len(r.seq)), "+", "".join(nodes), str(tpl), "0", str(tpl), "0",
                              str(tpl), "255", "ta:Z:truth"]))
    return "\n".join(out) + ("\n" if out else "")


def write_fasta(reads: List[SimRead], path: str) -> None:
    """single-line FASTA: the reference emits one read per sequence line (src/io.rs:100-122)"""
    with open(path, "w") as f:
        for r in reads:
            f.write(f">{r.name}\n{r.seq}\n")


# ---------------------------------------------------------------------------------------------------------
# graphs of BASELINE.json configs #4 and #5

# The HLA-zoo loci whose shipped graph.gfa is forward-only, acyclic and already numbered in topological order (the
# reference requires `odgi sort`-ed input and odgi is not available offline; the other 11 loci contain reverse links,
# back edges or self loops as shipped).  Config #4 = the disjoint union of these nine.
HLA_FORWARD_ACYCLIC = ["hla/1-simple", "DRB1-3123", "hla/6-DRB5-3127", "hla/11-C-3107-spoa", "hla/12-DMA-3108-spoa",
                       "hla/13-V-352962-spoa", "hla/14-DOB-3112-spoa", "hla/18-B-3106-smooth", "hla/19-MICB-4227-smooth"]


def config4_graph(data_dir: str, out_path: str) -> Tuple[int, int, int]:
    """the merged HLA graph of config #4 from the graph files under tests/golden/data"""
    import os

    return merge_gfas([os.path.join(data_dir, n + ".gfa") for n in HLA_FORWARD_ACYCLIC], out_path)


def merge_gfas(gfa_paths: List[str], out_path: str) -> Tuple[int, int, int]:
    """Disjoint union of GFA graphs with cumulative id offsets (ids stay contiguous and topological when every input's
    are).  Path names get the input's ordinal as a prefix.  Returns (nodes, edges, bases)."""
    off = 0
    n_edges = n_bases = 0
    with open(out_path, "w") as out:
        out.write("H\tVN:Z:1.0\n")
        for gi, gp in enumerate(gfa_paths):
            top = 0
            with open(gp) as f:
                for ln in f:
                    p = ln.rstrip("\n").split("\t")
                    if p[0] == "S":
                        top = max(top, int(p[1]))
                        n_bases += len(p[2])
                        out.write("S\t%d\t%s\n" % (int(p[1]) + off, p[2]))
                    elif p[0] == "L":
                        n_edges += 1
                        out.write("L\t%d\t%s\t%d\t%s\t%s\n" % (int(p[1]) + off, p[2], int(p[3]) + off, p[4], p[5] if len(p) > 5 else "0M"))
                    elif p[0] == "P":
                        steps = ",".join("%d%s" % (int(s[:-1]) + off, s[-1]) for s in p[2].split(",") if s)
                        out.write("P\tg%d_%s\t%s\t*\n" % (gi, p[1], steps))
            off += top
    return off, n_edges, n_bases


def synth_pangenome(out_path: str, total_bp: int = 1_000_000, seed: int = 77, n_haps: int = 16, mean_node: int = 32,
                    snp_every: int = 100, indel_every: int = 1000) -> Tuple[int, int, int]:
    """Config #5: a random backbone (uniform ACGT) cut into nodes of mean `mean_node` bp (geometric), a SNP bubble
    about every `snp_every` bp and a 1-20 bp indel bubble about every `indel_every` bp; ids are assigned in
    topological order (what `odgi sort -p Ygs` would produce); `n_haps` haplotype paths pick bubble arms at random.
    Returns (nodes, edges, bases)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    segs: List[bytes] = []
    edges: List[Tuple[int, int]] = []
    # per backbone step: ("n", id) plain node, ("s", ref, alt) SNP arms, ("i", id) optional indel node
    plan: List[Tuple] = []
    made = 0
    p_snp = mean_node / snp_every
    p_indel = mean_node / indel_every

    def new_node(n):
        segs.append(acgt[rng.integers(0, 4, size=n)].tobytes())
        return len(segs)  # 1-based id

    prev: List[int] = []
    skip_from: List[int] = []  # nodes that may also jump over the next (optional) node
    while made < total_bp:
        n = int(rng.geometric(1.0 / mean_node))
        a = new_node(n)
        made += n
        for pnode in prev + skip_from:
            edges.append((pnode, a))
        plan.append(("n", a))
        prev, skip_from = [a], []
        u = rng.random()
        if made >= total_bp:
            break
        if u < p_snp:
            ref = new_node(1)
            b = int(rng.integers(1, 4))
            alt_base = acgt[(int(np.searchsorted(acgt, segs[ref - 1][0])) + b) & 3]
            segs.append(bytes([alt_base]))
            alt = len(segs)
            edges += [(a, ref), (a, alt)]
            plan.append(("s", ref, alt))
            prev = [ref, alt]
            made += 1
        elif u < p_snp + p_indel:
            ins = new_node(int(rng.integers(1, 21)))
            edges.append((a, ins))
            plan.append(("i", ins))
            prev, skip_from = [ins], [a]
            made += len(segs[ins - 1])
    with open(out_path, "w") as out:
        out.write("H\tVN:Z:1.0\n")
        for i, s in enumerate(segs):
            out.write("S\t%d\t%s\n" % (i + 1, s.decode()))
        for a, b in sorted(edges):
            out.write("L\t%d\t+\t%d\t+\t0M\n" % (a, b))
        for h in range(n_haps):
            steps = []
            for st in plan:
                if st[0] == "n":
                    steps.append(st[1])
                elif st[0] == "s":
                    steps.append(st[1] if rng.random() < 0.5 else st[2])
                elif rng.random() < 0.5:
                    steps.append(st[1])
            out.write("P\thap%d\t%s\t*\n" % (h, ",".join("%d+" % s for s in steps)))
    return len(segs), len(edges), sum(len(s) for s in segs)
