#!/usr/bin/env python3
"""CPU baseline helper of bench.py: maps and aligns a FASTA of reads with the single-threaded oracle, optionally
in several processes over disjoint slices of the reads ("all host cores" mode).  Test infrastructure, like the
rest of oracle/: it is only ever run as a child process of bench.py's cpu_baseline leg.

    python oracle/cpu_bench.py <graph.gfa> <k> <reads.fa> <n_procs> [faithful]   ->  one JSON line

`faithful` switches the oracle's index accessors to the reference's cost shape (linear membership scan, bit-by-bit rank /
select, whole-sequence clone per node lookup: og_set_reference_faithful_costs); the results are the same.
"""
import json
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _work(args):
    gfa, k, names, seqs, faithful = args
    from oracle import oracle_py as o

    g = o.Graph.from_gfa(gfa)
    ix = o.Index(g, k)
    o.lib().og_set_reference_faithful_costs(1 if faithful else 0)
    t = time.perf_counter()
    _, _, st = o.map_reads(ix, names, seqs)
    return st["n_aligned_reads"], time.perf_counter() - t


def main():
    gfa, k, fasta, nproc = sys.argv[1], int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
    faithful = len(sys.argv) > 5 and sys.argv[5] == "faithful"
    from oracle import oracle_py as o

    o.build()
    names, seqs = [], []
    for line in open(fasta):
        line = line.strip()
        if line.startswith(">"):
            names.append(line[1:])
        elif line:
            seqs.append(line)
    nproc = max(1, min(nproc, len(seqs)))
    parts = [(gfa, k, names[i::nproc], seqs[i::nproc], faithful) for i in range(nproc)]
    t0 = time.perf_counter()
    if nproc == 1:
        res = [_work(parts[0])]
    else:
        with mp.get_context("fork").Pool(nproc) as pool:
            res = pool.map(_work, parts)
    wall = time.perf_counter() - t0
    print(json.dumps({"aligned": int(sum(r[0] for r in res)), "reads": len(seqs), "procs": nproc, "wall_s": wall,
                      "max_worker_s": max(r[1] for r in res)}))


if __name__ == "__main__":
    main()
