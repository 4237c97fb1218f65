"""The N>1 path on CPU: two gloo ranks shard a read list, map their slices (the CPU oracle stands in for the
per-GPU call), and rank 0's in-order concatenation must equal the single-process output."""
import os
import subprocess
import sys
import textwrap

from helpers import DATA, ROOT, pkg


def test_split_by_bases_is_contiguous_and_balanced():
    sp = pkg().sharding.split_by_bases
    lens = [100] * 10 + [1000] * 2 + [50] * 7
    for world in (1, 2, 3, 4, 8):
        parts = sp(lens, world)
        assert len(parts) == world and parts[0][0] == 0 and parts[-1][1] == len(lens)
        assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
    a, b = sp(lens, 2)
    assert abs(sum(lens[a[0]:a[1]]) - sum(lens[b[0]:b[1]])) <= 1000
    assert sp([], 2) == [(0, 0), (0, 0)]


def test_cpp_plan_shards_follows_the_same_rule_and_chunks_in_order():
    """vgh::plan_shards (host/vgh_map.cpp), the slicing the C++ driver `vgaligner map --devices ... --chunk-reads N` uses:
    slices equal sharding.split_by_bases, chunks tile every slice in read order."""
    import random

    p = pkg()
    rng = random.Random(5)
    cases = [[100] * 10 + [1000] * 2 + [50] * 7, [], [7], [rng.randint(1, 20000) for _ in range(997)], [10000] * 4000]
    for lens in cases:
        for world in (1, 2, 3, 8):
            want = p.sharding.split_by_bases(lens, world)
            whole = p.hostlib.plan_shards(lens, world, 0)
            assert [(b, e) for b, e, s in whole if e > b] == [(b, e) for b, e in want if e > b]
            for chunk in (1, 300, 4096):
                plan = p.hostlib.plan_shards(lens, world, chunk)
                assert all(0 < e - b <= chunk for b, e, s in plan)
                pos = 0
                for b, e, s in plan:  # contiguous, in read order, inside the slot's slice
                    assert b == pos and want[s][0] <= b and e <= want[s][1]
                    pos = e
                assert pos == len(lens)
                assert [s for _, _, s in plan] == sorted(s for _, _, s in plan)


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import oracle_py as o
    pkg = ge.load_package()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo")
    gfa = os.path.join({data!r}, "DRB1-3123.gfa")
    reads = pkg.readsim.simulate_reads(gfa, 9, 400, 0.03, 0.03, 0.04, seed=123)
    lens = [len(r.seq) for r in reads]
    s, e = pkg.sharding.split_by_bases(lens, world)[rank]
    # the C++ driver's plan (host/vgh_map.cpp::plan_shards) gives the same slice, cut into chunks of 2 reads here
    mine = [(b, c) for b, c, slot in pkg.hostlib.plan_shards(lens, world, 2) if slot == rank]
    assert mine[0][0] == s and mine[-1][1] == e and all(c - b <= 2 for b, c in mine)
    ix = o.Index(o.Graph.from_gfa(gfa), 11)
    ag, n_al = "", 0
    for b, c in mine:  # chunk by chunk, concatenated in read order: what vgh::map_reads_multi does per device slot
        _, a1, st1 = o.map_reads(ix, [r.name for r in reads[b:c]], [r.seq for r in reads[b:c]])
        ag += a1
        n_al += st1["n_aligned_reads"]
    st = {{"n_aligned_reads": n_al}}
    text = pkg.sharding.gather_in_order(ag, world, rank)
    el, aligned, n = pkg.sharding.reduce_timing(0.5 + rank, st["n_aligned_reads"], e - s, world)
    if rank == 0:
        json.dump({{"gaf": text, "elapsed": el, "aligned": aligned, "n": n}}, open(sys.argv[1], "w"))
    dist.destroy_process_group()
""")


def test_two_rank_gloo_sharding_matches_single_process(tmp_path, oracle):
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER.format(root=ROOT, data=DATA))
    out = tmp_path / "out.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", str(worker), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=600, capture_output=True)
    import json

    got = json.load(open(out))
    p = pkg()
    gfa = os.path.join(DATA, "DRB1-3123.gfa")
    reads = p.readsim.simulate_reads(gfa, 9, 400, 0.03, 0.03, 0.04, seed=123)
    ix = oracle.Index(oracle.Graph.from_gfa(gfa), 11)
    _, ag, st = oracle.map_reads(ix, [r.name for r in reads], [r.seq for r in reads])
    assert got["gaf"] == ag
    assert got["n"] == 9 and got["aligned"] == st["n_aligned_reads"]
    assert got["elapsed"] == 1.5  # MAX over ranks
