"""The `vgaligner` executable (rs-vgaligner_amd/host/vgaligner_main.cpp): flags of src/subcommands/cli.yml, output
naming of index_main.rs / map_main.rs, and -- on a GPU -- the golden GAF files end to end."""
import json
import os
import subprocess
import sys

import pytest

from helpers import DATA, ROOT, pkg

EXE = os.path.join(ROOT, "rs-vgaligner_amd", "vgaligner")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(args, cwd, ok=True, env=None):
    p = subprocess.run([EXE] + args, cwd=cwd, capture_output=True, text=True, timeout=600, env=None if env is None else dict(os.environ, **env))
    assert (p.returncode == 0) == ok, p.stderr
    return p


def test_index_subcommand_and_loud_failures(tmp_path):
    pkg()  # builds the executable if needed
    d = str(tmp_path)
    # long flags of cli.yml; default prefix = input minus 4 characters (index_main.rs:16-18)
    run(["index", "--input", os.path.join(DATA, "test.gfa"), "--kmer-length", "11", "--output", os.path.join(d, "t11"), "--threads", "2"], d)
    assert os.path.exists(os.path.join(d, "t11.idx"))
    gfa = os.path.join(d, "g.gfa")
    open(gfa, "w").write(open(os.path.join(DATA, "test.gfa")).read())
    run(["index", "-i", gfa, "-k", "5"], d)
    assert os.path.exists(os.path.join(d, "g.idx"))
    a = pkg().HostIndex.load(os.path.join(d, "t11.idx")).arrays()
    b = pkg().HostIndex.build_from_gfa(os.path.join(DATA, "test.gfa"), 11).arrays()
    assert a["kmer_keys"] == b["kmer_keys"] and a["seq_fwd"] == b["seq_fwd"]
    # errors exit with 101 like a Rust panic, and say why
    p = run(["index", "-i", gfa], d, ok=False)
    assert p.returncode == 101 and "kmer-length" in p.stderr
    p = run(["index", "-i", gfa, "-k", "5", "-r", "3"], d, ok=False)
    assert "sampling-rate" in p.stderr
    p = run(["map", "-i", os.path.join(d, "t11"), "-f", os.path.join(DATA, "single-read-test.fa")], d, ok=False)
    assert "poa-aligner" in p.stderr  # required by cli.yml:169-175
    p = run(["map", "-i", os.path.join(d, "t11"), "-f", os.path.join(DATA, "single-read-test.fa"), "-p", "abpoa", "-D"], d, ok=False)
    assert "--graph" in p.stderr  # map.rs:157 unwraps it
    p = run(["bogus"], d, ok=False)
    assert p.returncode == 2 and "USAGE" in p.stderr


def test_map_without_a_gpu_fails_loudly(tmp_path):
    pkg()
    d = str(tmp_path)
    run(["index", "-i", os.path.join(DATA, "test.gfa"), "-k", "11", "-o", os.path.join(d, "t")], d)
    p = subprocess.run([EXE, "map", "-i", os.path.join(d, "t"), "-f", os.path.join(DATA, "single-read-test.fa"), "-p", "abpoa"],
                       cwd=d, capture_output=True, text=True, timeout=600)
    if p.returncode == 0:
        pytest.skip("a GPU is present")
    # no CPU path: the product refuses instead of falling back
    assert p.returncode == 101 and "no MI355X device" in p.stderr


@pytest.mark.gpu
def test_cli_end_to_end_reproduces_golden_gaf(tmp_path):
    pkg()
    d = str(tmp_path)
    want = json.load(open(os.path.join(GOLDEN, "hot_path.json")))
    # config #1: test.gfa + single-read-test.fa -> the placeholder line in both files (default prefix: reads path - 3)
    fa = os.path.join(d, "single-read-test.fa")
    open(fa, "w").write(open(os.path.join(DATA, "single-read-test.fa")).read())
    gfa = os.path.join(DATA, "test.gfa")
    run(["index", "-i", gfa, "-k", "11", "-o", os.path.join(d, "t")], d)
    run(["map", "-i", os.path.join(d, "t"), "-f", fa, "-p", "abpoa", "-D", "-G", gfa], d)
    pre = fa[:-3]
    assert open(pre + "-chains.gaf").read() == want["config1_test_gfa"]["chains_gaf"]
    assert open(pre + "-alignments.gaf").read() == want["config1_test_gfa"]["alignments_gaf"]
    # DRB1, the golden 600 bp reads, explicit prefix, -C prints the alignments
    sys.path.insert(0, GOLDEN)
    import make_golden as mg

    cases = mg.golden_inputs(pkg(), d)
    gfa, k, reads = cases["drb1_600bp_ont"]
    fq = os.path.join(d, "r.fq")
    with open(fq, "w") as f:
        for name, seq in reads:
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, "I" * len(seq)))
    run(["index", "-i", gfa, "-k", str(k), "-o", os.path.join(d, "drb1.idx")], d)
    p = run(["map", "--index", os.path.join(d, "drb1.idx"), "--input-file", fq, "--poa-aligner", "abpoa", "--also-align", "--graph", gfa,
             "--out", os.path.join(d, "o"), "--write-console", "--max-gap-length", "1000", "--chain-min-anchors", "3", "--align-best-n", "1"], d)
    w = want["drb1_600bp_ont"]
    # --also-validate --validation-path (src/map.rs:186-208): one record per alignment
    run(["map", "-i", os.path.join(d, "drb1.idx"), "-f", fq, "-p", "abpoa", "-D", "-G", gfa, "-o", os.path.join(d, "v"), "-v", "-P",
         os.path.join(d, "val.txt")], d)
    hi = pkg().HostIndex.build_from_gfa(gfa, k)
    assert open(os.path.join(d, "val.txt")).read() == hi.validation_records(w["alignments_gaf"], [n for n, _ in reads], [s for _, s in reads])
    p2 = run(["map", "-i", os.path.join(d, "drb1.idx"), "-f", fq, "-p", "abpoa", "-D", "-G", gfa, "-o", os.path.join(d, "v"), "-v"], d, ok=False)
    assert "validation-path" in p2.stderr
    assert open(os.path.join(d, "o-chains.gaf")).read() == w["chains_gaf"]
    assert open(os.path.join(d, "o-alignments.gaf")).read() == w["alignments_gaf"] == p.stdout
    # rspoa is the reference's other backend: not built here, refused by name
    p = run(["map", "-i", os.path.join(d, "drb1.idx"), "-f", fq, "-p", "rspoa", "-D", "-G", gfa, "-o", os.path.join(d, "x")], d, ok=False)
    assert "rspoa" in p.stderr


@pytest.mark.gpu
def test_cli_streams_chunks_over_several_contexts_with_identical_output(tmp_path):
    """`vgaligner map --devices 0,0 --chunk-reads 300`: two contexts (two host threads) on one GPU, every slice streamed in
    batches of 300 reads -- the GAF files must equal the single-batch, single-context run byte for byte (GAF order = read
    order, src/map.rs:123-133,174-184)."""
    p = pkg()
    d = str(tmp_path)
    gfa = os.path.join(DATA, "DRB1-3123.gfa")
    reads = p.readsim.simulate_reads(gfa, 2000, 2500, 0.03, 0.03, 0.04, seed=4242)
    fa = os.path.join(d, "r.fa")
    with open(fa, "w") as f:
        for r in reads:
            f.write(">%s\n%s\n" % (r.name, r.seq))
    run(["index", "-i", gfa, "-k", "11", "-o", os.path.join(d, "drb1")], d)
    base = ["map", "-i", os.path.join(d, "drb1"), "-f", fa, "-p", "abpoa", "-D", "-G", gfa]
    one = run(base + ["-o", os.path.join(d, "one"), "--devices", "0", "--chunk-reads", "0"], d)
    two = run(base + ["-o", os.path.join(d, "two"), "--devices", "0,0", "--chunk-reads", "300"], d)
    assert "1 GPU context(s), 1 batch(es)" in one.stderr and "2 GPU context(s), 8 batch(es)" in two.stderr
    for suffix in ("-chains.gaf", "-alignments.gaf"):
        a, b = open(os.path.join(d, "one" + suffix)).read(), open(os.path.join(d, "two" + suffix)).read()
        assert a == b and a.count("\n") >= 2000
    al = open(os.path.join(d, "two-alignments.gaf")).read().splitlines()
    assert [ln.split("\t")[0] for ln in al] == [r.name for r in reads]
    assert sum(1 for ln in al if ln.split("\t")[5] != "*") >= 1990


@pytest.mark.gpu
def test_cli_two_contexts_on_the_merged_hla_graph_with_full_length_reads(tmp_path, config4_gfa):
    """BASELINE config #4 at reduced count through the product driver: 2 000 full-length (10 kbp or the whole path) reads of the
    19 merged HLA-zoo loci, `--devices 0,0 --chunk-reads 350` (two contexts sharing the GPU, six batches) against one context
    and one batch -- byte-identical GAF files, every read aligned, records in read order (src/map.rs:56-111,162-167)."""
    p = pkg()
    d = str(tmp_path)
    reads = p.readsim.config3_reads(config4_gfa, 2000)
    fa = os.path.join(d, "r.fa")
    p.readsim.write_fasta(reads, fa)
    run(["index", "-i", config4_gfa, "-k", "11", "-o", os.path.join(d, "hla")], d)
    base = ["map", "-i", os.path.join(d, "hla"), "-f", fa, "-p", "abpoa", "-D", "-G", config4_gfa]
    one = run(base + ["-o", os.path.join(d, "one"), "--device", "0", "--chunk-reads", "0"], d)
    two = run(base + ["-o", os.path.join(d, "two"), "--devices", "0,0", "--chunk-reads", "350"], d)
    assert "1 GPU context(s), 1 batch(es)" in one.stderr and "2 GPU context(s), 6 batch(es)" in two.stderr
    for suffix in ("-chains.gaf", "-alignments.gaf"):
        a, b = open(os.path.join(d, "one" + suffix)).read(), open(os.path.join(d, "two" + suffix)).read()
        assert a == b and a.count("\n") >= 2000
    al = open(os.path.join(d, "two-alignments.gaf")).read().splitlines()
    assert [ln.split("\t")[0] for ln in al] == [r.name for r in reads]
    assert sum(1 for ln in al if ln.split("\t")[5] != "*") >= 1990


@pytest.mark.gpu
def test_bench_two_ranks_under_torch_distributed_on_one_gpu():
    """The N > 1 path of bench.py as the driver launches it (python -m torch.distributed.run --nproc-per-node 2 ... bench.py
    --gpus 2), rehearsed on the one GPU of the test box (VGA_BENCH_REHEARSAL=1: ranks share the GPU, gloo instead of RCCL for
    the barrier and the max-over-ranks reduction).  One JSON line from rank 0, n_gpus 2, both ranks' reads in the total."""
    env = dict(os.environ, VGA_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", VGA_POOL_FRACTION="0.4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--reads", "300", "--cpu-sample", "0"]
    pr = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 1
    assert j["whole_job"]["reads"] == 600 and 590 <= j["whole_job"]["aligned"] <= 600
    assert abs(j["value"] - j["whole_job"]["aligned"] / (j["ms_per_step"] * 1e-3)) <= 0.01 * j["value"]


@pytest.mark.gpu
def test_cli_eight_contexts_on_one_gpu_rehearse_an_eight_gpu_node(tmp_path, config4_gfa):
    """Eight device slots as `--devices all` gives them on an 8-GPU node, rehearsed on the one GPU: `vgaligner map --devices
    0,0,0,0,0,0,0,0 --chunk-reads 1500` -- eight contexts and eight host threads, each context told its eighth of the GPU's memory
    (vga_ctx_set_pool_fraction) and of the host's threads (vga_ctx_set_host_threads), chunk pools that start small and grow under
    their keeper threads -- on 12 000 full-length reads of the merged HLA graph, against one context: byte-identical GAF files in
    read order (src/map.rs:56-111,162-167).  Wall times of both runs are printed (pytest -s; round 4 on GPU boxes: 1.5-2.2 s for one
    context, 7.3-10.5 s for eight, box to box -- eight contexts time-slice one GPU, each allocates its own state regions and pool
    segments, which the driver serialises and clears, each uploads the index and runs its own longest problems beside the others'
    bulk launches) and must stay within a factor of 6: what the test guards against is the small-share cliff of round 3 (a
    context with a fifth of the pool ran 40x slower before the keeper thread), not the cost of rehearsing eight GPUs on one."""
    import time
    p = pkg()
    d = str(tmp_path)
    reads = p.readsim.config3_reads(config4_gfa, 12000)
    fa = os.path.join(d, "r.fa")
    p.readsim.write_fasta(reads, fa)
    run(["index", "-i", config4_gfa, "-k", "11", "-o", os.path.join(d, "hla")], d)
    base = ["map", "-i", os.path.join(d, "hla"), "-f", fa, "-p", "abpoa", "-D", "-G", config4_gfa]
    t0 = time.perf_counter()
    one = run(base + ["-o", os.path.join(d, "one"), "--device", "0", "--chunk-reads", "0"], d)
    t1 = time.perf_counter()
    # VGA_POOL_CHECK=1: every chunk of the traceback pool carries its holder, and a chunk that is handed out while held (or
    # comes back from somebody else) fails the call -- the diagnostic that found the state-region flags being cleared late
    # on a busy GPU (round 4's wrong alignments with eight contexts)
    eight = run(base + ["-o", os.path.join(d, "eight"), "--devices", "0,0,0,0,0,0,0,0", "--chunk-reads", "1500"], d, env={"VGA_POOL_CHECK": "1"})
    t2 = time.perf_counter()
    print("one context %.2f s, eight contexts %.2f s" % (t1 - t0, t2 - t1))
    import re
    assert "1 GPU context(s), 1 batch(es)" in one.stderr
    mo = re.search(r"8 GPU context\(s\), (\d+) batch\(es\)", eight.stderr)  # (slices are balanced by bases: some hold more than 1 500 reads)
    assert mo and 8 <= int(mo.group(1)) <= 16, eight.stderr
    for suffix in ("-chains.gaf", "-alignments.gaf"):
        a, b = open(os.path.join(d, "one" + suffix)).read(), open(os.path.join(d, "eight" + suffix)).read()
        assert a == b and a.count("\n") >= 12000
    al = open(os.path.join(d, "eight-alignments.gaf")).read().splitlines()
    assert [ln.split("\t")[0] for ln in al] == [r.name for r in reads]
    assert sum(1 for ln in al if ln.split("\t")[5] != "*") >= 11900
    assert t2 - t1 <= 6.0 * (t1 - t0) + 2.0, (t1 - t0, t2 - t1)


@pytest.mark.gpu
def test_bench_four_ranks_under_torch_distributed_on_one_gpu():
    """bench.py --gpus 4 under torch.distributed.run (the driver's command for N = 4), rehearsed on the one GPU: the launcher is
    started before anything touches the GPU, VGA_BENCH_REHEARSAL=1 lets the four ranks share it (gloo for the barrier and the
    max-over-ranks reduction).  One JSON line, n_gpus 4, the four ranks' reads summed in whole_job, value = that sum over the
    slowest rank's time."""
    env = dict(os.environ, VGA_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1", VGA_POOL_FRACTION="0.2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--reads", "200", "--cpu-sample", "0"]
    pr = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert pr.returncode == 0, pr.stderr[-2000:]
    lines = [ln for ln in pr.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 4 and j["scaling"] == "weak" and j["steps"] == 1
    assert j["whole_job"]["reads"] == 800 and j["whole_job"]["ranks"] == 4 and 790 <= j["whole_job"]["aligned"] <= 800
    assert abs(j["value"] - j["whole_job"]["aligned"] / (j["ms_per_step"] * 1e-3)) <= 0.01 * j["value"]


@pytest.mark.gpu
def test_cli_config5_streams_several_chunks_of_the_1_mbp_pangenome(tmp_path):
    """BASELINE config 5 through the product CLI at a size that takes several chunks (60 000 reads x 10 kbp on the 1 Mbp synthetic
    pangenome: 4.9 GB of GAF; tests/prof_million_reads.py is the same harness at the full 1 000 000 reads -- 25 s, 81 GB): one
    alignment record per read in read order, every read aligned, and the records of the first 2 000 reads byte-identical to a run of
    those reads alone (src/map.rs:56-111,162-167,219-226)."""
    pkg()
    import prof_million_reads as pm
    d = str(tmp_path)
    res = pm.run(60000, os.path.join(d, "out"), os.path.join(d, "work"), parts=8, quiet=True)
    print(res)
    assert res["alignment_records"] == 60000 and res["aligned"] == 60000
    assert res["first_2000_records_identical_to_a_run_of_their_own"]
    assert any("batch(es)" in ln and " 1 batch" not in ln for ln in res["stderr_tail"]), res["stderr_tail"]

