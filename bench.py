#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X vgaligner hot path (map -> chain -> align).

Workload (BASELINE.json configs[2], the one the metric is quoted on): HLA DRB1-3123 graph, k=11,
10 000 synthetic 10 kbp ONT-error-profile reads (3 % sub / 3 % ins / 4 % del, PCG64 seed 77),
`--also-align`.  One "step" = one pass of the whole hot path over the batch: k-mer probe, anchor
sort, chaining DP + backtracking, subgraph extraction, banded POA DP, traceback, result hand-over.
Reads are resident in HBM before the timed region starts.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Rank 0 prints ONE JSON line.  Multi-GPU: reads shard embarrassingly (every rank maps its own batch of
the same size against a replicated index; no data-path collective) -> "scaling": "weak".
The cpu_baseline leg (rank 0, N=1 only) times the single-threaded CPU oracle on a bounded sample of the
same workload; it is a reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)
GFA = os.path.join(ROOT, "tests", "golden", "data", "DRB1-3123.gfa")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=10000, help="reads per GPU (default: the full config #3 batch)")
    ap.add_argument("--read-len", type=int, default=10000)
    ap.add_argument("--cpu-sample", type=int, default=40, help="reads of the workload timed on the CPU oracle (0 = skip)")
    ap.add_argument("--workload", choices=["config2", "config3", "config4", "config5"], default="config3",
                    help="config3 (default, the headline line): DRB1-3123; config4: the 19 merged, sorted HLA-zoo loci; config5: 1 Mbp synthetic "
                         "pangenome (same read model); config2: DRB1-3123, 150 bp reads with 1 %% substitutions, map-only "
                         "(anchor + chain kernels; use --reads 1000 for BASELINE's size).  Extra measurements, not the driver's line")
    ap.add_argument("--remain-rule", type=int, choices=[0, 1], default=None,
                    help="vga_poa_params.remain_rule: 0 longest path, 1 first out-edge (default: the library's default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the library has no CPU path)")
    # VGA_BENCH_REHEARSAL=1 (testing the N > 1 code path on a box with fewer GPUs than ranks): ranks share the GPUs that
    # exist and the timing reduction runs over gloo.  Never set by the driver; the numbers of such a run mean nothing.
    rehearsal = os.environ.get("VGA_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    # host threads of the library (subgraph extraction, CIGAR strings): share the node's cores between the ranks
    if world > 1 and "VGA_HOST_THREADS" not in os.environ:
        os.environ["VGA_HOST_THREADS"] = str(max(4, min(32, (os.cpu_count() or 32) // world)))

    os.environ.setdefault("VGA_TUNE_MALLOC", "1")  # opt in: result arrays stay in the heap between calls (vga_ctx_create)
    import __graft_entry__ as ge

    pkg = ge.load_package()

    # ---- workload: identical generator on every rank, different seed per rank (different reads, same shape)
    t0 = time.time()
    GFA = globals()["GFA"]
    wl_name = "config3: HLA DRB1-3123 graph"
    map_only = args.workload == "config2"
    if map_only:
        wl_name = "config2: HLA DRB1-3123 graph, map-only"
        if args.read_len == 10000:
            args.read_len = 150
    elif args.workload != "config3":
        import tempfile

        GFA = os.path.join(tempfile.mkdtemp(prefix="vga_bench_"), args.workload + ".gfa")
        if args.workload == "config4":
            nn, ne, nb_ = pkg.readsim.config4_graph(os.path.join(ROOT, "tests", "golden", "data"), GFA)
            wl_name = "config4: 19 of the 20 HLA-zoo loci (7-MICB-4277 left out: cyclic after sorting, the reference k-mer enumeration does not terminate on it), sorted (readsim.toposort_gfa) and merged (%d nodes, %d bp)" % (nn, nb_)
        else:
            nn, ne, nb_ = pkg.readsim.synth_pangenome(GFA)
            wl_name = "config5: synthetic pangenome (%d nodes, %d bp)" % (nn, nb_)
    if map_only:
        reads = pkg.readsim.simulate_reads(GFA, args.reads, args.read_len, 0.01, 0.0, 0.0, seed=pkg.sharding.bench_seed(rank))
    else:
        reads = pkg.readsim.simulate_reads(GFA, args.reads, args.read_len, 0.03, 0.03, 0.04, seed=pkg.sharding.bench_seed(rank))
    seqs = [r.seq for r in reads]
    t_gen = time.time() - t0

    # ---- index: built by the product's own C++ host, uploaded once
    hidx = pkg.HostIndex.build_from_gfa(GFA, 11)
    ctx = pkg.Context(local_rank)
    hidx.upload(ctx)
    batch = ctx.batch(seqs)  # reads are now resident in HBM

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    last = None
    poa_params = pkg.default_poa_params()
    if args.remain_rule is None:
        args.remain_rule = int(poa_params.remain_rule)
    poa_params.remain_rule = args.remain_rule
    if os.environ.get("VGA_BENCH_WF"):  # (diagnostics only: another band -- not the workload BASELINE.json names; the line says so)
        poa_params.wf = float(os.environ["VGA_BENCH_WF"])
    step = batch.map_raw if map_only else (lambda: batch.map_align_raw(poa_params=poa_params))
    for _ in range(args.warmup):
        last = step()
    barrier()
    t_start = time.perf_counter()
    kern = {}
    for _ in range(args.steps):
        last = step()
        for k in last["kernels"]:
            e = kern.setdefault(k["name"], {"ms": 0.0, "launches": 0, "bytes": 0, "busy_ms": 0.0})
            e["ms"] += k["ms"]
            e["busy_ms"] += k["busy_ms"]
            e["launches"] += k["launches"]
            e["bytes"] += k["algorithmic_bytes"]
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed, aligned_all, reads_all = pkg.sharding.reduce_timing(elapsed, last["aligned"], last["n_reads"], world,
                                                                 device="cuda" if world > 1 and not rehearsal else None)

    # the same batch under the OTHER remain rule (vga_poa_params.remain_rule: the open choice of the POA restatement with first-order
    # effects on cost), one untimed-region step on rank 0: reported beside the line's value, never part of it
    other_rule = None
    if rank == 0 and world == 1 and not map_only and not os.environ.get("VGA_BENCH_NO_OTHER_RULE"):
        try:
            pp2 = pkg.default_poa_params()
            pp2.remain_rule = 1 - args.remain_rule
            batch.map_align_raw(poa_params=pp2)  # (warm-up: the footprint scale adapts)
            t_o = time.perf_counter()
            lo = batch.map_align_raw(poa_params=pp2)
            dt_o = time.perf_counter() - t_o
            other_rule = {"poa_remain_rule": ["longest-path", "first-out-edge"][1 - args.remain_rule], "value": round(lo["aligned"] / dt_o, 2),
                          "unit": "aligned reads/s", "poa_cells": lo["poa_cells"], "steps": 1,
                          "note": "same reads, one step after the timed region; which rule the reference's abPOA computes is unverified (DESIGN.md section 2)"}
        except Exception as e:  # noqa: BLE001
            other_rule = {"error": str(e)[:200]}

    # measured device-to-device copy rate next to the nominal HBM peak (read + write bytes of 1 GiB copies).  After the timed
    # region: torch allocations made before it leave less HBM for the library's traceback pool (-10 % on the step)
    copy_gbs = None
    if rank == 0 and not os.environ.get("VGA_BENCH_NO_COPY"):
        try:
            n_el = 1 << 28  # 2 x 1 GiB of int32
            a = torch.empty(n_el, dtype=torch.int32, device="cuda").random_(0, 1000)
            b = torch.empty_like(a)
            b.copy_(a)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record()
            torch.cuda.synchronize()
            copy_gbs = round(10 * 2 * n_el * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del a, b
            torch.cuda.empty_cache()
        except Exception:
            copy_gbs = None


    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = aligned_all * args.steps / elapsed
    # ---- roofline of the dominant kernel (live hipEvent timing on the library's own stream)
    dom = max((k for k in kern if k.endswith("_dp") or k.endswith("traceback") or k.startswith("kmer") or k.startswith("anchor")),
              key=lambda k: kern[k]["ms"])
    d = kern[dom]
    # Two POA sub-batches are in flight at a time (two streams), so a launch's own duration -- what rocprofv3 --stats
    # lists, `avg_launch_ms` below -- includes the time it shares the GPU with its neighbour.  The roofline prices the
    # kernel on the wall time during which at least one of its launches was executing (`busy_ms`, the union of the
    # launch intervals from the same hipEvents): achieved = algorithmic bytes of all launches / busy time.
    avg_ms = d["ms"] / max(d["launches"], 1)
    busy_per_launch = d["busy_ms"] / max(d["launches"], 1)
    launches_per_step = max(d["launches"] / args.steps, 1)
    impl_bytes_per_launch = d["bytes"] / max(d["launches"], 1)  # the library's own byte model of the kernel
    formula = None
    operands = None
    if dom == "poa_band_dp" and last.get("poa_cells"):
        # SURVEY.md section 8(d):  B_poa = N + L + 4 C + 4 (N_path + L) + P  per step, every operand from this run's counters
        # (N graph bases = DP rows, L query bases, C band cells, N_path graph bases on the alignment paths, P CIGAR bytes)
        operands = {"N": last["poa_rows"], "L": sum(len(s) for s in seqs), "C": last["poa_cells"], "N_path": last["path_bases"],
                    "P": last["cigar_bytes"], "C_v": last["poa_value_cells"]}
        formula = "N + L + 4*C + 4*(N_path + L) + P   (SURVEY.md 8d, per step; / launches_per_step = per launch)"
        bytes_per_step = operands["N"] + operands["L"] + 4 * operands["C"] + 4 * (operands["N_path"] + operands["L"]) + operands["P"]
        bytes_per_launch = bytes_per_step / launches_per_step
    else:
        bytes_per_launch = impl_bytes_per_launch
    achieved = bytes_per_launch / (busy_per_launch * 1e-3) / 1e9 if busy_per_launch > 0 else 0.0
    traffic = None
    traffic_src = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if (tj.get("kernel") == dom and tj.get("reads") == args.reads and tj.get("read_len") == args.read_len
                    and args.workload == "config3" and tj.get("remain_rule", 0) == args.remain_rule):
                # measured with rocprofv3 PMC passes (profiles/traffic.json), per step; per launch = / launches per step
                traffic = int(tj["hbm_bytes_per_step"] / launches_per_step)
                traffic_src = "committed profile (profiles/traffic.json: %s), not measured in this run" % tj.get("tag", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "frac_8d": round(achieved / HBM_PEAK_GBS, 5) if formula else None,
                "algorithmic_bytes_per_launch": int(bytes_per_launch), "formula": formula, "operands_per_step": operands,
                "launches_per_step": launches_per_step, "busy_ms_per_launch": round(busy_per_launch, 3),
                "timing": "live: hipEvents on the library's launch streams; busy = union of the launch intervals (launches overlap)",
                "avg_launch_ms": round(avg_ms, 3), "concurrency": round(d["ms"] / d["busy_ms"], 3) if d["busy_ms"] > 0 else None,
                "launches": d["launches"],
                "peak_measured_copy": copy_gbs, "frac_of_measured_copy": round(achieved / copy_gbs, 5) if copy_gbs else None}
    if operands:
        # the same kernel priced on this implementation's own traffic model (1 direction byte per cell instead of 8d's int32,
        # 6-byte value rows written once and read back once) and on the compulsory part of it alone
        comp = operands["N"] + operands["L"] + operands["C"]
        impl_gbs = impl_bytes_per_launch / (busy_per_launch * 1e-3) / 1e9 if busy_per_launch > 0 else 0.0
        comp_gbs = comp / launches_per_step / (busy_per_launch * 1e-3) / 1e9 if busy_per_launch > 0 else 0.0
        roofline["frac_impl"] = round(impl_gbs / HBM_PEAK_GBS, 5)
        roofline["impl"] = {"formula": "N + L + C + 12*C_v + 6*(alignment columns)", "bytes_per_launch": int(impl_bytes_per_launch),
                            "achieved": round(impl_gbs, 2), "frac": round(impl_gbs / HBM_PEAK_GBS, 5)}
        roofline["compulsory"] = {"formula": "N + L + C", "bytes_per_launch": int(comp / launches_per_step), "achieved": round(comp_gbs, 2),
                                  "frac": round(comp_gbs / HBM_PEAK_GBS, 5)}
        # the ceiling the kernel is actually under: instruction issue.  Counts per band cell come from the committed PMC run
        # (profiles/instr.json: SQ_INSTS_VALU / SQ_INSTS_SALU of one bench step), the sustainable rate from the microbenchmark
        # (profiles/r02_valu_issue_microbench.txt: ~0.58 T wave64 VALU instructions/s chip-wide for this instruction mix)
        ip = os.path.join(ROOT, "profiles", "instr.json")
        if os.path.exists(ip) and args.workload == "config3":
            try:
                ij = json.load(open(ip))
                if ij.get("remain_rule", 0) == args.remain_rule:
                    valu = ij["valu_wave_instr_per_64_cells"] * operands["C"] / 64.0
                    rate = valu / (d["busy_ms"] / args.steps * 1e-3)
                    roofline["valu"] = {"bound": "valu_issue", "achieved": round(rate / 1e9, 1), "peak": ij["peak_Gwaveinst_per_s"], "unit": "G wave64 instr/s",
                                        "frac": round(rate / 1e9 / ij["peak_Gwaveinst_per_s"], 4), "valu_per_64_cells": ij["valu_wave_instr_per_64_cells"],
                                        "salu_per_64_cells": ij.get("salu_wave_instr_per_64_cells"),
                                        "source": "committed profile (profiles/instr.json), instruction counts not measured in this run; cells and busy time are live. " + str(ij.get("source"))}
            except Exception:
                pass

    # ---- CPU baseline: the oracle on a bounded sample of the same reads (N=1 only), run as a child process that never
    # touches the GPU: one core (the reference is single-threaded), and the same code over all host cores
    cpu = None
    cpu_all = None
    cpu_faithful = None
    if world == 1 and args.cpu_sample > 0 and not map_only:
        import subprocess
        import tempfile

        def run_oracle(ns, nproc, faithful=False):
            with tempfile.NamedTemporaryFile("w", suffix=".fa", delete=False) as f:
                for r in reads[:ns]:
                    f.write(">%s\n%s\n" % (r.name, r.seq))
                path = f.name
            try:
                p = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_bench.py"), GFA, "11", path, str(nproc)] + (["faithful"] if faithful else []),
                                   capture_output=True, text=True, timeout=900)
                if p.returncode != 0:
                    raise RuntimeError(p.stderr[-400:])
                return json.loads(p.stdout.strip().splitlines()[-1])
            finally:
                os.unlink(path)

        ns = min(args.cpu_sample, len(seqs))
        r1 = run_oracle(ns, 1)
        cpu = {"value": round(r1["aligned"] / r1["max_worker_s"], 4), "unit": "aligned reads/s", "cores": 1, "kind": "port",
               "sample": f"first {ns} reads of the workload, oracle/libvga_oracle.so (O(log n) k-mer lookup, no debug printing)",
               "seconds": round(r1["max_worker_s"], 2)}
        # B1 of BASELINE.md section 3: the same port with the reference's cost shape (linear membership scan per query k-mer,
        # src/index.rs:319; bit-by-bit rank / select, 427-480; whole-sequence clone per node lookup, 516-519), no debug printing
        nf = min(len(seqs), max(8, ns // 4))
        rf = run_oracle(nf, 1, faithful=True)
        cpu_faithful = {"value": round(rf["aligned"] / rf["max_worker_s"], 4), "unit": "aligned reads/s", "cores": 1, "kind": "port",
                        "sample": f"first {nf} reads of the workload, oracle with og_set_reference_faithful_costs(1)", "seconds": round(rf["max_worker_s"], 2)}
        ncore = max(1, min(os.cpu_count() or 1, 16))  # a GPU box gives one GPU a 16-core share
        nsa = min(len(seqs), max(ns, 4 * ncore))
        ra = run_oracle(nsa, ncore)
        cpu_all = {"value": round(ra["aligned"] / ra["wall_s"], 3), "unit": "aligned reads/s", "cores": ra["procs"], "kind": "port",
                   "sample": f"first {nsa} reads of the workload, one oracle process per core (index build included in the wall time)",
                   "seconds": round(ra["wall_s"], 2)}

    out = {
        "metric": "mapped reads/sec (150 bp vs HLA graph, anchors + chains)" if map_only else "aligned reads/sec (10 kbp ONT vs HLA graph)",
        "value": round(value, 2),
        "unit": "mapped reads/s" if map_only else "aligned reads/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int32+f64",
        "data": "synthetic",
        "config": {"workload": "%s, k=11, %d x %d bp %s reads per GPU%s"
                               % (wl_name, args.reads, args.read_len, "1 %-substitution" if map_only else "ONT-profile", "" if map_only else ", --also-align"),
                   "reads_per_gpu": args.reads, "read_len": args.read_len, "sharding": "reads, replicated index, no collective",
                   "poa_remain_rule": ["longest-path", "first-out-edge"][args.remain_rule],
                   **({"diagnostic_band_wf": float(os.environ["VGA_BENCH_WF"])} if os.environ.get("VGA_BENCH_WF") and not map_only else {})},
        "roofline": roofline,
        "other_remain_rule": other_rule,
        "cpu_baseline": cpu,
        "cpu_baseline_faithful": cpu_faithful,
        "cpu_baseline_all_cores": cpu_all,
        "reads_per_s": round(reads_all * args.steps / elapsed, 2),
        "whole_job": {"reads": int(reads_all), "aligned": int(aligned_all), "ranks": world},  # per step, summed over the ranks
        "per_step": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in last.items() if k != "kernels"},
        # sum of every launch's own duration: launches of poa_band_dp overlap (three in flight), so this is NOT time per step
        "kernels_summed_launch_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in kern.items()},
        "kernels_busy_ms_per_step": {k: round(v["busy_ms"] / args.steps, 3) for k, v in kern.items()},
        # algorithmic bytes (per kernel) over the kernel's busy time
        "kernels_algorithmic_gbs": {k: round(v["bytes"] / (v["busy_ms"] * 1e-3) / 1e9, 1) for k, v in kern.items() if v["busy_ms"] > 0 and v["bytes"]},
        "gen_s": round(t_gen, 1),
    }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
