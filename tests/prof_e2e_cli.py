"""End-to-end wall clock of the product CLI on config 3 (FASTA in -> GAF files out), for DESIGN.md section 5:
    python tests/prof_e2e_cli.py [n_reads] [extra vgaligner map flags ...]
E2E_WORKLOAD=config4 runs the merged, sorted HLA graph (19 loci) instead of DRB1-3123, config5 the 1 Mbp synthetic pangenome.
Not the bench line: it includes reading the FASTA, the index load + upload, GAF text generation and file output."""
import json, os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

p = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
extra = sys.argv[2:]
gfa = os.path.join(ROOT, "tests", "golden", "data", "DRB1-3123.gfa")
exe = os.path.join(ROOT, "rs-vgaligner_amd", "vgaligner")
with tempfile.TemporaryDirectory(dir="/tmp") as d:
    if os.environ.get("E2E_WORKLOAD") == "config4":
        gfa = os.path.join(d, "hla19.gfa")
        p.readsim.config4_graph(os.path.join(ROOT, "tests", "golden", "data"), gfa)
    if os.environ.get("E2E_WORKLOAD") == "config5":
        gfa = os.path.join(d, "syn1m.gfa")
        p.readsim.synth_pangenome(gfa, 1000000, seed=77)
    reads = p.readsim.config3_reads(gfa, n)
    fa = os.path.join(d, "reads.fa")
    p.readsim.write_fasta(reads, fa)
    t0 = time.perf_counter()
    subprocess.check_call([exe, "index", "-i", gfa, "-k", "11", "-o", os.path.join(d, "drb1")])
    t1 = time.perf_counter()
    # (stderr is read line by line so that the trace marks "start" and "done" -- VGA_TRACE=1 -- get a time stamp of ours:
    # what lies before the first and after the second is process start and exit)
    no_align = os.environ.get("E2E_NO_ALIGN") == "1"  # (chains only: what the process costs without the alignment pass)
    cmd = [exe, "map", "-i", os.path.join(d, "drb1"), "-f", fa, "-p", "abpoa"] + ([] if no_align else ["-D", "-G", gfa]) + ["-o", os.path.join(d, "out")] + extra
    pr = subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    err_lines, t_start_mark, t_done_mark = [], None, None
    for ln in pr.stderr:
        now = time.perf_counter()
        if "[vgh-trace] start " in ln:
            t_start_mark = now
        if "[vgh-trace] done " in ln:
            t_done_mark = now
        err_lines.append(ln)
    rc = pr.wait()
    t2 = time.perf_counter()
    stderr_text = "".join(err_lines)
    sys.stderr.write(stderr_text)
    assert rc == 0, stderr_text

    class r:  # (what the summary below reads)
        stderr = stderr_text
    al = os.path.join(d, "out-alignments.gaf")
    aligned = n if no_align else sum(1 for ln in open(al) if ln.split("\t")[5] != "*")
    print(json.dumps({"workload": os.environ.get("E2E_WORKLOAD", "config3"), "reads": n, "aligned": aligned, "index_s": round(t1 - t0, 2), "map_s": round(t2 - t1, 2),
                      "aligned_reads_per_s_end_to_end": round(aligned / (t2 - t1), 1),
                      "before_main_s": None if t_start_mark is None else round(t_start_mark - t1, 3),
                      "after_done_s": None if t_done_mark is None else round(t2 - t_done_mark, 3),
                      "chains_gaf_mb": round(os.path.getsize(os.path.join(d, "out-chains.gaf")) / 1e6, 1),
                      "alignments_gaf_mb": 0 if no_align else round(os.path.getsize(al) / 1e6, 1), "flags": extra, "stderr_tail": r.stderr.strip().splitlines()[-4:]}))
