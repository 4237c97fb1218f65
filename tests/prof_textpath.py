"""GAF text path replay (GPU box): one chunk of a workload is mapped and aligned once, then its GAF text is generated and appended
to the two files K times (vgh::textpath_replay) -- GB/s of GAF text through text threads + file appends, with no GPU work inside.

    python3 tests/prof_textpath.py [config3|config4|config5] [reads] [repeat] [threads ...]
"""
import json, os, shutil, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

pkg = ge.load_package()
wl = sys.argv[1] if len(sys.argv) > 1 else "config5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 4
threads = [int(x) for x in sys.argv[4:]] or [0]
tmp = tempfile.mkdtemp(prefix="vga_textpath_")
gfa = os.path.join(ROOT, "tests", "golden", "data", "DRB1-3123.gfa")
if wl == "config4":
    gfa = os.path.join(tmp, "c4.gfa"); pkg.readsim.config4_graph(os.path.join(ROOT, "tests", "golden", "data"), gfa)
elif wl == "config5":
    gfa = os.path.join(tmp, "c5.gfa"); pkg.readsim.synth_pangenome(gfa)
reads = pkg.readsim.simulate_reads(gfa, n, 10000, 0.03, 0.03, 0.04, seed=77)
hidx = pkg.HostIndex.build_from_gfa(gfa, 11)
ctx = pkg.Context(0)
hidx.upload(ctx)
for t in threads:
    r = hidx.textpath_replay(ctx, [x.name for x in reads], [x.seq for x in reads], os.path.join(tmp, "out%d" % t), repeat, t)
    r.update({"workload": wl, "reads": n, "repeat": repeat, "dir": tmp, "fs": os.popen("df -T %s | tail -1" % tmp).read().split()[1:2]})
    print("TEXTPATH " + json.dumps(r))
shutil.rmtree(tmp, ignore_errors=True)
