"""Row mix of one config-3 POA problem (diagnostics, GPU box): which rows are simple / hot / multi-predecessor / far / wide,
band widths, wave-steps per row at NT = 256.  Uses VGA_POA_DUMP_ROWS (row records of the launch's first problem)."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dump = "/tmp/vga_rows.txt"
os.environ["VGA_POA_DUMP_ROWS"] = dump
import __graft_entry__ as ge

p = ge.load_package()
gfa = os.path.join(ROOT, "tests", "golden", "data", "DRB1-3123.gfa")
nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 8
tot = collections.Counter()
for seed in range(nreads):
    reads = p.readsim.simulate_reads(gfa, 1, 10000, 0.03, 0.03, 0.04, seed=1000 + seed)
    hidx = p.HostIndex.build_from_gfa(gfa, 11)
    ctx = p.Context(0)
    hidx.upload(ctx)
    b = ctx.batch([r.seq for r in reads])
    al = b.align(b.map())
    rows = [tuple(int(x) for x in l.split()) for l in open(dump)]
    prev = None
    for (r, beg, end, lmax, rmax, pred, npred, voff) in rows:
        W = ((end - (beg & ~3) + 1 + 3) & ~3)
        wide = W + 8 > 4096
        if r == 0:
            prev = (beg, end, wide)
            continue
        first = npred != 0
        simple = (not prev[2]) and ((not first) or (npred == 1 and pred == r - 1))
        hot = simple and not wide and end <= prev[1] + 1
        tot["rows"] += 1
        tot["cells"] += end - beg + 1
        tot["simple"] += simple
        tot["hot"] += hot
        tot["hot_cells"] += (end - beg + 1) if hot else 0
        tot["multi"] += npred > 1
        tot["far_single"] += first and npred == 1 and pred != r - 1
        tot["wide"] += wide
        tot["below_wide"] += prev[2]
        tot["jump_right"] += simple and not wide and end > prev[1] + 1
        tot["first"] += first
        ws = (W + 255) // 256
        tot["wave_steps"] += ws
        tot["steps"] += (W + 1023) // 1024
        tot["ws_cells_capacity"] += ws * 256
        prev = (beg, end, wide)
    ctx = None
n = tot["rows"]
print("rows %d cells %d mean width %.0f" % (n, tot["cells"], tot["cells"] / n))
for k in ("simple", "hot", "first", "multi", "far_single", "wide", "below_wide", "jump_right"):
    print("  %-12s %6.2f %% of rows" % (k, 100.0 * tot[k] / n))
print("  hot cells %.2f %% of cells; steps per row %.2f; wave-steps per row %.2f; cells per wave-step %.1f (of 256)" % (
    100.0 * tot["hot_cells"] / tot["cells"], tot["steps"] / n, tot["wave_steps"] / n, tot["cells"] / tot["wave_steps"]))
