"""diagnostic: config #5 (1 Mbp synthetic pangenome) through map + align, with VGA_TRACE output"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
p = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
gfa = "/tmp/syn1m.gfa"
print(p.readsim.synth_pangenome(gfa, int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000))
t = time.time(); hi = p.HostIndex.build_from_gfa(gfa, 11); print("index build s", time.time() - t)
ctx = p.Context(0); hi.upload(ctx)
reads = p.readsim.config3_reads(gfa, n)
seqs = [r.seq for r in reads]
b = ctx.batch(seqs)
t = time.time(); mo = b.map(); print("map s", time.time() - t, "anchors/read", mo.n_anchors / n)
arr = hi.arrays()
starts = arr["node_seq_idx"]
for r in range(min(n, 8)):
    a0 = int(mo.anchor_off[r])
    for ph, ch in mo.chains_of(r)[:2]:
        if ph: print(r, "placeholder"); continue
        tb = mo.target_begin[a0:][ch]; print(r, "chain", len(ch), "target span", int(tb.min()), int(tb.max()), "read from", reads[r].offset)
t = time.time(); al = b.align(mo); print("align s", time.time() - t, "aligned", int(al.aligned.sum()), "cells", al.poa_cells, "rows", al.poa_rows)
