"""diagnostic: several contexts in one process (create / use / close / create again)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
p = ge.load_package()
gfa = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data", "DRB1-3123.gfa")
hi = p.HostIndex.build_from_gfa(gfa, 11)
seqs = [r.seq for r in p.readsim.config3_reads(gfa, 8, 2000)]
def run(tag):
    ctx = p.Context(0); hi.upload(ctx)
    b = ctx.batch(seqs); al = b.align(b.map()); print(tag, int(al.aligned.sum()), flush=True)
    return ctx
a = run("A")
b = run("B"); print("closing B", flush=True); sys.stderr.flush(); b.close(); print("closed B", flush=True)
c = run("C")
