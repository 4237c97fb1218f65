"""Small driver used under rocprofv3: one map+align pass over N config-3 reads (no oracle, no torch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    rl = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
    p = ge.load_package()
    gfa = os.path.join(ge.ROOT, "tests", "golden", "data", "DRB1-3123.gfa")
    reads = p.readsim.simulate_reads(gfa, n, rl, 0.03, 0.03, 0.04, seed=77)
    hi = p.HostIndex.build_from_gfa(gfa, 11)
    ctx = p.Context(0)
    hi.upload(ctx)
    b = ctx.batch([r.seq for r in reads])
    for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 1):
        out = b.map_align_raw()
    print({k: v for k, v in out.items() if k != "kernels"})
    for k in out["kernels"]:
        print(k)

if __name__ == "__main__":
    main()
