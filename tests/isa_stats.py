"""Static ISA statistics of the POA DP kernels (diagnostics only): instruction counts by kind, spill moves
(v_readlane / v_writelane), v_readfirstlane, registers, scratch, and -- with -DPOA_MARKERS -- counts per marked region.

    python3 tests/isa_stats.py [kernel-prefix ...]      e.g.  python3 tests/isa_stats.py k_poa_dp_t5ILi256ELb1 k_poa_dp_t4ILi256ELb1
"""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "rs-vgaligner_amd", "csrc", "vga_poa.hip")
out = "/tmp/vga_poa_marked.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-DPOA_MARKERS",
                       "-S", "--cuda-device-only", "-I", os.path.dirname(src), src, "-o", out], stderr=subprocess.DEVNULL)
text = open(out).read()
lines = text.split("\n")
want = sys.argv[1:] or ["k_poa_dp_t5ILi256ELb1", "k_poa_dp_t4ILi256ELb1"]


def kind(op):
    return "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"


for w in want:
    start = next(i for i, l in enumerate(lines) if re.match(r"_Z\d+" + re.escape(w) + r".*:", l))
    name = lines[start].split(":")[0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    k = text.index(".amdhsa_kernel " + name)
    md = text[k:text.index(".end_amdhsa_kernel", k)]
    g = lambda key: re.search(key + r"\s+(\S+)", md).group(1)
    region = "prologue"
    cnt = collections.OrderedDict()
    tot = collections.Counter()
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"; MARK (\w+)", t)
        if m:
            region = m.group(1)
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        for c in (cnt.setdefault(region, collections.Counter()), tot):
            c[kind(op)] += 1
            if op.startswith("v_readlane") or op.startswith("v_writelane"):
                c["spill"] += 1
            if op.startswith("v_readfirstlane"):
                c["rfl"] += 1
            if op.startswith("scratch_"):
                c["scratch"] += 1
    print("%s: vgpr %s sgpr %s scratch %s B | valu %d (lane moves %d, readfirstlane %d) salu %d lds %d vmem %d scratch ops %d" % (
        w, g(".amdhsa_next_free_vgpr"), g(".amdhsa_next_free_sgpr"), g(".amdhsa_private_segment_fixed_size"), tot["valu"], tot["spill"],
        tot["rfl"], tot["salu"], tot["lds"], tot["vmem"], tot["scratch"]))
    for r, c in cnt.items():
        print("    %-14s valu %4d (lane moves %3d, rfl %3d)  salu %4d  lds %3d  vmem %3d" % (r, c["valu"], c["spill"], c["rfl"], c["salu"], c["lds"], c["vmem"]))
