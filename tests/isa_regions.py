"""Static instruction counts of k_poa_dp_t4<NT, true> between the POA_MARK region markers (diagnostics only).

    python3 tests/isa_regions.py [NT]
"""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nt = sys.argv[1] if len(sys.argv) > 1 else "256"
src = os.path.join(ROOT, "rs-vgaligner_amd", "csrc", "vga_poa.hip")
out = "/tmp/vga_poa_marked.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-DPOA_MARKERS",
                       "-S", "--cuda-device-only", "-I", os.path.dirname(src), src, "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
# the default instantiation: compile-time penalties, 4 columns per lane, 32-bit row state
start = next(i for i, l in enumerate(lines) if l.startswith("_Z11k_poa_dp_t4ILi%sELb1EE" % nt))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
region = "prologue"
cnt = collections.OrderedDict()
for l in lines[start:end]:
    t = l.strip()
    m = re.match(r"; MARK (\w+)", t)
    if m:
        region = m.group(1)
        continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    c = cnt.setdefault(region, collections.Counter())
    op = t.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    c[kind] += 1
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        c["spill"] += 1
for k, c in cnt.items():
    print("%-14s valu %4d (spill %3d)  salu %4d  lds %3d  vmem %3d" % (k, c["valu"], c["spill"], c["salu"], c["lds"], c["vmem"]))

if len(sys.argv) > 2:  # list the spill instructions of one region
    region = "prologue"
    for l in lines[start:end]:
        t = l.strip()
        m = re.match(r"; MARK (\w+)", t)
        if m:
            region = m.group(1)
            continue
        if region == sys.argv[2] and (t.startswith("v_readlane") or t.startswith("v_writelane")):
            print("   ", t)
