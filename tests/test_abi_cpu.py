"""CPU-side checks of the product library: it loads, exports every symbol include/vga_hip.h declares,
and refuses to run without a GPU (no silent fallback).  No compute calls."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT, pkg


def test_library_builds_and_exports_every_declared_symbol():
    import __graft_entry__ as ge

    ge.build()
    p = pkg()
    L = p.load_library()
    header = open(os.path.join(ROOT, "include", "vga_hip.h")).read()
    declared = set(re.findall(r"\b(vga_[a-z_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(p.binding.ABI_SYMBOLS), declared ^ set(p.binding.ABI_SYMBOLS)
    for sym in sorted(declared):
        assert hasattr(L, sym), f"libvga_hip.so does not export {sym}"
    assert L.vga_abi_version() == 6


# every struct of include/vga_hip.h and the ctypes class binding.py declares for it
ABI_STRUCTS = {
    "vga_kmerpos": "KmerPos", "vga_index_desc": "IndexDesc", "vga_map_params": "MapParams", "vga_map_result": "MapResult",
    "vga_poa_params": "PoaParams", "vga_poa_result": "PoaResult", "vga_align_result": "AlignResult", "vga_kernel_time": "KernelTime", "vga_chain_text": "ChainText",
}


def _header_structs():
    """{struct name: [field names in declaration order]} parsed from include/vga_hip.h."""
    header = open(os.path.join(ROOT, "include", "vga_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(vga_[a-z_]+)\s*;", header, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.findall(r"([A-Za-z_][A-Za-z_0-9]*)\s*$", part.strip())[0])
        out[name] = fields
    return out


def test_header_is_plain_c99_and_layouts_match_ctypes(tmp_path):
    """Compiles a C99 program against include/vga_hip.h (gcc -std=c99 -Wall -Werror -pedantic), lets it print sizeof of every
    struct and offsetof of every field, and compares with the ctypes classes of binding.py field by field: a silent
    reorder or a changed type on either side fails here."""
    import subprocess

    structs = _header_structs()
    assert set(structs) == set(ABI_STRUCTS), set(structs) ^ set(ABI_STRUCTS)
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "vga_hip.h"', "int main(void) {"]
    for name, fields in structs.items():
        src.append('  printf("S %s %%zu\\n", sizeof(%s));' % (name, name))
        for f in fields:
            src.append('  printf("F %s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));' % (name, f, name, f, name, f))
    src += ['  printf("V %d\\n", VGA_OK + VGA_ERR_ARG + VGA_ERR_HIP + VGA_ERR_NOMEM + VGA_ERR_UNSUPPORTED + VGA_ERR_NO_INDEX + VGA_ERR_NO_DEVICE + VGA_ERR_POOL);',
            "  return 0;", "}"]
    cfile = tmp_path / "abi.c"
    cfile.write_text("\n".join(src) + "\n")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(cfile), "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split("\n")
    sizes = {l.split()[1]: int(l.split()[2]) for l in out if l.startswith("S ")}
    offs = {(l.split()[1], l.split()[2]): (int(l.split()[3]), int(l.split()[4])) for l in out if l.startswith("F ")}
    b = pkg().binding
    for name, cls_name in ABI_STRUCTS.items():
        cls = getattr(b, cls_name)
        assert C.sizeof(cls) == sizes[name], (name, C.sizeof(cls), sizes[name])
        assert [f[0] for f in cls._fields_] == structs[name], (name, [f[0] for f in cls._fields_], structs[name])
        for fname, ftype in cls._fields_:
            d = getattr(cls, fname)
            assert (d.offset, d.size) == offs[(name, fname)], (name, fname, (d.offset, d.size), offs[(name, fname)])
    assert b.KMERPOS_DTYPE.itemsize == sizes["vga_kmerpos"] == 24
    assert sizes["vga_map_params"] == 32 and sizes["vga_poa_params"] == 40


def test_no_cpu_fallback_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    p = pkg()
    with pytest.raises(p.VgaError) as ei:
        p.Context(0)
    assert ei.value.code == -6  # VGA_ERR_NO_DEVICE


def test_product_never_imports_the_oracle():
    """The product tree must not reference oracle/ (tests, smoke and bench's cpu_baseline may)."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "rs-vgaligner_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_py|libvga_oracle|vga_oracle\.h|from oracle|import oracle", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
