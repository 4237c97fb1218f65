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
    assert L.vga_abi_version() == 3


def test_struct_layouts_match_header():
    p = pkg()
    b = p.binding
    assert C.sizeof(b.KmerPos) == 24
    assert C.sizeof(b.MapParams) == 32
    assert C.sizeof(b.PoaParams) == 40
    assert b.KMERPOS_DTYPE.itemsize == 24


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
