"""rs-vgaligner_amd: MI355X-native map -> chain -> align path behind rs-vgaligner's interface.

The product is csrc/ (HIP kernels + the C ABI of include/vga_hip.h, built into libvga_hip.so) and
the C++ host in host/ (index build, GAF output, `vgaligner index|map`).  The Python modules here are
thin ctypes plumbing for tests and bench.py.  The directory name is not a valid Python identifier;
__graft_entry__.load_package() imports it as `rs_vgaligner_amd`.
"""
from . import binding, gafcompare, hostlib, readsim, sharding  # noqa: F401
from .hostlib import HostIndex  # noqa: F401
from .binding import Context, VgaError, default_map_params, default_poa_params, load_library  # noqa: F401
