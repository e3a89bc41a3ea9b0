"""phy-engine_amd: MI355X-native transient hot path behind Phy-Engine's solver seam.

The directory name is not a Python identifier; load it with `pe_load.load()` (repo root) which registers it
as the package `phy_engine_amd`.
  deck  -- netlist description + BASELINE workloads (host, pure Python)
  ffi   -- ctypes binding of libpe_hip.so (the product: HIP kernels + C ABI, see include/pe_hip.h)
"""
from . import deck  # noqa: F401
from . import ffi  # noqa: F401
