"""GPU: the C++ plug-in API mirror (phy-engine_amd/include/phy_engine) and the FFI loader subset
(include/phy_engine_dll_api.h), exercised by reference-style test programs (tests/cpp/*.cpp: one main() per file,
exit 0 = pass, the idiom of the reference's test/CMakeLists.txt:42-64).  Each mirrors a test of the reference:
  rc_step_tr      test/0005.models/rc_step_tr.cpp
  dc_divider      test/0004.solver/dc.cpp
  op_pn_junction  test/0011.nonlinear/op_pn_junction.cpp
  bridge_tr       config C2 through full_bridge_rectifier + refusal of a host-only user model
  dll_smoke       test/0008.dll/dll_main_smoke.cpp
"""
import os
import subprocess

import pytest

from parity_common import ROOT

CPP = os.path.join(ROOT, "tests", "cpp")
TESTS = ["rc_step_tr", "dc_divider", "op_pn_junction", "bridge_tr", "dll_smoke"]


@pytest.fixture(scope="module")
def built():
    subprocess.run(["make", "-C", CPP], check=True, capture_output=True)
    return os.path.join(CPP, "_build")


@pytest.mark.gpu
@pytest.mark.parametrize("name", TESTS)
def test_reference_style_program(built, name):
    out = subprocess.run([os.path.join(built, name)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, f"{name} exited {out.returncode}: {out.stderr}"


def test_cpp_api_compiles_and_loader_symbols_exported(built, pe):
    """CPU: the reference-style programs compile against the host layer, and libpe_hip.so exports the FFI loader."""
    for name in TESTS:
        assert os.path.exists(os.path.join(built, name))
    import re
    hdr = open(os.path.join(ROOT, "include", "phy_engine_dll_api.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr.split('extern "C" {')[1], flags=re.S)
    names = sorted(set(re.findall(r"\b([a-z_]+)\s*\(", body)))
    lib = pe.ffi.lib()
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/phy_engine_dll_api.h but not exported"
    assert {"create_circuit", "destroy_circuit", "circuit_analyze", "circuit_sample", "analyze_circuit", "phy_engine_last_error"} <= set(names)


def test_loader_fails_loudly_without_gpu(built, pe):
    """CPU: no device -> analysis reports failure with a message (never a silent CPU result)."""
    if pe.ffi.lib().pe_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    out = subprocess.run([os.path.join(built, "dll_smoke")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "no CPU fallback" in out.stderr
