"""GPU: the C++ plug-in API mirror (phy-engine_amd/include/phy_engine) and the FFI loader subset
(include/phy_engine_dll_api.h), exercised by reference-style test programs (tests/cpp/*.cpp: one main() per file,
exit 0 = pass, the idiom of the reference's test/CMakeLists.txt:42-64):
  known_answers   own netlists with closed-form answers: the 3 V / 10 + 20 ohm string (3 V, 2 V, 0.1 A), RC charging to 1 - 1/e,
                  the diode operating point (KCL with the Shockley law; the real reference's 0.62944165 V)
  user_model_overlay  plug-in models with host hooks only (no device table): diode, capacitor with its own companion, cubic conductor
  overlay_batch   the host-stamp overlay on a batch of three instances (PE_HIP_OVERLAY_INSTANCE, C ABI): equals three runs of one, bit for bit
  bridge_tr       config C2 through full_bridge_rectifier + a host-stamped user resistor
  dll_smoke       test/0008.dll/dll_main_smoke.cpp
  linear_models   the known answers of test/0005.models/{vccs_dc,vcvs_gain,cccs_dc,ccvs_dc,op_amp_follower,transformer_ratio,
                  transformer_center_tap_ratio,switch_r_open,generator_dc,coupled_inductors_TR,relay_hysteresis}.cpp
  hip_sparse_lu_seam  the solver-seam binding of INTEGRATION.md section A as written there: is_available / solve_csr_real / the complex
                  solve_csr_timed + solve_csr (cuda_sparse_lu.h:295-312, 465-473; circuit.h:1134, 1320, 1332), cached patterns, a singular system
  dll_elements / transistors / dll_mixed_signal / dll_digital_blocks   the loader's element codes 7-23, 50-53, 19 + 200-212, 220-229 (dll_api.h:51-135)
The same programs also run on the CPU against the host emulation of the kernels (tests/emu) to check the host-side logic.
"""
import os
import subprocess

import pytest

from parity_common import ROOT, make

CPP = os.path.join(ROOT, "tests", "cpp")
TESTS = ["known_answers", "user_model_overlay", "overlay_batch", "bridge_tr", "dll_smoke", "linear_models", "dll_elements", "transistors", "dll_mixed_signal", "ac_lowpass", "dll_digital_blocks", "hip_sparse_lu_seam"]  # adc_flash: checked against the golden below


@pytest.fixture(scope="module")
def built():
    make("-C", CPP)
    return os.path.join(CPP, "_build")


@pytest.fixture(scope="module")
def built_emu():
    make("-C", os.path.join(ROOT, "tests", "emu"))
    make("-j8", "-C", CPP, "emu")
    return os.path.join(CPP, "_build_emu")


@pytest.mark.parametrize("name", TESTS)
def test_reference_style_program_under_host_emulation(built_emu, name):
    """CPU: plug-in API, loader and digital event queue logic with the kernels emulated by a one-thread team (tests/emu,
    test infrastructure).  The parity proper is the GPU run of the same programs below."""
    out = subprocess.run([os.path.join(built_emu, name)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, f"{name} exited {out.returncode}: {out.stderr}"


@pytest.mark.gpu
@pytest.mark.parametrize("name", TESTS)
def test_reference_style_program(built, name):
    out = subprocess.run([os.path.join(built, name)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, f"{name} exited {out.returncode}: {out.stderr}"


@pytest.mark.gpu
def test_adc_mixed_signal_matches_reference(built):
    """Config C4: flash ADC (analog ladder on the GPU, comparators + NOT/AND one-hot encoder on the host event queue).
    Digital outputs and comparator states bit-exact against the real reference (tests/golden/adc_c4.json, made by
    oracle/ref_adc.cpp); ladder node voltages within 1e-12.  The sample vin = 8/16 Vref sits EXACTLY on a threshold:
    the comparator there compares two doubles that differ in the last bits between any two LU implementations, so for
    that one sample the decision is checked against this engine's own ladder voltage instead of the reference's bit."""
    import json
    out = subprocess.run([os.path.join(built, "adc_flash")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout)["samples"]
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "adc_c4.json")))["samples"]
    assert len(got) == len(ref) == 7
    for g, r in zip(got, ref):
        assert g["ok"] == r["ok"] == 1
        assert abs(g["v_vin"] - r["v_vin"]) <= 1e-12
        assert max(abs(a - b) for a, b in zip(g["ladder"], r["ladder"])) <= 1e-12
        on_threshold = any(abs(r["vin"] - t) < 1e-9 for t in r["ladder"][:15])
        if on_threshold:
            exp_cmp = [1 if g["v_vin"] >= t else 0 for t in g["ladder"][:15]]
            assert g["cmp"] == exp_cmp
            assert sum(g["out"]) == 1 and g["out"].index(1) == sum(exp_cmp)
        else:
            assert g["cmp"] == r["cmp"], (g["vin"], g["cmp"], r["cmp"])
            assert g["out"] == r["out"], (g["vin"], g["out"], r["out"])
        # the transient leg of C4 (TR dt 1e-6 x 10 steps per sample, then the ticks again): same checks
        gt, rt = g["tr"], r["tr"]
        assert gt["ok"] == rt["ok"] == 1
        assert abs(gt["v_vin"] - rt["v_vin"]) <= 1e-12
        assert max(abs(a - b) for a, b in zip(gt["ladder"], rt["ladder"])) <= 1e-12
        if on_threshold:
            exp_cmp = [1 if gt["v_vin"] >= t else 0 for t in gt["ladder"][:15]]
            assert gt["cmp"] == exp_cmp
            assert sum(gt["out"]) == 1 and gt["out"].index(1) == sum(exp_cmp)
        else:
            assert gt["cmp"] == rt["cmp"] and gt["out"] == rt["out"], (g["vin"], gt, rt)


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
