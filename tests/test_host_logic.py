"""CPU: host logic of the product (no compute on a GPU): the C-ABI library loads and exports every symbol the
header declares, fails loudly without a device, and the symbolic analysis satisfies its structural invariants."""
import os
import re
import subprocess

import numpy as np
import pytest

from parity_common import ROOT, golden, make


def test_library_exports_every_header_symbol(pe):
    hdr = open(os.path.join(ROOT, "include", "pe_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(pe_hip_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = pe.ffi.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/pe_hip.h but not exported by libpe_hip.so"
    assert sorted(pe.ffi.EXPORTS) == declared


def test_library_exports_every_loader_symbol(pe):
    """Every function include/phy_engine_dll_api.h declares (the loader subset AND the refusing out-of-scope entry points) is
    exported by libpe_hip.so; the refusing ones set the error message and return their failure value (no GPU needed)."""
    import ctypes as C
    hdr = open(os.path.join(ROOT, "include", "phy_engine_dll_api.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b([a-z][a-z_0-9]+)\s*\(", body)) - {"defined", "sizeof"})
    assert len(declared) >= 90, declared
    lib = pe.ffi.lib()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/phy_engine_dll_api.h but not exported by libpe_hip.so"
    lib.phy_engine_last_error.restype = C.c_char_p
    lib.verilog_runtime_create.restype = C.c_void_p
    lib.phy_engine_clear_error()
    assert lib.verilog_runtime_create(b"", 0, b"", 0, None, None, 0) is None
    assert b"verilog_runtime_create" in lib.phy_engine_last_error()
    lib.pl_experiment_create.restype = C.c_void_p
    assert lib.pl_experiment_create(0) is None and b"pl_experiment_create" in lib.phy_engine_last_error()
    assert lib.pl_pe_circuit_analyze(None) != 0
    lib.verilog_synth_get_loop_unroll_limit.restype = C.c_size_t
    assert lib.verilog_synth_get_loop_unroll_limit() == 64  # src/dll_main.cpp:61


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/phy_engine"), reason="build container only: the reference's ctypes client is never copied or shipped")
def test_reference_ctypes_client_binds_and_runs(emu_lib):
    """the reference's own Python client (python/phy_engine, imported in place from /root/reference) binds
    all 90 symbols of libpe_hip.so, and -- against the loader built on the host emulation, since this container has no GPU --
    drives create_circuit -> circuit_analyze -> circuit_sample on the known answer of test/0008.dll/dll_main_smoke.cpp
    (VDC 5 V across 1 kOhm: 5 V, 5 mA), and gets a loud refusal from an out-of-scope entry point."""
    emu_dir = os.path.dirname(emu_lib)
    make("-C", emu_dir, "libphyengine_emu.so")
    product = os.path.join(ROOT, "phy-engine_amd", "libpe_hip.so")
    code = f"""
import ctypes as ct, os, sys
sys.path.insert(0, '/root/reference/python')
os.environ['PHY_ENGINE_LIB'] = {os.path.join(emu_dir, 'libphyengine_emu.so')!r}
import phy_engine
from phy_engine import _ffi
_ffi._configure_library(ct.CDLL({product!r}))          # the product library: every symbol the client binds exists
E, W = phy_engine.Element, phy_engine.Wire
# elements: 0 = VDC 5 V, 1 = R 1 kOhm, 2 = ground placeholder; wires (ele, pin, ele, pin)
c = phy_engine.Circuit([E(4, (5.0,)), E(1, (1000.0,)), E(0)], [W(0, 0, 1, 0), W(0, 1, 1, 1), W(0, 1, 2, 0)])
c.set_analyze_type(phy_engine.AnalyzeType.DC)
s = c.analyze_and_sample()
vdc, res = s.components
assert abs(abs(vdc.pin_voltages[0] - vdc.pin_voltages[1]) - 5.0) < 1e-9, s
assert abs(abs(vdc.branch_currents[0]) - 5e-3) < 1e-12, s
c.close()
try:
    phy_engine.VerilogRuntime('module top; endmodule')
    raise SystemExit('the Verilog runtime must refuse')
except phy_engine.PhyEngineError as e:
    assert 'not available' in str(e), e
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)


def test_no_silent_cpu_fallback(pe):
    """Without a HIP device the engine must refuse to exist (no CPU numeric path in the product)."""
    if pe.ffi.lib().pe_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(pe.ffi.PeHipError) as e:
        pe.ffi.Engine()
    assert e.value.code == pe.ffi.ERR_NO_DEVICE and "no CPU fallback" in str(e.value)


def test_product_sources_never_touch_the_oracle():
    bad = []
    for base in ("phy-engine_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"pe_oracle|load_oracle|oracle/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, f"product code references the oracle: {bad}"


def _mesh_pattern(pe, oracle_mod, W, nonlinear=False):
    d = pe.deck.rc_mesh(W, W, 1, nonlinear)
    o = oracle_mod.Oracle(d)
    o.prepare()
    o.update_tr_step(1e-10)
    o.t = 1e-10
    A, _ = o.assemble("TR")
    A = A.tocsr()
    A.sort_indices()
    return d, A


@pytest.mark.parametrize("W", [8, 32, 100])
def test_symbolic_invariants_on_mesh(pe, oracle_mod, W):
    d, A = _mesh_pattern(pe, oracle_mod, W)
    n = A.shape[0]
    info = pe.ffi.analyze_pattern(n, A.indptr, A.indices, A.data)
    p, u, par = pe.ffi.analyze_pattern_fronts(n, A.indptr, A.indices, A.data)
    assert info["rows"] == n and info["nnz_a"] == A.nnz
    assert p.sum() == n and np.all(p >= 1)                      # every unknown is a pivot exactly once
    assert np.all((par == -1) | (par > np.arange(len(par))))    # postorder: parents after children
    assert np.all(u[par == -1] == 0)                            # roots have no update rows
    ch = par >= 0
    assert np.all(u[ch] <= (p + u)[par[ch]])                    # a child's update rows live in its parent's front
    assert info["nnz_lu"] >= A.nnz and info["nnz_lu_stored"] >= info["nnz_lu"] * 0.99
    if W == 100:
        # Eigen/COLAMD fill on this matrix is 645 757; the ND ordering must not be worse
        assert n == 10002 and A.nnz == 49605
        assert info["nnz_lu"] < 645757
        assert info["n_row_swaps"] == 2                         # exactly the V-source branch row <-> its node row


def test_symbolic_handles_zero_diagonal_chain(pe, oracle_mod):
    """VDC in series with L: the node between them has no conductance (zero MNA diagonal, g_min = 0)."""
    d = pe.deck.rlc_series_vl()
    o = oracle_mod.Oracle(d)
    o.update_tr_step(1e-6)
    A, _ = o.assemble("TR")
    A = A.tocsr()
    A.sort_indices()
    info = pe.ffi.analyze_pattern(A.shape[0], A.indptr, A.indices, A.data)
    assert info["n_row_swaps"] >= 2


def test_structurally_singular_is_reported(pe):
    rp = np.array([0, 1, 2, 2], dtype=np.int32)  # third equation empty
    ci = np.array([0, 1], dtype=np.int32)
    with pytest.raises(pe.ffi.PeHipError) as e:
        pe.ffi.analyze_pattern(3, rp, ci, np.ones(2))
    assert e.value.code == pe.ffi.ERR_SINGULAR


def test_deck_tables_follow_reference_numbering(pe):
    """circuit.h:509-531: branches are numbered in model order after the digital drives; FBR -> four diodes."""
    d = pe.deck.bridge_rectifier()
    n_nodes, n_br, tables = pe.ffi.deck_tables(d, n_drives=2)
    kinds = {t[0]: t for t in tables}
    assert n_nodes == 3 and n_br == 3
    assert list(kinds[pe.ffi.VAC][2]) == [2]
    dn = kinds[pe.ffi.DIODE][1].tolist()
    assert dn == [[1, 3], [2, 3], [0, 1], [0, 2]]               # full_bridge_rectifier.h:19-24
    assert kinds[pe.ffi.DIODE][3][0, 10] == 0.0                 # tt_in_tr off: FBR has no iterate_tr


def test_mesh_deck_matches_survey_counts(pe):
    d = pe.deck.rc_mesh(100, 100, 1, True)
    assert d.rows == 10002 and d.count("R") == 19801 and d.count("C") == 10000 and d.count("D") == 1249  # (i+j) % 8 == 0 on 0..99 x 0..99 (SURVEY.md says "1 250": off by one)
    base, r, c = pe.deck.rc_mesh_params(100, 100, [1, 2], True)
    rr = np.array([p[0] for k, _, p in base.devices if k == "R"])
    assert np.array_equal(r[0], rr)                             # instance 0 of a sweep == the single-instance deck
    d2 = pe.deck.rc_mesh(100, 100, 2, True)
    assert np.array_equal(c[1], np.array([p[0] for k, _, p in d2.devices if k == "C"]))


# ---- host emulation of the kernels' index logic (tests/emu: one-thread team, test infrastructure only) -------
@pytest.fixture(scope="module")
def emu_lib():
    emu = os.path.join(ROOT, "tests", "emu")
    make("-C", emu)
    return os.path.join(emu, "libpe_hip_emu.so")


@pytest.mark.parametrize("name,parts", [("rc_step", 1), ("rlc_series_vl_trop", 1), ("diode_op", 1), ("bridge_c2", 1), ("mesh32_nl_seed2", 1),
                                        ("ladder_c1", 1), ("mesh32_nl_seed2", 6), ("mesh32_lin", 16), ("ladder_c1", 4),
                                        ("mesh32_nl_seed2", -1), ("mesh32_lin", -4), ("ladder_c1", -1),
                                        # split schedule with every kind of x-dependent device: Newton iterations after the first stamp
                                        # only the slots those devices contribute to (stamp_dynamic_chunk)
                                        ("cmos_inverter_tr", 4), ("bjt_amp_tr", 4), ("relay_ramp_tr", 4), ("nmos_triode_op", 4), ("bridge_c2", 4),
                                        ("diode_op", 2)] +
                         [(n, 1) for n in ("vccs_dc", "vcvs_gain", "cccs_dc", "ccvs_dc", "op_amp_follower", "transformer_ratio", "generator_dc",
                                           "switch_open_dc", "switch_closed_dc", "switch_open_ropen1e6_dc", "generators_tr", "generators_trop",
                                           "iac_rc_tr", "iac_rc_dc", "iac_rc_trop", "coupled_l_k0_tr", "coupled_l_k09_tr", "coupled_l_k09_trop",
                                           "coupled_l_dc", "controlled_mix_tr", "nmos_cutoff_dc", "nmos_sat_dc", "nmos_triode_op",
                                           "cmos_inverter_tr", "bjt_amp_tr", "center_tap_ratio", "relay_ramp_tr")])
def test_front_code_indexing_under_host_emulation(emu_lib, name, parts):
    """Runs pe_front.hpp + pe_engine*.cpp with a ONE-THREAD team in a subprocess against the reference goldens.
    parts > 1 = the multi-workgroup schedule (level-1 cut + top levels, one launch per phase); parts < 0 = the same
    with a small LDS so that the large-front code paths are exercised too.
    This validates indexing/orchestration only; the parity proper is tests/test_gpu_parity.py on the MI355X."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
os.environ['PHY_ENGINE_HIP_PARTS'] = '{abs(parts)}'
if {parts} < 0:   # a 12 KB LDS: fronts no longer fit whole, the pivot-panel / pull / chain-link paths run
    os.environ['PHY_ENGINE_HIP_LDS_BYTES'] = '12288'
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np
from parity_common import *
meta, gx, deck = golden({name!r})
eng = pe.ffi.Engine()
snaps, trace, fail = run_engine_case(eng, meta, deck)
assert fail == -1 and len(snaps) == len(gx)
assert max_err(snaps[:, 0, :], gx, 1e-9, 1e-6) <= 1.0
assert list(trace) == meta['newton_iters']
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)


@pytest.mark.parametrize("name,geometry", [("mesh32_nl_seed2", "1024"), ("mesh32_lin", "1024"), ("ladder_c1", "1024"), ("bridge_c2", "1024")])
def test_sweep_geometry_under_host_emulation(emu_lib, name, geometry):
    """The launch geometry of the 1 024-instance sweep (four 4-wavefront workgroups per CU, split schedule) on the host emulation:
    wave fronts up to order 45 -- whole in the 10 KB slot up to 35, in the panel layout above (SymbolicOptions::wave_slot) -- the
    odd LDS leading dimensions and the lean backward pass (front_backward_lean), against the reference goldens."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
os.environ['PHY_ENGINE_HIP_GEOMETRY_BATCH'] = {geometry!r}
os.environ['PHY_ENGINE_HIP_SPLIT'] = '1'
os.environ['PHY_ENGINE_HIP_PARTS'] = '4'
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np
from parity_common import *
meta, gx, deck = golden({name!r})
eng = pe.ffi.Engine()
snaps, trace, fail = run_engine_case(eng, meta, deck)
assert fail == -1 and len(snaps) == len(gx)
assert max_err(snaps[:, 0, :], gx, 1e-9, 1e-6) <= 1.0
assert list(trace) == meta['newton_iters']
assert eng.info()['n_wavefronts'] == 4 and eng.info()['n_parts'] == 4
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)


@pytest.mark.parametrize("name", ["ac_rc_lowpass", "ac_rlc_diode_acop", "ac_linear_mix", "ac_nmos_amp"])
def test_ac_real_equivalent_system_under_host_emulation(emu_lib, name):
    """The AC path's host logic (real-equivalent 2N system, value vector per omega, operating-point hand-over) with the
    kernels emulated; compared with the reference's complex phasors."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np
from parity_common import *
meta, gx, deck = golden({name!r})
eng = pe.ffi.Engine()
xs = run_ac_case(eng, meta, deck)
g = golden_complex(meta, gx)
assert xs.shape == g.shape
assert np.all(np.abs(xs - g) <= 1e-9 + 1e-6 * np.abs(g)), np.max(np.abs(xs - g))
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)


def test_checkpoint_resume_under_host_emulation(emu_lib):
    """Host logic of pe_hip_checkpoint_save / load: an interrupted transient continues bit-identically; a blob of another
    circuit is refused."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load()
deck, r, c = pe.deck.rc_mesh_params(8, 8, [1, 2], True)
ov = {{"R": r[:, :, None], "C": c[:, :, None]}}
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0); eng.load_deck(deck, batch=2, overrides=ov); eng.reset()
eng.analyze_tr(1e-10, 30); want = eng.solution().copy()
eng.reset(); eng.analyze_tr(1e-10, 12); blob = eng.checkpoint()
e2 = pe.ffi.Engine(); e2.set_options(g_min=0.0); e2.load_deck(deck, batch=2, overrides=ov); e2.restore(blob); e2.analyze_tr(1e-10, 18)
assert np.array_equal(e2.solution(), want)
e3 = pe.ffi.Engine(); e3.load_deck(pe.deck.rc_step())
try:
    e3.restore(blob); raise SystemExit(1)
except pe.ffi.PeHipError:
    pass
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)


def test_large_fronts_under_host_emulation(emu_lib, oracle_mod):
    """A 200 x 200 diode mesh (40 002 rows, fronts up to 300, 16 top levels in the multi-workgroup schedule -- 21 before the top
    fronts were regrouped against a CU's whole LDS): the panel
    layout's LDS regions (panels + right-hand-side column + staged child maps) must fit what the launch allocates.  Regression
    test for an LDS overrun found with AddressSanitizer on the emulation build."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load(); orc = pe_load.load_oracle()
deck = pe.deck.rc_mesh(200, 200, 1, True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0); eng.load_deck(deck); eng.reset()
st = eng.analyze_tr(1e-10, 2)
o = orc.Oracle(deck); o.analyze_tr(1e-10, 2)
assert st['newton_iters'] == sum(o.newton_iters)
assert np.max(np.abs(eng.solution()[0] - o.x) / (1e-6 + 1e-5 * np.abs(o.x))) <= 1.0
i = eng.info()
assert i['lds_bytes'] <= 163840 and i['max_front'] > 150 and i['n_top_levels'] > 12
# the same circuit with the top fronts cut to the ordinary workgroup's LDS share (first analysis pass only, knob TOP_BIG=0): more, smaller
# links at the top (fronts regrouped against a CU's LDS: pe_engine_policy.cpp regroup_wide_top), the same elimination -> the same answer
eng2 = pe.ffi.Engine(); eng2.set_knob('TOP_BIG', 0); eng2.set_options(g_min=0.0); eng2.load_deck(deck); eng2.reset()
st2 = eng2.analyze_tr(1e-10, 2)
i2 = eng2.info()
assert i2['n_top_levels'] > i['n_top_levels'] and i2['n_fronts'] > i['n_fronts'], (i2['n_top_levels'], i['n_top_levels'])
assert st2['newton_iters'] == st['newton_iters']
assert np.max(np.abs(eng2.solution()[0] - eng.solution()[0]) / (1e-9 + 1e-9 * np.abs(o.x))) <= 1.0
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=900)


def test_lds_guard_refuses_the_round3_grouping(emu_lib):
    """Product-side LDS guard (reference contract: "return false, never corrupt", circuit.h:1517).  Knob TEST_OLD_TOP_RUNS re-enables the
    round-3 grouping bug -- a run of single-front top levels joins levels of different LDS classes and takes the launch of its first
    level -- under the population rule it occurred with (half-CU levels up to 1 280 workgroups, geometry of 1 024 instances): the front
    118 x 40 (panel layout, 8 076 doubles) would land on a launch with 5 072.  The load must be refused with PE_HIP_ERR_INTERNAL at plan
    time (upload_symbolic: check_lds_plan); with the knob off the same plan loads and solves.  The per-engine launch knobs (MID_TOP,
    EW_GRID, QUAD_LDS) live in the engine's view: two engines of one process keep their own."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load()
deck = pe.deck.rc_mesh(100, 100, 1, True)
def engine(old):
    eng = pe.ffi.Engine(); eng.set_knob('GEOMETRY_BATCH', 1024); eng.set_knob('TOP_HALF_WGS', 1280); eng.set_knob('TEST_OLD_TOP_RUNS', old)
    eng.set_options(g_min=0.0)
    return eng
good = engine(0); good.load_deck(deck); good.reset()
assert good.analyze_tr(1e-10, 1)['newton_iters'] == 2
bad = engine(1)
try:
    bad.load_deck(deck); bad.reset(); bad.analyze_tr(1e-10, 1)
    raise SystemExit('the mis-grouped launch plan was accepted')
except pe.ffi.PeHipError as e:
    assert e.code == pe.ffi.ERR_INTERNAL and 'needs 8076 doubles of LDS' in str(e) and 'has 5072' in str(e), str(e)
# the good engine is untouched by the other engine's knobs and failure
assert good.analyze_tr(1e-10, 1)['newton_iters'] >= 1
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=900)


def test_launch_plan_of_a_sweep_under_host_emulation(emu_lib, oracle_mod):
    """The top levels of the 10k-node mesh as a sweep of 128 / 512 / 1 024 instances runs them (knob GEOMETRY_BATCH: the geometry, the
    second analysis pass that forms the top fronts against the LDS of their launch, and the launch plan of pe_top_plan.hpp are those
    of the large sweep; one instance is computed).  The emulation gives every level of a launch that launch's LDS and asserts that a
    front fits: a level with half-CU fronts that rode along in a run of ordinary levels overran the LDS on the device
    (round 3; caught here since the emulation follows the same plan).  Same Newton counts and solution as the oracle."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load(); orc = pe_load.load_oracle()
deck = pe.deck.rc_mesh(100, 100, 1, True)
o = orc.Oracle(deck); o.analyze_tr(1e-10, 2)
levels = {{}}
for g in (128, 512, 1024):
    eng = pe.ffi.Engine(); eng.set_knob('GEOMETRY_BATCH', g); eng.set_options(g_min=0.0); eng.load_deck(deck); eng.reset()
    st = eng.analyze_tr(1e-10, 2)
    assert st['newton_iters'] == sum(o.newton_iters), (g, st['newton_iters'])
    assert np.max(np.abs(eng.solution()[0] - o.x) / (1e-6 + 1e-5 * np.abs(o.x))) <= 1.0, g
    levels[g] = eng.info()['n_top_levels']
    eng0 = pe.ffi.Engine(); eng0.set_knob('GEOMETRY_BATCH', g); eng0.set_knob('TOP_BIG', 0); eng0.set_options(g_min=0.0); eng0.load_deck(deck); eng0.reset()
    eng0.analyze_tr(1e-10, 2)
    assert eng0.info()['n_top_levels'] > levels[g], (g, eng0.info()['n_top_levels'], levels[g])
    assert np.max(np.abs(eng0.solution()[0] - eng.solution()[0]) / (1e-9 + 1e-9 * np.abs(o.x))) <= 1.0, g
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=900)


def test_solver_seam_large_front_under_host_emulation(emu_lib):
    """solve_csr_real with ONE dense front of order 450 and with a 2-D mesh whose top separator exceeds the default LDS
    reserve: the seam runs the same LDS-fit escalation as the resident circuit (round-1 advisor finding: an LDS overrun past
    ~376 rows).  The emulation build asserts `front image + right-hand-side column <= region` in front_factor."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r})
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla, pe_load
pe = pe_load.load()
rng = np.random.default_rng(7)
n = 450
A = sp.csr_matrix(rng.standard_normal((n, n)) + n * np.eye(n)); A.sort_indices()
b = rng.standard_normal(n)
eng = pe.ffi.Engine()
x, _ = eng.solve_csr(n, A.indptr, A.indices, A.data, b, copy_pattern=True)
assert np.max(np.abs(A @ x - b)) < 1e-9
# 5-point Laplacian on a 70 x 70 grid + a dense 400-clique coupling block on the diagonal: fronts of several hundred rows
g = 70
L = sp.kron(sp.eye(g), sp.diags([-1, 4.5, -1], [-1, 0, 1], shape=(g, g))) + sp.kron(sp.diags([-1, -1], [-1, 1], shape=(g, g)), sp.eye(g))
L = sp.csr_matrix(L); L.sort_indices()
b2 = rng.standard_normal(g * g)
x2, _ = eng.solve_csr(g * g, L.indptr, L.indices, L.data, b2, copy_pattern=True)
assert np.max(np.abs(x2 - spla.splu(L.tocsc()).solve(b2))) < 1e-9
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=600)


def test_complex_solver_seam_under_host_emulation(emu_lib):
    """pe_hip_solve_csr_complex (the twin of cuda_sparse_lu::solve_csr_timed, cuda_sparse_lu.h:304-312) on the host emulation: the
    assembled complex systems of the `ac_rlc_diode_acop` golden (first call analyses, later calls reuse the pattern and fall back to a
    re-analysis when the cached pivot order does not suit the new frequency), a complex mesh against scipy, a singular system."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
from parity_common import *
orc = pe_load.load_oracle()
meta, gx, deck = golden('ac_rlc_diode_acop')
g = golden_complex(meta, gx)
o = orc.Oracle(deck); o.g_min = meta['gmin']; o.prepare(); assert o.solve('OP') >= 0
eng = pe.ffi.Engine()
for k, w in enumerate(meta['omegas']):
    A, rhs = o.stamp_ac(w)
    keys = sorted(A.keys())
    M = sp.csr_matrix((np.array([A[q] for q in keys], dtype=complex), (np.array([q[0] for q in keys]), np.array([q[1] for q in keys]))), shape=(o.rows, o.rows))
    M.sort_indices()
    x, tm = eng.solve_csr_complex(o.rows, M.indptr, M.indices, M.data, rhs, copy_pattern=(k == 0))
    assert np.all(np.abs(x - g[k]) <= 1e-9 + 1e-6 * np.abs(g[k])), (w, x, g[k])
mo = orc.Oracle(pe.deck.rc_mesh(16, 16, 5, False)); mo.update_tr_step(1e-10); mo.t = 1e-10
A, b = mo.assemble('TR'); A = A.tocsr(); A.sort_indices()
rng = np.random.default_rng(3)
Z = sp.csr_matrix((A.data * (1.0 + 1j * rng.uniform(-2.0, 2.0, A.nnz)), A.indices, A.indptr), shape=A.shape)
bz = b * (1.0 - 0.5j) + 1j * rng.standard_normal(len(b)) * 1e-3
x, _ = eng.solve_csr_complex(Z.shape[0], Z.indptr, Z.indices, Z.data, bz, copy_pattern=True)
xr = spla.splu(Z.tocsc()).solve(bz)
assert np.max(np.abs(x - xr)) <= 1e-9 * max(1.0, np.max(np.abs(xr)))
x2, _ = eng.solve_csr_complex(Z.shape[0], Z.indptr, Z.indices, Z.data * (2.0 - 1.0j), bz, copy_pattern=False)
assert np.max(np.abs((2.0 - 1.0j) * x2 - xr)) <= 1e-9 * max(1.0, np.max(np.abs(xr)))
S = sp.csr_matrix(np.array([[1.0 + 1.0j, 2.0 + 2.0j], [2.0 + 2.0j, 4.0 + 4.0j]]))
try:
    eng.solve_csr_complex(2, S.indptr, S.indices, S.data, np.array([1.0, 1.0j]), copy_pattern=True)
    raise SystemExit('a singular complex system was accepted')
except pe.ffi.PeHipError as e:
    assert e.code in (pe.ffi.ERR_SINGULAR, pe.ffi.ERR_INACCURATE), e.code
# the engine is usable after the refusal; a cached pattern of another size is not reused (the reference's callers pass copy_pattern = false
# whenever THEY think the pattern is unchanged); an empty system is a no-op; inconsistent sizes are argument errors, not crashes
D = sp.csr_matrix(np.diag([2.0 + 1.0j, 1.0 - 3.0j, 4.0j])); D.sort_indices()
x, _ = eng.solve_csr_complex(3, D.indptr, D.indices, D.data, np.array([2.0 + 1.0j, 2.0 - 6.0j, -4.0]), copy_pattern=False)
assert np.max(np.abs(x - np.array([1.0, 2.0, 1.0j]))) < 1e-14
x0, _ = eng.solve_csr_complex(0, np.zeros(1, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0, dtype=complex), np.zeros(0, dtype=complex))
assert x0.shape == (0,)
import ctypes as C
l = pe.ffi.lib()
rp = np.array([0, 1, 3], dtype=np.int32); ci = np.array([0, 0, 1], dtype=np.int32); va = np.ones(6); bb = np.ones(4); xx = np.zeros(4)
ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int)); dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
assert l.pe_hip_solve_csr_complex(eng._h, 2, 2, ip(rp), ip(ci), dp(va), dp(bb), dp(xx), 1, None) == pe.ffi.ERR_ARG      # nnz != row_ptr[n]
assert l.pe_hip_solve_csr_complex(eng._h, 2, 3, ip(rp), ip(ci), None, dp(bb), dp(xx), 1, None) == pe.ffi.ERR_ARG        # null values
assert l.pe_hip_solve_csr_complex(None, 2, 3, ip(rp), ip(ci), dp(va), dp(bb), dp(xx), 1, None) == pe.ffi.ERR_ARG
bad_ci = np.array([0, 1, 0], dtype=np.int32)                                                                             # unsorted columns in row 1
assert l.pe_hip_solve_csr_complex(eng._h, 2, 3, ip(rp), ip(bad_ci), dp(va), dp(bb), dp(xx), 1, None) == pe.ffi.ERR_ARG
far_ci = np.array([0, 0, 7], dtype=np.int32)                                                                             # column out of range
assert l.pe_hip_solve_csr_complex(eng._h, 2, 3, ip(rp), ip(far_ci), dp(va), dp(bb), dp(xx), 1, None) == pe.ffi.ERR_ARG
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=600)


def test_static_matrix_stamp_is_invalidated_by_everything_that_changes_it(emu_lib, tmp_path):
    """Round 4: after a full stamp at step size dt the first Newton iteration of the following transient steps gathers only the x-dependent
    matrix slots + the whole right-hand side (stamp mode 2; the rest of the matrix is the same from one time point to the next while dt and the
    parameters stay).  The knowledge must be dropped by everything that changes a static value: a parameter update, another dt, an operating
    point in between (it stamps the DC companions), g_min, a reset.  A scripted sequence of those, split schedule, two instances, against the
    same sequence with the knob STATIC_A=0 (always the full stamp): bit for bit, same Newton counts."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load()
deck, r, c = pe.deck.rc_mesh_params(12, 12, [1, 2], True)
e = pe.ffi.Engine(); e.set_knob('SPLIT', 1); e.set_knob('PARTS', 3); e.set_options(g_min=0.0)
e.load_deck(deck, batch=2, overrides={{"R": r[:, :, None], "C": c[:, :, None]}}); e.reset()
out = []
def snap(): out.append(e.solution().copy()); out.append(np.array(e.state()['iters'], dtype=float))
e.analyze_tr(1e-10, 4); snap()
e.update_param(pe.ffi.R, 5, 0, [2500.0, 700.0]); e.analyze_tr(1e-10, 3); snap()      # a resistor changes: static slot
e.analyze_tr(2e-10, 3); snap()                                                          # another step size: every companion conductance
e.analyze_dc(pe.ffi.MODE_OP); snap(); e.analyze_tr(2e-10, 3); snap()                    # an operating point in between
e.set_options(g_min=1e-9); e.analyze_tr(2e-10, 2); snap()                               # g_min sits on every node diagonal
e.reset(); e.analyze_tr(2e-10, 3); snap()
np.save(sys.argv[1], np.concatenate([o.ravel() for o in out]))
"""
    res = []
    for knob in ("1", "0"):
        f = str(tmp_path / f"s{{knob}}.npy".format(knob=knob))
        subprocess.run(["python3", "-c", code, f], check=True, timeout=600, env=dict(os.environ, PHY_ENGINE_HIP_STATIC_A=knob))
        res.append(np.load(f))
    assert res[0].shape == res[1].shape and np.array_equal(res[0], res[1])


def test_failed_solve_is_not_sticky_under_host_emulation(emu_lib):
    """circuit.h:242-254: a failed transient rolls tr_duration back and returns false; the NEXT analyze() tries again from that
    state.  Here: the g_min = 0 bridge fails (singular with all four diodes off), the caller raises g_min, and the same resident
    circuit continues from the rolled-back time (round-1 advisor finding: the failure used to be permanent until a reset)."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))
import numpy as np
from parity_common import golden, pe
meta, gx, deck = golden('bridge_gmin0_fail')
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0); eng.load_deck(deck); eng.reset()
st = eng.analyze_tr(meta['dt'], meta['steps'], check=False)
assert st['rc'] in (pe.ffi.ERR_SINGULAR, pe.ffi.ERR_NO_CONVERGENCE)
s0 = eng.state(); t_fail = float(s0['t'][0]); n_ok = int(s0['steps'][0])
assert abs(t_fail - n_ok * meta['dt']) < 1e-12
eng.set_options(g_min=1e-12)
st2 = eng.analyze_tr(meta['dt'], 50, check=False)
assert st2['rc'] == 0 and st2['steps'] == 50, st2
s1 = eng.state()
assert s1['status'][0] == 0 and abs(float(s1['t'][0]) - (n_ok + 50) * meta['dt']) < 1e-12
assert np.all(np.isfinite(eng.solution()[0]))
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)


def test_tuning_knobs_are_per_engine_under_host_emulation(emu_lib):
    """pe_hip_set_knob / pe_hip_get_knob (include/pe_hip.h): the PHY_ENGINE_HIP_* family per ENGINE instead of per process -- two engines
    of one process run the same circuit under different schedules (parts of the split schedule), an engine knob wins over the
    environment variable, a knob set after the first analysis takes effect at the next one, and the results agree."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
os.environ['PHY_ENGINE_HIP_PARTS'] = '2'
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))
import numpy as np
import pe_load
pe = pe_load.load()
deck = pe.deck.rc_mesh(12, 12, 1, True)
def engine(**knobs):
    e = pe.ffi.Engine(); e.set_options(g_min=0.0)
    for k, v in knobs.items(): e.set_knob(k, v)
    e.load_deck(deck); e.reset(); e.analyze_tr(1e-10, 3)
    return e
a = engine(SPLIT=1, PARTS=3)
b = engine(PHY_ENGINE_HIP_SPLIT=1)            # (the prefix is accepted; PARTS comes from the environment: 2)
c = engine(SPLIT=0)
assert a.info()['n_parts'] == 3 and b.info()['n_parts'] == 2 and c.info()['n_parts'] == 1, (a.info()['n_parts'], b.info()['n_parts'], c.info()['n_parts'])
assert a.get_knob('PARTS') == 3 and b.get_knob('PARTS') == 2 and a.get_knob('ABSORB_M') is None
xa, xb, xc = a.solution()[0], b.solution()[0], c.solution()[0]
assert np.max(np.abs(xa - xc)) < 1e-9 and np.max(np.abs(xb - xc)) < 1e-9
c.set_knob('SPLIT', 1); c.set_knob('PARTS', 4)   # resident circuit: re-analysed at the next analysis, the transient continues
c.analyze_tr(1e-10, 2); a.analyze_tr(1e-10, 2)
assert c.info()['n_parts'] == 4
assert np.max(np.abs(c.solution()[0] - a.solution()[0])) < 1e-9
# the launch-shape knobs (round 3: function-local statics of the launcher, process-wide and frozen at first use) live in each engine's view
os.environ['PHY_ENGINE_HIP_EW_GRID'] = '3'
d = engine(SPLIT=1, MID_TOP=100, EW_GRID=7, QUAD_LDS=4096)
e = engine(SPLIT=1)                               # (created later, environment changed since the first launch of the process)
ia, id_, ie = a.info(), d.info(), e.info()
assert (id_['mid_top_limit'], id_['ew_grid'], id_['quad_lds_pad']) == (100, 7, 4096), id_
assert (ie['mid_top_limit'], ie['ew_grid'], ie['quad_lds_pad']) == (512, 3, 0), ie
assert (ia['mid_top_limit'], ia['ew_grid'], ia['quad_lds_pad']) == (512, 0, 0), ia
assert np.max(np.abs(d.solution()[0] - e.solution()[0])) < 1e-9
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=600)


def test_revived_instance_keeps_its_own_time_point_under_host_emulation(emu_lib):
    """Round-2 advisor finding (medium): in a batch on the host-driven split schedule, an instance that failed and was rolled back in
    call 1 is live again in call 2 -- at ITS time point, not the group's.  Two instances of the diode mesh, instance 1 driven hard
    enough that a Newton limit of 2 iterations stops it early while instance 0 (diodes off) runs on; call 2 (limit lifted) must
    leave each instance exactly where the same instance ends up when it is run ALONE through the same two calls: time, step count and
    solution.  (Sources are evaluated at an instance's own t: solving the revived instance at the group's time would show here.)"""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
os.environ['PHY_ENGINE_HIP_SPLIT'] = '1'
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))
import numpy as np
import pe_load
pe = pe_load.load()
deck = pe.deck.rc_mesh(12, 12, 1, True)
nvac = deck.count('VAC'); assert nvac == 1
vp = [0.02, 2.0]
def vac(amps):
    return np.array([[[a, 2.0 * np.pi * 1e8, 0.0]] for a in amps])
def two_calls(amps):
    eng = pe.ffi.Engine(); eng.set_options(g_min=0.0, max_newton=2)
    eng.load_deck(deck, batch=len(amps), overrides={{'VAC': vac(amps)}}); eng.reset()
    st1 = eng.analyze_tr(1e-10, 12, check=False)
    s1 = eng.state()
    eng.set_options(g_min=0.0, max_newton=64)
    st2 = eng.analyze_tr(1e-10, 5, check=False)
    return st1, s1, st2, eng.state(), eng.solution()
st1, s1, st2, s2, x = two_calls(vp)
assert s1['status'][0] == 0 and s1['steps'][0] == 12, s1                      # instance 0 ran all 12 steps
assert s1['status'][1] == pe.ffi.ERR_NO_CONVERGENCE and 0 < s1['steps'][1] < 12, s1   # instance 1 was stopped and rolled back
assert st2['rc'] == 0 and st2['n_failed'] == 0, st2
assert s2['steps'][0] == 17 and s2['steps'][1] == s1['steps'][1] + 5, s2       # each went on from its own time point
assert abs(s2['t'][0] - 17e-10) < 1e-20 and abs(s2['t'][1] - (s1['steps'][1] + 5) * 1e-10) < 1e-20, s2
for b in range(2):
    a1, as1, a2, as2, ax = two_calls([vp[b]])
    assert as1['steps'][0] == s1['steps'][b] and as2['steps'][0] == s2['steps'][b] and as2['t'][0] == s2['t'][b], (b, as1, as2)
    assert np.array_equal(ax[0], x[b]), (b, float(np.max(np.abs(ax[0] - x[b]))))
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=600)


def test_residual_safety_net_under_host_emulation(emu_lib):
    """Static pivoting with a safety net (the reference pivots partially, Eigen SparseLU.h:464-469): after every linear solve the
    normwise backward error is checked per instance; above residual_tol the solve is refined (correction solve on the residual),
    else the pivot order is re-matched on that instance's values, else the step fails as PE_HIP_ERR_INACCURATE and is rolled back.
      (a) an LU made inexact on purpose (emulation-only knob PE_EMU_PIVOT_ERROR: every pivot reciprocal off by 1e-7) gives a
          visibly wrong transient without the net and the golden transient with it -- every solve repaired by refinement;
      (b) an unreachable tolerance exercises the failure path: refinement and re-matching both give up, the step is rolled back,
          and the failure is not sticky (the next analyze() with a sane tolerance continues);
      (c) a 2-instance sweep whose instance 1 opens a switch (r_open = 1e12 where instance 0 has 0) and scales half the resistors
          by 1e6, solved with the pivot order matched on instance 0: both at oracle accuracy, no false alarm."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
sys.path.insert(0, {ROOT!r}); sys.path.insert(0, os.path.join({ROOT!r}, 'tests'))
import numpy as np
from parity_common import adversarial_pivot_sweep, golden, pe, max_err
import pe_load
meta, gx, deck = golden('mesh32_nl')
g10 = gx[meta['snap_steps'].index(10)]
def run(tol):
    eng = pe.ffi.Engine(); eng.set_options(g_min=0.0, residual_tol=tol); eng.load_deck(deck); eng.reset()
    st = eng.analyze_tr(1e-10, 10, check=False)
    return eng, st
if os.environ.get('PE_EMU_PIVOT_ERROR'):
    eng, st = run(-1.0)                                   # (a) net disabled: inexact LU goes through
    assert st['rc'] == 0 and max_err(eng.solution()[0], g10, 1e-9, 1e-7) > 1.0
    eng, st = run(0.0)                                    # default tolerance 1e-10
    assert st['rc'] == 0 and st['steps'] == 10, st
    assert max_err(eng.solution()[0], g10, 1e-9, 1e-7) <= 1.0
    sn = eng.safety_net(); assert sn['refined'] >= 10 and sn['careful'], sn
else:
    eng, st = run(1e-30)                                  # (b)
    assert st['rc'] == pe.ffi.ERR_INACCURATE and st['steps'] == 0, st
    s = eng.state(); assert s['t'][0] == 0.0 and s['status'][0] == pe.ffi.ERR_INACCURATE
    sn = eng.safety_net(); assert sn['careful'] and sn['rematched'] >= 1, sn
    eng.set_options(g_min=0.0, residual_tol=0.0)
    st = eng.analyze_tr(1e-10, 10, check=False)
    # (like the reference, the retried step re-applies update_tr_step on the failed iterate: the continuation is a valid run, not the golden one)
    assert st['rc'] == 0 and st['steps'] == 10 and np.all(np.isfinite(eng.solution()[0]))
    orc = pe_load.load_oracle()                           # (c)
    d0, d1, ov = adversarial_pivot_sweep()
    e2 = pe.ffi.Engine(); e2.set_options(g_min=0.0); e2.load_deck(d0, batch=2, overrides=ov); e2.reset()
    e2.analyze_dc(pe.ffi.MODE_DC)
    x = e2.solution()
    for k, d in enumerate((d0, d1)):
        o = orc.Oracle(d); o.analyze_dc('DC')
        assert max_err(x[k], o.x, 1e-9, 1e-7) <= 1.0, (k, np.max(np.abs(x[k] - o.x)))
    assert e2.safety_net() == {{'refined': 0, 'rematched': 0, 'careful': False}}
"""
    subprocess.run(["python3", "-c", code], check=True, timeout=300)
    subprocess.run(["python3", "-c", code], check=True, timeout=300, env=dict(os.environ, PE_EMU_PIVOT_ERROR="1e-7"))


def test_x_dependent_only_stamp_is_bit_identical_under_host_emulation(emu_lib, tmp_path):
    """The same as tests/test_gpu_parity.py::test_x_dependent_only_stamp_is_bit_identical, on the host emulation with every kind of
    x-dependent device in the split schedule (junctions, MOS, BJT, relay)."""
    code = f"""
import os, sys
os.environ['PE_HIP_LIB'] = {emu_lib!r}
os.environ['PHY_ENGINE_HIP_PARTS'] = '4'
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
import numpy as np
from parity_common import *
res = []
for name in ("mesh32_nl_seed2", "cmos_inverter_tr", "bjt_amp_tr", "relay_ramp_tr", "bridge_c2"):
    meta, gx, deck = golden(name)
    eng = pe.ffi.Engine()
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert eng.info()['n_parts'] == 4
    res.append(np.asarray(snaps).ravel()); res.append(np.asarray(trace, dtype=float))
np.save(sys.argv[1], np.concatenate(res))
"""
    out = []
    for knob in ("0", "1"):
        f = str(tmp_path / f"x{knob}.npy")
        subprocess.run(["python3", "-c", code, f], check=True, timeout=600, env=dict(os.environ, PHY_ENGINE_HIP_FULL_STAMP=knob))
        out.append(np.load(f))
    assert np.array_equal(out[0], out[1])
