"""GPU (MI355X): the HIP path, called through the C ABI (libpe_hip.so), against
  (1) golden vectors captured from the real reference (tests/golden, scripts/make_golden.py),
  (2) the CPU oracle on freshly seeded inputs,
  (3) size-independent properties at BASELINE.json's full size.
Tolerances (fp64): linear circuits abs 1e-9 + rel 1e-7; non-linear abs 1e-6 + rel 1e-5 (the Newton stop rule is
1e-3 relative) -- in practice the iterates agree to ~1e-9 and the Newton counts are identical."""
import os

import numpy as np
import pytest

from parity_common import golden, golden_complex, max_err, pe, run_ac_case, run_engine_case

pytestmark = pytest.mark.gpu

LIN = (1e-9, 1e-7)
NL = (1e-6, 1e-5)


@pytest.fixture(scope="module")
def eng():
    e = pe.ffi.Engine(device=0)  # raises when libpe_hip.so / the GPU is missing: no fallback exists
    yield e
    e.close()


@pytest.mark.parametrize("name,tol", [
    ("rc_step", LIN), ("rl_step", LIN), ("rlc_series_vl", LIN), ("rlc_series_vl_trop", LIN), ("divider_dc", LIN),
    ("ladder_c1", LIN), ("diode_op", NL), ("pn_tt_tr", NL), ("bridge_c2", NL),
    ("mesh32_lin", LIN), ("mesh32_nl", NL), ("mesh100_lin", LIN), ("mesh100_nl", NL),
])
def test_golden_parity(eng, name, tol):
    meta, gx, deck = golden(name)
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert fail == -1 and meta["fail_step"] == -1
    assert len(snaps) == len(gx)
    assert max_err(snaps[:, 0, :], gx, *tol) <= 1.0
    assert list(trace) == meta["newton_iters"]          # same Newton trajectory as the reference, step by step


# ---- remaining linear stampers (IAC, VCCS, VCVS, CCCS, CCVS, op-amp, transformer, switch, the four
# generators, coupled inductors); goldens = the reference's own model tests (test/0005.models) + transient variants
@pytest.mark.parametrize("name,tol", [
    ("vccs_dc", LIN), ("vcvs_gain", LIN), ("cccs_dc", LIN), ("ccvs_dc", LIN), ("op_amp_follower", LIN), ("transformer_ratio", LIN),
    ("generator_dc", LIN), ("switch_open_dc", LIN), ("switch_closed_dc", LIN), ("switch_open_ropen1e6_dc", LIN), ("generators_tr", LIN),
    ("generators_trop", LIN), ("iac_rc_tr", LIN), ("iac_rc_dc", LIN), ("iac_rc_trop", LIN), ("coupled_l_k0_tr", LIN),
    ("coupled_l_k09_tr", LIN), ("coupled_l_k09_trop", LIN), ("coupled_l_dc", LIN), ("controlled_mix_tr", NL),
])
def test_stamper_golden_parity(eng, name, tol):
    meta, gx, deck = golden(name)
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert fail == -1 and meta["fail_step"] == -1
    assert len(snaps) == len(gx)
    assert max_err(snaps[:, 0, :], gx, *tol) <= 1.0
    assert list(trace) == meta["newton_iters"]


@pytest.mark.parametrize("name", ["nmos_cutoff_dc", "nmos_sat_dc", "nmos_triode_op", "cmos_inverter_tr", "bjt_amp_tr", "center_tap_ratio",
                                  "relay_ramp_tr"])
def test_three_pin_nonlinear_golden_parity(eng, name):
    """Level-1 MOSFETs and the forward-active BJT (device kinds 18-21), the relay with its hysteresis state (22) and the
    center-tap transformer (23): same Newton trajectory as the reference."""
    meta, gx, deck = golden(name)
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert fail == -1 and meta["fail_step"] == -1
    assert len(snaps) == len(gx)
    assert max_err(snaps[:, 0, :], gx, *NL) <= 1.0
    assert list(trace) == meta["newton_iters"]


@pytest.mark.parametrize("name", ["bjt_npn_ce_dc_fail", "bjt_pnp_ce_op_fail", "bjt_amp_trop_fail"])
def test_unlimited_exponential_fails_like_reference(eng, name):
    """The reference's BJT exponential has no junction limiting: a cold-start operating point overflows and the reference
    reports failure (64 Newton iterations).  The engine must fail too (non-finite iterate or no convergence), not invent an answer."""
    meta, gx, deck = golden(name)
    assert meta["fail_step"] == 0
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert fail == 0
    assert eng.state()["status"][0] in (pe.ffi.ERR_SINGULAR, pe.ffi.ERR_NO_CONVERGENCE)


def test_mosfet_param_update(eng):
    """update_param on a resident MOSFET (Vth) and BJT (Temp -> N*Ut derived on the host) re-solves with the new values."""
    d = pe.deck.nmos_common_source(2.0)
    eng.set_options(g_min=0.0)
    eng.load_deck(d)
    eng.reset()
    eng.analyze_dc(pe.ffi.MODE_DC)
    v_a = eng.solution()[0][2]
    eng.update_param(pe.ffi.NMOS, 0, 2, [1.5])      # Vth 1.0 -> 1.5: less current, higher drain voltage
    eng.analyze_dc(pe.ffi.MODE_DC)
    v_b = eng.solution()[0][2]
    Kp, lam = 2e-3, 0.02
    # saturation: (5 - vd) / 2000 = 0.5 Kp Vov^2 (1 + lam vd)
    for v, vov in ((v_a, 1.0), (v_b, 0.5)):
        assert abs((5.0 - v) / 2000.0 - 0.5 * Kp * vov * vov * (1.0 + lam * v)) < 1e-9
    assert v_b > v_a


def test_switch_toggle_and_param_updates(eng):
    """test/0005.models/cutthrough.cpp idea: the same resident circuit, switch opened / closed through update_param;
    a controlled source's gain and a generator's level changed the same way."""
    d = pe.deck.Deck()
    d.n_nodes = 4
    d.add("VDC", (1, 0), 2.0)
    d.add("SW", (1, 2), 0.0)
    d.add("R", (2, 0), 1000.0)
    d.add("VCVS", (3, 0, 2, 0), 3.0)
    d.add("R", (3, 0), 500.0)
    d.add("SQR", (4, 0), 4.0, 1.0, 1e3, 0.5, 0.0)
    d.add("R", (4, 0), 100.0)
    eng.set_options(g_min=0.0)
    eng.load_deck(d)
    eng.reset()
    eng.analyze_dc(pe.ffi.MODE_DC)
    x = eng.solution()[0]
    assert abs(x[1] - 2.0 * 1000.0 / (1e12 + 1000.0)) < 1e-15 and abs(x[3] - 4.0) < 1e-12      # open: r_open 1e12
    eng.update_param(pe.ffi.SWITCH, 0, 0, [1.0])
    eng.update_param(pe.ffi.VCVS, 0, 0, [-1.5])
    eng.update_param(pe.ffi.VGEN, 0, 1, [7.0])                                              # column 1 = Vh
    eng.analyze_dc(pe.ffi.MODE_DC)
    x = eng.solution()[0]
    assert abs(x[1] - 2.0) < 1e-12 and abs(x[2] + 3.0) < 1e-12 and abs(x[3] - 7.0) < 1e-12
    eng.set_options(g_min=0.0, r_open=1e6)
    eng.update_param(pe.ffi.SWITCH, 0, 0, [0.0])
    eng.analyze_dc(pe.ffi.MODE_DC)
    assert abs(eng.solution()[0][1] - 2.0 * 1000.0 / (1e6 + 1000.0)) < 1e-12


def test_bridge_gmin0_fails_like_reference(eng):
    """g_min = 0: the bridge becomes singular when all four diodes are off; the reference gives up at step 80.
    The step at which a near-singular pivot breaks Newton is implementation-defined: accept 76..82, and identical
    results before."""
    meta, gx, deck = golden("bridge_gmin0_fail")
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert max_err(snaps[:1, 0, :], gx[:1], *NL) <= 1.0
    assert 76 <= fail <= 82
    st = eng.state()
    assert st["status"][0] in (pe.ffi.ERR_SINGULAR, pe.ffi.ERR_NO_CONVERGENCE)
    assert abs(st["t"][0] - (fail - 1) * meta["dt"]) < 1e-12   # tr_duration rolled back (circuit.h:249-253)


def test_failed_solve_is_not_sticky(eng, oracle_mod):
    """circuit.h:242-254: after a failed transient the next analyze() tries again from the rolled-back state.  The g_min = 0
    bridge fails; with g_min raised to 1e-12 the SAME resident circuit continues (no reset) and then follows the oracle started
    from the same state."""
    meta, gx, deck = golden("bridge_gmin0_fail")
    eng.set_options(g_min=0.0)
    eng.load_deck(deck)
    eng.reset()
    st = eng.analyze_tr(meta["dt"], meta["steps"], check=False)
    assert st["rc"] in (pe.ffi.ERR_SINGULAR, pe.ffi.ERR_NO_CONVERGENCE)
    s0 = eng.state()
    n_ok = int(s0["steps"][0])
    assert abs(float(s0["t"][0]) - n_ok * meta["dt"]) < 1e-12
    eng.set_options(g_min=1e-12)
    st2 = eng.analyze_tr(meta["dt"], 200, check=False)
    assert st2["rc"] == 0 and st2["steps"] == 200
    s1 = eng.state()
    assert s1["status"][0] == 0 and abs(float(s1["t"][0]) - (n_ok + 200) * meta["dt"]) < 1e-12
    # the continuation is a valid transient of the g_min = 1e-12 bridge: a fresh engine run with that g_min from t = 0 reaches the
    # same state (the first n_ok steps differ by g_min * v ~ 1e-11 A only)
    x_cont = eng.solution()[0].copy()
    e2 = pe.ffi.Engine(device=0)
    try:
        e2.set_options(g_min=1e-12)
        e2.load_deck(deck)
        e2.reset()
        e2.analyze_tr(meta["dt"], n_ok + 200)
        assert max_err(x_cont, e2.solution()[0], 1e-6, 1e-5) <= 1.0
    finally:
        e2.close()


def test_floating_network_reports_singular(eng):
    """test/0003.circuits/operations.cpp: R || C with no ground, DC.  Must not crash; this engine reports the
    singular system (the reference's Eigen path happens to return x = 0 for the all-zero right-hand side)."""
    meta, gx, deck = golden("floating_rc_dc")
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert fail == 0 and eng.state()["status"][0] == pe.ffi.ERR_SINGULAR


def test_batched_sweep_matches_per_instance_reference(eng):
    """C5 exact per-instance parity: seeds 1..8 of the 32x32 mesh as ONE batch of 8 vs 8 reference runs."""
    seeds = list(range(1, 9))
    for nonlinear, tag, tol in ((False, "lin", LIN), (True, "nl", NL)):
        deck, r, c = pe.deck.rc_mesh_params(32, 32, seeds, nonlinear)
        eng.set_options(g_min=0.0)
        eng.load_deck(deck, batch=8, overrides={"R": r[:, :, None], "C": c[:, :, None]})
        eng.reset()
        eng.analyze_tr(1e-10, 100)
        x = eng.solution()
        for k, sd in enumerate(seeds):
            name = f"mesh32_{tag}" if sd == 1 else f"mesh32_{tag}_seed{sd}"
            meta, gx, _ = golden(name)
            ref = gx[meta["snap_steps"].index(100)]
            assert max_err(x[k], ref, *tol) <= 1.0, (tag, sd)


def test_mesh100_second_seed(eng):
    meta, gx, deck = golden("mesh100_lin_seed2")
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    assert fail == -1 and max_err(snaps[:, 0, :], gx, *LIN) <= 1.0


def test_fresh_seeds_against_oracle(eng, oracle_mod):
    """Inputs the fixtures have never seen: HIP path vs the CPU oracle (checker) on the same seeded decks."""
    for seed, nonlinear, tol in ((11, False, LIN), (12, True, NL)):
        deck = pe.deck.rc_mesh(24, 24, seed, nonlinear)
        o = oracle_mod.Oracle(deck)
        o.analyze_tr(2e-10, 40)
        eng.set_options(g_min=0.0)
        eng.load_deck(deck)
        eng.reset()
        eng.analyze_tr(2e-10, 40)
        assert max_err(eng.solution()[0], o.x, *tol) <= 1.0
        assert list(eng.newton_trace()) == o.newton_iters


def test_solve_csr_real_seam(eng, oracle_mod):
    """Drop-in for cuda_sparse_lu::solve_csr_real (cuda_sparse_lu.h:465-473): host CSR in, x out; residual and
    agreement with a CPU sparse LU; reference acceptance is max|dx| < 1e-6 (test/0013.cuda/cuda_random_links_correctness.cu:129)."""
    import scipy.sparse.linalg as spla
    deck = pe.deck.rc_mesh(40, 40, 5, False)
    o = oracle_mod.Oracle(deck)
    o.update_tr_step(1e-10)
    o.t = 1e-10
    A, b = o.assemble("TR")
    A = A.tocsr()
    A.sort_indices()
    x, tm = eng.solve_csr(A.shape[0], A.indptr, A.indices, A.data, b, copy_pattern=True)
    xr = spla.splu(A.tocsc()).solve(b)
    assert np.max(np.abs(x - xr)) < 1e-9
    assert np.max(np.abs(A @ x - b)) < 1e-12
    x2, _ = eng.solve_csr(A.shape[0], A.indptr, A.indices, A.data * 2.0, b, copy_pattern=False)  # cached pattern
    assert np.max(np.abs(2.0 * x2 - xr)) < 1e-9


def test_solve_csr_real_seam_large_front(eng):
    """The seam with one dense front of order 450 (> the default LDS reserve behind the panels): the same LDS-fit escalation
    as the resident circuit path (round-1 advisor finding)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(7)
    n = 450
    A = sp.csr_matrix(rng.standard_normal((n, n)) + n * np.eye(n))
    A.sort_indices()
    b = rng.standard_normal(n)
    x, _ = eng.solve_csr(n, A.indptr, A.indices, A.data, b, copy_pattern=True)
    assert np.max(np.abs(A @ x - b)) < 1e-9


def complex_csr_of_ac_point(o, omega):
    """The complex CSR the reference's solve_once would hand to its solver at this AC point (sorted columns, circuit.h:1171-1226)."""
    import scipy.sparse as sp
    A, rhs = o.stamp_ac(omega)
    keys = sorted(A.keys())
    M = sp.csr_matrix((np.array([A[q] for q in keys], dtype=complex), (np.array([q[0] for q in keys]), np.array([q[1] for q in keys]))),
                      shape=(o.rows, o.rows))
    M.sort_indices()
    return M, rhs


def test_solve_csr_complex_seam(eng, oracle_mod):
    """Drop-in for the complex twin cuda_sparse_lu::solve_csr_timed (cuda_sparse_lu.h:304-312, called at circuit.h:1332 when the stamped
    system is not all-real): the assembled complex system of every frequency point of the real-reference golden `ac_rlc_diode_acop`
    goes through pe_hip_solve_csr_complex -- first call analyses the pattern, the later ones reuse it (copy_pattern = 0) -- and must
    land on the golden phasors; then a 1 600-node R-C mesh at one frequency against a CPU complex sparse LU."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    meta, gx, deck = golden("ac_rlc_diode_acop")
    g = golden_complex(meta, gx)
    o = oracle_mod.Oracle(deck)
    o.g_min = meta["gmin"]
    o.prepare()
    assert o.solve("OP") >= 0
    for k, w in enumerate(meta["omegas"]):
        M, rhs = complex_csr_of_ac_point(o, w)
        x, tm = eng.solve_csr_complex(o.rows, M.indptr, M.indices, M.data, rhs, copy_pattern=(k == 0))
        assert np.all(np.abs(x - g[k]) <= 1e-9 + 1e-6 * np.abs(g[k])), (w, x, g[k])
    # G + j omega C of a 40 x 40 mesh (diagonally dominant in modulus, entries of both kinds in every row), cached-pattern second call
    mdeck = pe.deck.rc_mesh(40, 40, 5, False)
    mo = oracle_mod.Oracle(mdeck)
    mo.update_tr_step(1e-10)
    mo.t = 1e-10
    A, b = mo.assemble("TR")
    A = A.tocsr()
    A.sort_indices()
    rng = np.random.default_rng(3)
    Z = sp.csr_matrix((A.data * (1.0 + 1j * rng.uniform(-2.0, 2.0, A.nnz)), A.indices, A.indptr), shape=A.shape)
    bz = b * (1.0 - 0.5j) + 1j * rng.standard_normal(len(b)) * 1e-3
    x, _ = eng.solve_csr_complex(Z.shape[0], Z.indptr, Z.indices, Z.data, bz, copy_pattern=True)
    xr = spla.splu(Z.tocsc()).solve(bz)
    assert np.max(np.abs(x - xr)) <= 1e-9 * max(1.0, np.max(np.abs(xr)))
    assert np.max(np.abs(Z @ x - bz)) <= 1e-12 * max(1.0, np.max(np.abs(bz)))
    x2, _ = eng.solve_csr_complex(Z.shape[0], Z.indptr, Z.indices, Z.data * (2.0 - 1.0j), bz, copy_pattern=False)
    assert np.max(np.abs((2.0 - 1.0j) * x2 - xr)) <= 1e-9 * max(1.0, np.max(np.abs(xr)))
    # a singular complex system is refused, like the reference's `return false`
    S = sp.csr_matrix(np.array([[1.0 + 1.0j, 2.0 + 2.0j], [2.0 + 2.0j, 4.0 + 4.0j]]))
    with pytest.raises(pe.ffi.PeHipError):
        eng.solve_csr_complex(2, S.indptr, S.indices, S.data, np.array([1.0, 1.0j]), copy_pattern=True)


def test_stamped_matrix_matches_oracle(eng, oracle_mod):
    """The device-side MNA gather reproduces the oracle's assembled matrix entry by entry (values and pattern)."""
    deck = pe.deck.rc_mesh(16, 16, 9, True)
    eng.set_options(g_min=1e-12)
    eng.load_deck(deck)
    eng.reset()
    eng.analyze_tr(1e-10, 1)
    rp, ci, va, rhs = eng.matrix(0)
    import scipy.sparse as sp
    Ad = sp.csr_matrix((va, ci, rp), shape=(eng.rows, eng.rows))
    o = oracle_mod.Oracle(deck, g_min=1e-12)
    o.analyze_tr(1e-10, 1)
    # the oracle's last stamp used the last Newton iterate's predecessor; rebuild at the same point:
    o2 = oracle_mod.Oracle(deck, g_min=1e-12)
    o2.prepare()
    o2.update_tr_step(1e-10)
    o2.t = 1e-10
    for _ in range(o.newton_iters[-1]):
        A, b = o2.assemble("TR")
        import scipy.sparse.linalg as spla
        o2.x = spla.splu(A).solve(b)
    diff = (Ad - A.tocsr())
    assert abs(diff).max() <= 1e-12 * abs(A).max()
    assert np.max(np.abs(rhs - b)) <= 1e-12 * max(1.0, np.max(np.abs(b)))


# ---- full-size properties (BASELINE config C3: M10k) -----------------------------------------------------
def test_full_size_properties(eng):
    """M10k linear, 40 steps: (a) linearity: doubling the source doubles every voltage; (b) DC limit: after
    the transient has died out (dt*steps >> RC? no -- checked via KCL instead) the stamped system is satisfied:
    ||A x - b||_inf small for the last step's matrix; (c) a batch of identical instances gives bitwise-identical
    results (determinism: no atomics, fixed summation order)."""
    import scipy.sparse as sp
    deck = pe.deck.rc_mesh(100, 100, 1, False)
    eng.set_options(g_min=0.0)
    eng.load_deck(deck, batch=3)
    eng.reset()
    nV = 0  # VDC index 0
    eng.update_param(pe.ffi.VDC, nV, 0, [1.0, 2.0, 1.0])
    eng.analyze_tr(1e-10, 40)
    x = eng.solution()
    assert np.array_equal(x[0], x[2])                                   # (c) bitwise
    assert np.max(np.abs(x[1] - 2.0 * x[0])) <= 1e-12 * np.max(np.abs(x[1]))   # (a)
    rp, ci, va, rhs = eng.matrix(0)
    A = sp.csr_matrix((va, ci, rp), shape=(eng.rows, eng.rows))
    assert np.max(np.abs(A @ x[0] - rhs)) <= 1e-12 * max(1.0, np.max(np.abs(rhs)))  # (b) residual of the last solve
    info = eng.info()
    assert info["rows"] == 10002 and info["nnz_a"] == 49605 and info["nnz_lu"] < 645757


def test_reuse_factor_option_gives_same_answer(eng):
    """refactor_every_solve = 0 (legitimate for a linear circuit at constant dt) changes speed only."""
    deck = pe.deck.rc_mesh(32, 32, 4, False)
    out = []
    for refac in (1, 0):
        eng.set_options(g_min=0.0, refactor_every_solve=refac)
        eng.load_deck(deck)
        eng.reset()
        eng.analyze_tr(1e-10, 50)
        out.append(eng.solution()[0])
    eng.set_options(g_min=0.0, refactor_every_solve=1)
    assert np.max(np.abs(out[0] - out[1])) <= 1e-13


def test_dc_then_tr_switches_symbolic(eng, oracle_mod):
    """OP solve followed by a transient on the same engine (TROP-like use, circuit.h:257-289): the static and the
    TR pivot matchings differ (inductor D = 0 in DC), results must still match the oracle."""
    deck = pe.deck.rlc_series_vl()
    o = oracle_mod.Oracle(deck)
    o.analyze_tr(1e-6, 50, trop=True)
    eng.set_options(g_min=0.0)
    eng.load_deck(deck)
    eng.reset()
    eng.analyze_dc(pe.ffi.MODE_TROP)
    eng.analyze_tr(1e-6, 50)
    assert max_err(eng.solution()[0], o.x, *LIN) <= 1.0


# ---- multi-workgroup schedule (one circuit spread over several workgroups; knob PHY_ENGINE_HIP_PARTS) ----------
@pytest.mark.parametrize("name,tol,parts", [("mesh32_nl", NL, 6), ("mesh32_lin", LIN, 16), ("ladder_c1", LIN, 4), ("mesh100_nl", NL, 1),
                                            ("mesh100_nl", NL, 13), ("bridge_c2", NL, 2), ("mesh100_nl", NL, -1), ("mesh32_nl", NL, -1)])
def test_golden_parity_multi_workgroup(name, tol, parts, monkeypatch):
    """Same goldens through the split schedule (parts + top levels, one launch per phase: pe_engine_newton.cpp run_m2_tr) and, with
    parts = -1, through the resident single-workgroup kernel (PHY_ENGINE_HIP_SPLIT=0 / =1 override the size rule:
    circuits of >= 3000 rows run the split schedule, smaller ones the resident kernel)."""
    if parts < 0:
        monkeypatch.setenv("PHY_ENGINE_HIP_SPLIT", "0" if name.startswith("mesh100") else "1")
    else:
        monkeypatch.setenv("PHY_ENGINE_HIP_PARTS", str(parts))
    e = pe.ffi.Engine(device=0)
    try:
        meta, gx, deck = golden(name)
        snaps, trace, fail = run_engine_case(e, meta, deck)
        assert fail == -1 and len(snaps) == len(gx)
        assert max_err(snaps[:, 0, :], gx, *tol) <= 1.0
        assert list(trace) == meta["newton_iters"]
    finally:
        e.close()


def test_multi_workgroup_batch_matches_single_workgroup(monkeypatch):
    """A 4-instance M10k sweep: the split multi-workgroup schedule (default for circuits of this size) and the resident
    single-workgroup kernel agree to rounding (different elimination trees => different summation order, so not bitwise)."""
    deck = pe.deck.rc_mesh(100, 100, 1, True)
    out = []
    for parts in (1, 0):
        if parts:
            monkeypatch.setenv("PHY_ENGINE_HIP_SPLIT", "0")
        else:
            monkeypatch.delenv("PHY_ENGINE_HIP_SPLIT", raising=False)
        e = pe.ffi.Engine(device=0)
        e.set_options(g_min=0.0)
        e.load_deck(deck, batch=4)
        e.reset()
        e.update_param(pe.ffi.VAC, 0, 0, [2.0, 3.0, 1.0, 2.5])   # amplitude per instance
        e.analyze_tr(1e-10, 10)
        out.append((e.solution().copy(), e.state()["iters"].copy()))
        e.close()
    assert np.array_equal(out[0][1], out[1][1])
    assert np.max(np.abs(out[0][0] - out[1][0])) <= 1e-9 * np.max(np.abs(out[0][0]))


@pytest.mark.parametrize("geometry", ["64", "128", "256", "1024"])
@pytest.mark.parametrize("split", ["0", "auto"])
def test_launch_geometries_on_large_circuit(geometry, split, monkeypatch):
    """M10k golden under every launch geometry the batch-size policy selects (symbolic_options in pe_engine_policy.cpp: 8 wavefronts x
    1 workgroup per CU, 4 x 4 with 4 or 8 parts), each in the split schedule and in the resident kernel; 2 identical instances."""
    monkeypatch.setenv("PHY_ENGINE_HIP_GEOMETRY_BATCH", geometry)
    if split != "auto":
        monkeypatch.setenv("PHY_ENGINE_HIP_SPLIT", split)
    e = pe.ffi.Engine(device=0)
    try:
        meta, gx, deck = golden("mesh100_nl")
        snaps, trace, fail = run_engine_case(e, meta, deck, batch=2)
        assert fail == -1 and len(snaps) == len(gx)
        for b in range(2):
            assert max_err(snaps[:, b, :], gx, *NL) <= 1.0
        assert list(trace) == meta["newton_iters"]
        assert (e.info()["n_parts"] > 1) == (split == "auto")
    finally:
        e.close()


def test_reuse_factor_multi_workgroup():
    """Same as test_reuse_factor_option_gives_same_answer on M10k (batch 1 => the multi-workgroup schedule)."""
    deck = pe.deck.rc_mesh(100, 100, 3, False)
    out = []
    e = pe.ffi.Engine(device=0)
    for refac in (1, 0):
        e.set_options(g_min=0.0, refactor_every_solve=refac)
        e.load_deck(deck)
        e.reset()
        e.analyze_tr(1e-10, 12)
        out.append(e.solution()[0])
    e.close()
    assert np.max(np.abs(out[0] - out[1])) <= 1e-13


# ---- small-signal AC (complex system solved in real-equivalent form by the same kernels) ----------
@pytest.mark.parametrize("name", ["ac_rc_lowpass", "ac_rlc_diode_acop", "ac_linear_mix", "ac_nmos_amp"])
def test_ac_golden_parity(eng, name):
    meta, gx, deck = golden(name)
    xs = run_ac_case(eng, meta, deck)
    g = golden_complex(meta, gx)
    assert xs.shape == g.shape
    assert np.all(np.abs(xs - g) <= 1e-9 + 1e-6 * np.abs(g))


def test_ac_sweep_batch_of_instances(eng):
    """AC over a batch: instance b of a C sweep has its corner at omega = 1 / (R C_b); |v_out| = 1 / sqrt(2) there."""
    d = pe.deck.ac_rc_lowpass()
    caps = np.array([1e-6, 2e-6, 5e-7, 1e-7])
    eng.set_options(g_min=0.0)
    eng.load_deck(d, batch=4, overrides={"C": caps[:, None, None]})
    eng.reset()
    for b, c in enumerate(caps):
        x, rc = eng.analyze_ac(1.0 / (1000.0 * c))
        assert rc == 0 and abs(abs(x[b][1]) - 2.0 ** -0.5) < 1e-12


def test_checkpoint_resume_is_bit_exact(eng):
    """a transient interrupted by checkpoint -> new engine -> restore continues bit-identically
    (non-linear mesh with diodes' limiting / transit state, a 3-instance sweep)."""
    deck, r, c = pe.deck.rc_mesh_params(32, 32, [1, 2, 3], True)
    ov = {"R": r[:, :, None], "C": c[:, :, None]}
    eng.set_options(g_min=0.0)
    eng.load_deck(deck, batch=3, overrides=ov)
    eng.reset()
    eng.analyze_tr(1e-10, 60)
    want, want_state = eng.solution().copy(), eng.state()
    eng.reset()
    eng.analyze_tr(1e-10, 25)
    blob = eng.checkpoint()
    e2 = pe.ffi.Engine(device=0)
    e2.set_options(g_min=0.0)
    e2.load_deck(deck, batch=3, overrides=ov)
    e2.restore(blob)
    e2.analyze_tr(1e-10, 35)
    got, got_state = e2.solution(), e2.state()
    e2.close()
    assert np.array_equal(got, want)
    assert np.array_equal(got_state["iters"], want_state["iters"]) and np.array_equal(got_state["steps"], want_state["steps"])
    assert np.array_equal(got_state["t"], want_state["t"])
    # a checkpoint of another circuit is refused
    e3 = pe.ffi.Engine(device=0)
    e3.load_deck(pe.deck.rc_step())
    with pytest.raises(pe.ffi.PeHipError):
        e3.restore(blob)
    e3.close()


def test_large_circuit_40k_nodes(oracle_mod):
    """200 x 200 diode mesh (40 002 rows, fronts up to 300 rows, 22 top levels): single instance (multi-workgroup schedule)
    and a 2-instance batch against the oracle."""
    deck = pe.deck.rc_mesh(200, 200, 1, True)
    o = oracle_mod.Oracle(deck)
    o.analyze_tr(1e-10, 3)
    for batch in (1, 2):
        e = pe.ffi.Engine(device=0)
        e.set_options(g_min=0.0)
        e.load_deck(deck, batch=batch)
        e.reset()
        st = e.analyze_tr(1e-10, 3)
        x = e.solution()
        assert st["newton_iters"] == batch * sum(o.newton_iters)
        for b in range(batch):
            assert max_err(x[b], o.x, *NL) <= 1.0
        assert e.info()["max_front"] >= 250
        e.close()


@pytest.mark.parametrize("geometry", [None, "1024", "500"])
def test_dense_circuit_single_large_front(oracle_mod, geometry, monkeypatch):
    """A complete graph of 400 resistors-nodes: ONE front of order 401, far beyond what fits LDS whole at any launch geometry
    (panel layout with p of 8-24, long chain of links); DC against the oracle, for the three workgroup geometries."""
    if geometry:
        monkeypatch.setenv("PHY_ENGINE_HIP_GEOMETRY_BATCH", geometry)
    n = 400
    d = pe.deck.Deck()
    d.n_nodes = n
    rng = np.random.default_rng(1)
    for i in range(1, n + 1):
        for j in range(i + 1, n + 1):
            d.add("R", (i, j), float(100.0 + 900.0 * rng.random()))
    d.add("VDC", (1, 0), 1.0)
    d.add("R", (n, 0), 50.0)
    o = oracle_mod.Oracle(d)
    assert o.analyze_dc("DC")
    e = pe.ffi.Engine(device=0)
    e.set_options(g_min=0.0)
    e.load_deck(d, batch=2)
    e.reset()
    e.analyze_dc(pe.ffi.MODE_DC)
    x = e.solution()
    assert e.info()["max_front"] == n + 1
    for b in range(2):
        assert max_err(x[b], o.x, *LIN) <= 1.0
    e.close()


# ---- circuits of >= 3000 rows other than the RC mesh through the split schedule (the default there) ----------
def test_ac_on_large_circuit_split_schedule(oracle_mod):
    """Small-signal AC of a 45 x 45 diode mesh: the real-equivalent system has 4 054 rows, so the AC engine runs the split
    schedule (parts + top levels) with its iterative refinement; three frequencies around the mesh's corner against the oracle."""
    deck = pe.deck.rc_mesh(45, 45, 1, True)
    omegas = [2e8, 2e9, 2e10]
    o = oracle_mod.Oracle(deck)
    ref = o.analyze_ac(omegas)
    assert ref is not None and all(r is not None for r in ref)
    e = pe.ffi.Engine(device=0)
    try:
        e.set_options(g_min=0.0)
        e.load_deck(deck)
        e.reset()
        e.analyze_dc(pe.ffi.MODE_OP)
        for w, r in zip(omegas, ref):
            x, rc = e.analyze_ac(w)
            assert rc == 0
            assert np.all(np.abs(x[0] - r) <= 1e-9 + 1e-6 * np.abs(r)), w
    finally:
        e.close()


def test_large_rlc_line_split_schedule(oracle_mod):
    """A 1 100-section L-C line with series loss, driven by a 1 V step through 50 ohm: 1 101 nodes + 1 100 inductor branches + the
    source = 3 303 rows (zero diagonal entries on every branch row: static pivoting does real work), 40 transient steps, two
    instances with different inductances, against the oracle."""
    n = 1100
    d = pe.deck.Deck()
    d.n_nodes = 2 * n + 2
    d.add("VDC", (1, 0), 1.0)
    d.add("R", (1, 2), 50.0)
    for k in range(n):
        a, mid, b = 2 + 2 * k, 3 + 2 * k, 4 + 2 * k
        d.add("L", (a, mid), 2.5e-9)
        d.add("R", (mid, b), 0.05)
        d.add("C", (b, 0), 1e-12)
    d.add("R", (2 * n + 2, 0), 50.0)
    scale = np.array([1.0, 1.3])
    refs = []
    for s in scale:
        dd = pe.deck.Deck()
        dd.n_nodes = d.n_nodes
        for kind, pins, par in d.devices:
            vals = list(par)
            if kind == "L":
                vals[0] *= s
            dd.add(kind, pins, *vals)
        o = oracle_mod.Oracle(dd)
        o.analyze_tr(2e-11, 40)
        refs.append(o.x.copy())
    e = pe.ffi.Engine(device=0)
    try:
        e.set_options(g_min=0.0)
        e.load_deck(d, batch=2, overrides={"L": (2.5e-9 * scale)[:, None, None] * np.ones((2, n, 1))})
        e.reset()
        st = e.analyze_tr(2e-11, 40)
        x = e.solution()
        assert e.info()["rows"] >= 3000 and e.info()["n_parts"] > 1 and st["rc"] == 0
        for b in range(2):
            assert max_err(x[b], refs[b], *LIN) <= 1.0
    finally:
        e.close()


def test_floating_node_in_large_circuit_reports_singular_split_schedule(oracle_mod):
    """A 60 x 60 linear mesh (3 604 rows: split schedule) plus two extra nodes joined by one resistor and nothing else: the matrix is
    structurally fine and numerically singular ([[g, -g], [-g, g]]), the reference's factorisation fails (circuit.h:1517).  Both
    instances must report it -- the bad-pivot flag of a front travels through the per-front check, the workgroup's flag word and
    the host's Newton loop -- and a healthy circuit loaded afterwards on the same engine must solve."""
    deck = pe.deck.rc_mesh(60, 60, 1, False)
    n_extra = deck.n_nodes + 2
    deck.n_nodes = n_extra
    deck.add("R", (n_extra - 1, n_extra), 1000.0)
    o = oracle_mod.Oracle(deck)
    assert not o.analyze_dc("DC")
    e = pe.ffi.Engine(device=0)
    try:
        e.set_options(g_min=0.0)
        e.load_deck(deck, batch=2)
        e.reset()
        st = e.analyze_dc(pe.ffi.MODE_DC, check=False)
        assert st["rc"] != 0 and e.info()["n_parts"] > 1
        assert list(e.state()["status"]) == [pe.ffi.ERR_SINGULAR, pe.ffi.ERR_SINGULAR]
        good = pe.deck.rc_mesh(60, 60, 1, False)
        og = oracle_mod.Oracle(good)
        assert og.analyze_dc("DC")
        e.load_deck(good, batch=2)
        e.reset()
        e.analyze_dc(pe.ffi.MODE_DC)
        assert max_err(e.solution()[1], og.x, *LIN) <= 1.0
    finally:
        e.close()


def test_failing_transient_rolls_back_in_split_schedule(monkeypatch):
    """test_bridge_gmin0_fails_like_reference through the split schedule (forced on the small circuit): the host Newton loop of
    run_m2_tr must stop at the same kind of failure, keep the results before it and roll the time back (circuit.h:249-253)."""
    monkeypatch.setenv("PHY_ENGINE_HIP_SPLIT", "1")
    e = pe.ffi.Engine(device=0)
    try:
        meta, gx, deck = golden("bridge_gmin0_fail")
        snaps, trace, fail = run_engine_case(e, meta, deck)
        assert max_err(snaps[:1, 0, :], gx[:1], *NL) <= 1.0
        assert 76 <= fail <= 82
        st = e.state()
        assert st["status"][0] in (pe.ffi.ERR_SINGULAR, pe.ffi.ERR_NO_CONVERGENCE)
        assert abs(st["t"][0] - (fail - 1) * meta["dt"]) < 1e-12
    finally:
        e.close()


@pytest.mark.parametrize("name,tol", [("generators_tr", LIN), ("generators_trop", LIN), ("iac_rc_trop", LIN), ("coupled_l_k09_tr", LIN), ("coupled_l_k09_trop", LIN),
                                      ("controlled_mix_tr", NL), ("cmos_inverter_tr", NL), ("bjt_amp_tr", NL), ("relay_ramp_tr", NL), ("center_tap_ratio", NL),
                                      ("nmos_triode_op", NL)])
def test_stampers_through_split_schedule(name, tol, monkeypatch):
    """The per-phase kernels of the split schedule (k_m2_eval / k_m2_companion: time-dependent sources, coupled inductors, relay
    state, three-pin devices) on the stamper goldens, forced on these small circuits: same results and Newton trajectory."""
    monkeypatch.setenv("PHY_ENGINE_HIP_SPLIT", "1")
    e = pe.ffi.Engine(device=0)
    try:
        meta, gx, deck = golden(name)
        snaps, trace, fail = run_engine_case(e, meta, deck)
        assert fail == -1 and len(snaps) == len(gx)
        assert max_err(snaps[:, 0, :], gx, *tol) <= 1.0
        assert list(trace) == meta["newton_iters"]
    finally:
        e.close()


def test_checkpoint_and_ac_in_split_schedule(monkeypatch):
    """Checkpoint -> new engine -> restore continues bit-identically, and the AC goldens, with the split schedule forced on."""
    monkeypatch.setenv("PHY_ENGINE_HIP_SPLIT", "1")
    deck, r, c = pe.deck.rc_mesh_params(32, 32, [1, 2, 3], True)
    ov = {"R": r[:, :, None], "C": c[:, :, None]}
    e1 = pe.ffi.Engine(device=0)
    e2 = pe.ffi.Engine(device=0)
    try:
        e1.set_options(g_min=0.0)
        e1.load_deck(deck, batch=3, overrides=ov)
        e1.reset()
        e1.analyze_tr(1e-10, 40)
        want = e1.solution().copy()
        e1.reset()
        e1.analyze_tr(1e-10, 15)
        blob = e1.checkpoint()
        e2.set_options(g_min=0.0)
        e2.load_deck(deck, batch=3, overrides=ov)
        e2.restore(blob)
        e2.analyze_tr(1e-10, 25)
        assert np.array_equal(e2.solution(), want)
        for name in ("ac_rlc_diode_acop", "ac_linear_mix"):
            meta, gx, d = golden(name)
            xs = run_ac_case(e1, meta, d)
            g = golden_complex(meta, gx)
            assert np.all(np.abs(xs - g) <= 1e-9 + 1e-6 * np.abs(g)), name
    finally:
        e1.close()
        e2.close()


def test_c5_sweep_1024_instances_parity_and_statistics():
    """Config C5 AT SIZE (BASELINE.json configs[4]): the 1024-instance Monte-Carlo sweep of the M10k-NL mesh, seeds
    1..1024, as ONE batch on the GPU -- the very workload bench.py times.
      (i)   exact per-instance parity: instances seed = 1..8 against eight runs of the real reference at steps 10 and 100
            (tests/golden/mesh100_nl{,_seed2..8}), NL tolerance (Newton's stop rule is 1e-3 relative);
      (ii)  per-node {sum, sum of squares, min, max} over the instances seed = 1..32 after 10 steps against the same statistics of
            32 reference runs (tests/golden/mesh100_nl_stats32: the 32-instance CPU subset of SURVEY.md 8d);
      (iii) the device-side statistics kernel over all 1024 instances (pe_hip_sweep_statistics, the payload of the sweep's one
            all-reduce) against numpy on the downloaded solutions, and the full sweep's mean against the subset's within sampling error."""
    import json
    B = 1024
    seeds = list(range(1, B + 1))
    deck, r, c = pe.deck.rc_mesh_params(100, 100, seeds, True)
    e = pe.ffi.Engine(device=0)
    try:
        e.set_options(g_min=0.0)
        e.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]})
        e.reset()
        st = e.analyze_tr(1e-10, 10)
        assert st["n_failed"] == 0 and st["steps"] == 10 * B
        x10 = e.solution(0, 32)
        stats_dev = e.sweep_statistics()
        x_all = e.solution()
        # (iii) device kernel == numpy on the same data (different summation order only)
        ref_all = np.stack([x_all.sum(axis=0), (x_all * x_all).sum(axis=0), x_all.min(axis=0), x_all.max(axis=0)])
        assert np.max(np.abs(stats_dev[2:] - ref_all[2:])) == 0.0
        assert max_err(stats_dev[:2], ref_all[:2], 1e-9, 1e-12) <= 1.0
        # (ii) the 32-instance subset against 32 reference runs
        meta = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "mesh100_nl_stats32.json")))
        gs = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "mesh100_nl_stats32.bin")).reshape(4, -1)
        assert meta["seeds"] == list(range(1, 33)) and meta["steps"] == 10 and gs.shape[1] == e.rows
        mine = np.stack([x10.sum(axis=0), (x10 * x10).sum(axis=0), x10.min(axis=0), x10.max(axis=0)])
        assert max_err(mine[0], gs[0], 32 * NL[0], NL[1]) <= 1.0
        assert max_err(mine[1], gs[1], 32 * NL[0], 2 * NL[1]) <= 1.0
        assert max_err(mine[2:], gs[2:], *NL) <= 1.0
        # the whole sweep's mean lies within sampling error of the subset's mean (6 standard errors of the 32-sample mean)
        mean32 = gs[0] / 32.0
        var32 = np.maximum(gs[1] / 32.0 - mean32 * mean32, 0.0)
        mean_all = stats_dev[0] / B
        assert np.mean(np.abs(mean_all - mean32) <= 6.0 * np.sqrt(var32 / 32.0) + 1e-6) > 0.99
        # (i) seeds 1..8 at step 10, then at step 100
        golds = [golden("mesh100_nl" if sd == 1 else f"mesh100_nl_seed{sd}") for sd in range(1, 9)]
        for k, (gm, gx, _) in enumerate(golds):
            assert max_err(x10[k], gx[gm["snap_steps"].index(10)], *NL) <= 1.0, f"seed {k + 1} at step 10"
        st = e.analyze_tr(1e-10, 90)
        assert st["n_failed"] == 0
        x100 = e.solution(0, 8)
        for k, (gm, gx, _) in enumerate(golds):
            assert max_err(x100[k], gx[gm["snap_steps"].index(100)], *NL) <= 1.0, f"seed {k + 1} at step 100"
    finally:
        e.close()


def test_residual_safety_net_paths_on_gpu():
    """The residual safety net of the static-pivot LU on the device (k_m2_residual / k_m2_refine_apply / k_m2_retest and the resident
    kernel's check).  An unreachable residual_tol drives every stage: detection in the resident kernel, the switch to the
    host-driven schedule, two refinement rounds, a re-match on the instance's own values, and finally PE_HIP_ERR_INACCURATE with
    the step rolled back; the failure is not sticky.  With the default tolerance the same transient must run without any
    intervention (no false alarm) -- as every other golden test implicitly checks -- and a 2-instance sweep with an open /
    closed switch and a 1e6 resistor scale must match the oracle per instance."""
    meta, gx, deck = golden("mesh32_nl")
    g10 = gx[meta["snap_steps"].index(10)]
    e = pe.ffi.Engine(device=0)
    try:
        e.set_options(g_min=0.0, residual_tol=1e-30)
        e.load_deck(deck)
        e.reset()
        st = e.analyze_tr(1e-10, 10, check=False)
        assert st["rc"] == pe.ffi.ERR_INACCURATE and st["steps"] == 0
        s = e.state()
        assert s["t"][0] == 0.0 and s["status"][0] == pe.ffi.ERR_INACCURATE
        sn = e.safety_net()
        assert sn["careful"] and sn["rematched"] >= 1
        e.set_options(g_min=0.0, residual_tol=0.0)
        st = e.analyze_tr(1e-10, 10, check=False)          # careful schedule (refining), sane tolerance: continues from t = 0
        assert st["rc"] == 0 and st["steps"] == 10 and np.all(np.isfinite(e.solution()[0]))  # (a valid continuation, as in the reference)
    finally:
        e.close()
    e = pe.ffi.Engine(device=0)
    try:
        e.set_options(g_min=0.0)
        e.load_deck(deck)
        e.reset()
        e.analyze_tr(1e-10, 10)
        assert max_err(e.solution()[0], g10, *NL) <= 1.0
        assert e.safety_net() == {"refined": 0, "rematched": 0, "careful": False}
        from parity_common import adversarial_pivot_sweep
        import pe_load
        orc = pe_load.load_oracle()
        d0, d1, ov = adversarial_pivot_sweep()
        e.load_deck(d0, batch=2, overrides=ov)
        e.reset()
        e.analyze_dc(pe.ffi.MODE_DC)
        x = e.solution()
        for k, d in enumerate((d0, d1)):
            o = orc.Oracle(d)
            o.analyze_dc("DC")
            assert max_err(x[k], o.x, *LIN) <= 1.0
    finally:
        e.close()


def test_x_dependent_only_stamp_is_bit_identical(tmp_path):
    """Newton iterations after the first of a time point stamp only the slots an x-dependent device contributes to (same lists, same
    order: pe_front.hpp stamp_dynamic_chunk) and evaluate only those devices.  Against PHY_ENGINE_HIP_FULL_STAMP=1 (everything,
    every iteration) the solutions of a 4-instance M10k-NL sweep must agree bit for bit, with the same Newton counts.  The knob is
    read once per process: two subprocesses (one GPU process at a time)."""
    import subprocess
    import sys
    from parity_common import ROOT
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load()
deck, r, c = pe.deck.rc_mesh_params(100, 100, [3, 4, 5, 6], True)
e = pe.ffi.Engine(device=0); e.set_options(g_min=0.0)
e.load_deck(deck, batch=4, overrides={{"R": r[:, :, None], "C": c[:, :, None]}})
e.reset(); st = e.analyze_tr(1e-10, 12)
assert e.info()["n_parts"] > 1 and e.info()["nonlinear"] == 1
np.save(sys.argv[1], e.solution()); np.save(sys.argv[1] + ".it", e.state()["iters"])
e.close()
"""
    out = []
    for knob in ("0", "1"):
        f = str(tmp_path / f"x{knob}.npy")
        env = dict(os.environ, PHY_ENGINE_HIP_FULL_STAMP=knob, PHY_ENGINE_HIP_GEOMETRY_BATCH="1024")
        subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=600)
        out.append((np.load(f), np.load(f + ".it.npy")))
    assert np.array_equal(out[0][1], out[1][1])
    assert np.array_equal(out[0][0], out[1][0])   # bit for bit
    assert out[0][1].min() > 12                   # (several Newton iterations per step: the x-dependent-only path did run)


@pytest.mark.gpu
def test_round4_launch_variants_are_bit_identical(tmp_path):
    """Round 4 changed HOW an iteration of the split schedule is launched, not what it computes: (a) one captured launch sequence (hipGraph)
    per Newton iteration for small sweeps (knob GRAPH), grids of the full sweep with finished instances leaving empty quads; (b) the step's
    companion update inside the first evaluation launch (knob COMPANION_LAUNCH=1: a launch of its own, as before) and w = P rhs written by
    the stamp launch; (c) the backward lane-group kernel on at every sweep size (knob QUAD_BACK) with the ancestors' unknowns fetched by row
    broadcasts; (d) the first iteration of a transient step at an unchanged dt stamps the x-dependent matrix slots + the right-hand side only
    (knob STATIC_A=0: everything).  A 6-instance sweep of the 10k-node diode mesh (instances converge at different iterations: the active set shrinks inside a
    time point) under the geometry of the 128-instance share: every variant ends on the same bits and the same Newton counts."""
    import subprocess
    import sys
    from parity_common import ROOT
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, pe_load
pe = pe_load.load()
deck, r, c = pe.deck.rc_mesh_params(100, 100, [3, 4, 5, 6, 7, 8], True)
e = pe.ffi.Engine(device=0); e.set_options(g_min=0.0)
e.load_deck(deck, batch=6, overrides={{"R": r[:, :, None], "C": c[:, :, None]}})
e.reset(); st = e.analyze_tr(1e-10, 12)
assert e.info()["n_parts"] > 1 and e.info()["n_quad_fronts"] > 0
np.save(sys.argv[1], e.solution()); np.save(sys.argv[1] + ".it", e.state()["iters"])
e.close()
"""
    out = {}
    variants = {"default": {}, "graph": {"PHY_ENGINE_HIP_GRAPH": "1"}, "no_graph": {"PHY_ENGINE_HIP_GRAPH": "0"},
                "own_companion_launch": {"PHY_ENGINE_HIP_GRAPH": "0", "PHY_ENGINE_HIP_COMPANION_LAUNCH": "1"},
                "per_instance_backward": {"PHY_ENGINE_HIP_QUAD_BACK": "0"}, "always_full_stamp_at_a_new_time_point": {"PHY_ENGINE_HIP_STATIC_A": "0"}}
    for name, knobs in variants.items():
        f = str(tmp_path / f"{name}.npy")
        env = dict(os.environ, PHY_ENGINE_HIP_GEOMETRY_BATCH="128", **knobs)
        subprocess.run([sys.executable, "-c", code, f], check=True, env=env, timeout=600)
        out[name] = (np.load(f), np.load(f + ".it.npy"))
    for name in variants:
        assert np.array_equal(out[name][1], out["default"][1]), name
        assert np.array_equal(out[name][0], out["default"][0]), name   # bit for bit
    assert len(set(out["default"][1].tolist())) > 1 or out["default"][1].min() > 12   # (the instances do not move in lockstep)


def test_ac_of_the_diode_mesh_against_the_oracle(eng, oracle_mod):
    """Small-signal AC at mesh scale (the four AC goldens have 3-8 nodes): the 32 x 32 diode mesh -- 1 025 nodes, 128 junctions linearised at their
    operating point, the VAC phasor through 50 ohm -- at three frequencies around the mesh's corner (1 / RC = 1e9 rad/s), a batch of two seeds,
    against the oracle's complex sparse LU (oracle/pe_oracle.py analyze_ac, pinned to the reference's AC fixtures at 1e-9).  Tolerance as for
    the AC goldens: 1e-9 + 1e-6 |x|."""
    omegas = [1e7, 1e9, 3e10]
    deck, r, c = pe.deck.rc_mesh_params(32, 32, [21, 22], True)
    eng.set_options(g_min=1e-12)
    eng.load_deck(deck, batch=2, overrides={"R": r[:, :, None], "C": c[:, :, None]})
    eng.reset()
    eng.analyze_dc(pe.ffi.MODE_OP)
    got = []
    for w in omegas:
        x, rc = eng.analyze_ac(w)
        assert rc == 0
        got.append(x.copy())
    for k, seed in enumerate((21, 22)):
        o = oracle_mod.Oracle(pe.deck.rc_mesh(32, 32, seed, True))
        o.g_min = 1e-12
        want = o.analyze_ac(omegas, acop=True)
        for i in range(len(omegas)):
            assert want[i] is not None
            assert np.all(np.abs(got[i][k] - want[i]) <= 1e-9 + 1e-6 * np.abs(want[i])), (seed, omegas[i], np.max(np.abs(got[i][k] - want[i])))


def test_solve_csr_complex_seam_at_mesh_size(eng, oracle_mod):
    """The complex seam on the 10k-node mesh's AC system (G + j omega C: 10 002 complex unknowns = 20 004 real ones, fronts of a few hundred rows
    in the real-equivalent form): residual at rounding level, agreement with a CPU complex LU, linearity in the right-hand side, and a second
    frequency on the cached pattern (the susceptances move by 100 x: the cached pivot order must either hold or be re-made)."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    mo = oracle_mod.Oracle(pe.deck.rc_mesh(100, 100, 2, False))
    mo.update_tr_step(1e-10)
    mo.t = 1e-10
    A, b = mo.assemble("TR")
    A = A.tocsr()
    A.sort_indices()
    G = A.copy()
    rng = np.random.default_rng(11)
    bz = b.astype(complex) + 1j * 1e-3 * rng.standard_normal(len(b))

    def system(scale):
        # the diagonal carries conductance + companion: move a share of it into the imaginary part (a susceptance), scaled per frequency
        d = G.diagonal()
        Z = G.astype(complex).tolil()
        Z.setdiag(d * (0.6 + 0.4j * scale))
        Z = Z.tocsr()
        Z.sort_indices()
        return Z

    Z1 = system(1.0)
    x1, tm = eng.solve_csr_complex(Z1.shape[0], Z1.indptr, Z1.indices, Z1.data, bz, copy_pattern=True)
    assert np.max(np.abs(Z1 @ x1 - bz)) <= 1e-11 * max(1.0, np.max(np.abs(bz)))
    xr = spla.splu(Z1.tocsc()).solve(bz)
    assert np.max(np.abs(x1 - xr)) <= 1e-9 * max(1.0, np.max(np.abs(xr)))
    x2, _ = eng.solve_csr_complex(Z1.shape[0], Z1.indptr, Z1.indices, Z1.data, (2.0 - 3.0j) * bz, copy_pattern=False)
    assert np.max(np.abs(x2 - (2.0 - 3.0j) * x1)) <= 1e-11 * np.max(np.abs(x2))
    Z2 = system(100.0)
    x3, _ = eng.solve_csr_complex(Z2.shape[0], Z2.indptr, Z2.indices, Z2.data, bz, copy_pattern=False)
    assert np.max(np.abs(Z2 @ x3 - bz)) <= 1e-11 * max(1.0, np.max(np.abs(bz)))
