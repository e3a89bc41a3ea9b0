"""ctypes binding of libpe_hip.so (include/pe_hip.h).  Thin: plain pointers in, numpy arrays out.

The library is the product; this file only marshals.  There is no CPU fallback: if the shared library
is missing, or no HIP device is visible, `Engine()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PE_HIP_LIB", os.path.join(_HERE, "libpe_hip.so"))

# pe_hip_kind
R, CAP, L, VDC, VAC, IDC, DIODE = 1, 2, 3, 4, 5, 6, 7
IAC, VCCS, VCVS, CCCS, CCVS, OPAMP, XFMR, SWITCH, VGEN, COUPLED_L = 8, 9, 10, 11, 12, 13, 14, 15, 16, 17
NMOS, PMOS, BJT_NPN, BJT_PNP, RELAY, XFMR_CT = 18, 19, 20, 21, 22, 23
DIODE_NPARAM = 11
MODE_OP, MODE_DC, MODE_TR, MODE_TROP = 0, 1, 4, 5
OK, ERR_ARG, ERR_NO_DEVICE, ERR_SINGULAR, ERR_NO_CONVERGENCE, ERR_INTERNAL, ERR_INACCURATE = 0, -1, -2, -3, -4, -5, -6

EXPORTS = [
    "pe_hip_device_count", "pe_hip_create", "pe_hip_destroy", "pe_hip_last_error", "pe_hip_solve_csr_real", "pe_hip_solve_csr_complex", "pe_hip_build_id",
    "pe_hip_load_circuit", "pe_hip_set_options", "pe_hip_get_info", "pe_hip_set_digital_drives", "pe_hip_set_overlay", "pe_hip_set_knob", "pe_hip_get_knob", "pe_hip_update_param",
    "pe_hip_reset", "pe_hip_analyze_dc", "pe_hip_analyze_tr", "pe_hip_get_solution", "pe_hip_set_solution",
    "pe_hip_get_instance_state", "pe_hip_sweep_statistics", "pe_hip_measure_hbm_ceiling", "pe_hip_get_safety_net_counters", "pe_hip_get_newton_trace", "pe_hip_get_matrix", "pe_hip_analyze_pattern",
    "pe_hip_analyze_pattern_fronts", "pe_hip_get_phase_clocks", "pe_hip_get_phase_clocks_ex", "pe_hip_analyze_ac", "pe_hip_get_solution_ac", "pe_hip_checkpoint_size", "pe_hip_checkpoint_save", "pe_hip_checkpoint_load", "pe_hip_set_time",
    "pe_hip_sweep_create", "pe_hip_sweep_destroy", "pe_hip_sweep_last_error", "pe_hip_sweep_devices", "pe_hip_sweep_shard", "pe_hip_sweep_set_options",
    "pe_hip_sweep_load_circuit", "pe_hip_sweep_reset", "pe_hip_sweep_operating_point", "pe_hip_sweep_run", "pe_hip_sweep_reduce", "pe_hip_sweep_get_solution",
    "pe_hip_sweep_get_instance_state",
]


class DeviceTable(C.Structure):
    _fields_ = [("kind", C.c_int), ("count", C.c_int), ("nodes", C.POINTER(C.c_int)), ("branch", C.POINTER(C.c_int)),
                ("params", C.POINTER(C.c_double)), ("params_batched", C.c_int)]


class Options(C.Structure):
    _fields_ = [("v_abstol", C.c_double), ("v_reltol", C.c_double), ("i_abstol", C.c_double), ("i_reltol", C.c_double),
                ("g_min", C.c_double), ("max_newton", C.c_int), ("refactor_every_solve", C.c_int), ("r_open", C.c_double), ("residual_tol", C.c_double)]


class Timings(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("h2d_ms", "solve_ms", "d2h_ms", "solve_host_ms", "total_host_ms", "analyze_ms")]


class Info(C.Structure):
    _fields_ = [("rows", C.c_int), ("n_nodes", C.c_int), ("n_branches", C.c_int), ("batch", C.c_int), ("nnz_a", C.c_int),
                ("nnz_lu", C.c_longlong), ("nnz_lu_stored", C.c_longlong), ("n_fronts", C.c_int), ("max_front", C.c_int),
                ("tree_depth", C.c_int), ("n_row_swaps", C.c_int), ("factor_flops", C.c_double),
                ("bytes_per_instance", C.c_longlong), ("n_r", C.c_int), ("n_c", C.c_int), ("n_l", C.c_int), ("n_v", C.c_int),
                ("n_i", C.c_int), ("n_d", C.c_int), ("nonlinear", C.c_int), ("n_parts", C.c_int), ("n_top_levels", C.c_int),
                ("n_wavefronts", C.c_int), ("lds_bytes", C.c_int), ("nnz_lu_stored_top", C.c_longlong), ("n_wave_fronts", C.c_int),
                ("n_quad_fronts", C.c_int), ("nnz_lu_stored_quad", C.c_longlong), ("mid_top_limit", C.c_int), ("ew_grid", C.c_int),
                ("quad_lds_pad", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class RunStats(C.Structure):
    _fields_ = [("steps", C.c_longlong), ("newton_iters", C.c_longlong), ("gpu_ms", C.c_double), ("n_launches", C.c_int),
                ("n_failed", C.c_int), ("dominant_ms", C.c_double), ("dominant_launches", C.c_int)]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not built: run `make -C phy-engine_amd/csrc` (or __graft_entry__.build())")
        l = C.CDLL(LIB_PATH)
        l.pe_hip_last_error.restype = C.c_char_p
        l.pe_hip_last_error.argtypes = [C.c_void_p]
        l.pe_hip_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        l.pe_hip_destroy.argtypes = [C.c_void_p]
        l.pe_hip_destroy.restype = None
        l.pe_hip_load_circuit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DeviceTable)]
        l.pe_hip_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        l.pe_hip_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        l.pe_hip_set_digital_drives.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        l.pe_hip_update_param.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
        l.pe_hip_reset.argtypes = [C.c_void_p]
        l.pe_hip_analyze_dc.argtypes = [C.c_void_p, C.c_int, C.POINTER(RunStats)]
        l.pe_hip_analyze_tr.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(RunStats)]
        l.pe_hip_get_solution.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        l.pe_hip_set_solution.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        l.pe_hip_sweep_statistics.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        l.pe_hip_get_safety_net_counters.argtypes = [C.c_void_p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
        l.pe_hip_measure_hbm_ceiling.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
        l.pe_hip_get_instance_state.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong),
                                                C.POINTER(C.c_longlong), C.POINTER(C.c_double)]
        l.pe_hip_get_newton_trace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.pe_hip_get_matrix.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double)]
        l.pe_hip_solve_csr_real.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int,
                                            C.POINTER(Timings)]
        l.pe_hip_solve_csr_complex.argtypes = l.pe_hip_solve_csr_real.argtypes
        l.pe_hip_build_id.restype = C.c_char_p
        l.pe_hip_build_id.argtypes = []
        l.pe_hip_analyze_pattern.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(Info)]
        _lib = l
    return _lib


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class PeHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pe_hip error {code}: {msg}")
        self.code = code


def analyze_pattern(n, row_ptr, col_ind, values=None):
    """Host-only symbolic analysis statistics (no GPU needed)."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
    ci = np.ascontiguousarray(col_ind, dtype=np.int32)
    info = Info()
    vals = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    rc = lib().pe_hip_analyze_pattern(int(n), _ip(rp), _ip(ci), None if vals is None else _dp(vals), C.byref(info))
    if rc != 0:
        raise PeHipError(rc, "analyze_pattern failed")
    return info.asdict()


def deck_tables(deck, batch=1, overrides=None, n_drives=0):
    """Deck -> (n_nodes, n_branches, [(kind, nodes[int32 count x 2], branch[int32]|None, params[f64], batched)]).

    Branch indices follow circult::prepare (circuit.h:509-531): digital drives first, then models in order.
    `overrides`: {kind_name: array [batch][count][ncol]} per-instance parameters (Monte-Carlo sweeps).
    FBR devices expand to their four PN junctions (full_bridge_rectifier.h:19-50) with tt_in_tr = 0.
    """
    overrides = overrides or {}
    from .deck import NBRANCH, VGEN_LAYOUT
    names = ("R", "C", "L", "VDC", "VAC", "IDC", "D", "IAC", "VCCS", "VCVS", "CCCS", "CCVS", "OPAMP", "XFMR", "SW", "VGEN", "KL", "NMOS", "PMOS",
             "NPN", "PNP", "RELAY", "XCT")
    groups = {k: {"nodes": [], "branch": [], "par": []} for k in names}
    k = n_drives
    for kind, nodes, par in deck.devices:
        if kind == "FBR":
            A, B, P, M = nodes
            dflt = (1e-14, 1.0, 0.0, 2.0, 27.0, 1e-3, 40.0, 1.0, 1.0, 0.0, 0.0)
            for (a, c) in ((A, P), (B, P), (M, A), (M, B)):
                groups["D"]["nodes"].append((a, c))
                groups["D"]["par"].append(dflt)
            continue
        nb = NBRANCH[kind]
        if kind in VGEN_LAYOUT:
            ty, pos = VGEN_LAYOUT[kind]
            dflt = (5.0, 0.0, 1e3, 0.5, 0.0, 0.0, 0.0)
            par = (float(ty),) + tuple(par[q] if q >= 0 else dflt[j] for j, q in enumerate(pos))
            kind = "VGEN"
        g = groups[kind]
        g["nodes"].append(nodes)
        g["par"].append(tuple(par) + ((1.0,) if kind == "D" else ()))
        for _ in range(nb):
            g["branch"].append(k)
            k += 1
    code = {"R": R, "C": CAP, "L": L, "VDC": VDC, "VAC": VAC, "IDC": IDC, "D": DIODE, "IAC": IAC, "VCCS": VCCS, "VCVS": VCVS, "CCCS": CCCS,
            "CCVS": CCVS, "OPAMP": OPAMP, "XFMR": XFMR, "SW": SWITCH, "VGEN": VGEN, "KL": COUPLED_L, "NMOS": NMOS, "PMOS": PMOS,
            "NPN": BJT_NPN, "PNP": BJT_PNP, "RELAY": RELAY, "XCT": XFMR_CT}
    tables = []
    for name, g in groups.items():
        if not g["nodes"]:
            continue
        nodes = np.ascontiguousarray(np.array(g["nodes"], dtype=np.int32))
        branch = np.ascontiguousarray(np.array(g["branch"], dtype=np.int32)) if g["branch"] else None
        if name in overrides:
            par = np.ascontiguousarray(overrides[name], dtype=np.float64)
            assert par.shape[0] == batch and par.shape[1] == len(nodes)
            batched = 1
        else:
            par = np.ascontiguousarray(np.array(g["par"], dtype=np.float64))
            batched = 0
        tables.append((code[name], nodes, branch, par, batched))
    return deck.n_nodes, k, tables


def analyze_pattern_fronts(n, row_ptr, col_ind, values=None):
    """Host-only: (pivots, updates, parent) arrays of the assembly tree, fronts in postorder."""
    rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
    ci = np.ascontiguousarray(col_ind, dtype=np.int32)
    vals = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
    cap = max(1, int(n))
    p = np.zeros(cap, dtype=np.int32)
    u = np.zeros(cap, dtype=np.int32)
    par = np.zeros(cap, dtype=np.int32)
    nf = C.c_int()
    fn = lib().pe_hip_analyze_pattern_fronts
    fn.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                   C.POINTER(C.c_int), C.POINTER(C.c_int)]
    rc = fn(int(n), _ip(rp), _ip(ci), None if vals is None else _dp(vals), cap, _ip(p), _ip(u), _ip(par), C.byref(nf))
    if rc != 0:
        raise PeHipError(rc, "analyze_pattern_fronts failed")
    k = nf.value
    return p[:k].copy(), u[:k].copy(), par[:k].copy()


class Engine:
    """One resident circuit (optionally a batch of parameter instances) on one GPU."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().pe_hip_create(int(device), C.byref(self._h))
        if rc != 0:
            raise PeHipError(rc, lib().pe_hip_last_error(None).decode())
        self.rows = 0
        self.batch = 1
        self._keep = []

    def close(self):
        if self._h:
            lib().pe_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != 0 and rc not in allow:
            raise PeHipError(rc, lib().pe_hip_last_error(self._h).decode())
        return rc

    def set_options(self, g_min=0.0, v_abstol=0.0, v_reltol=0.0, i_abstol=0.0, i_reltol=0.0, max_newton=0, refactor_every_solve=1, r_open=0.0, residual_tol=0.0):
        o = Options(v_abstol, v_reltol, i_abstol, i_reltol, g_min, max_newton, refactor_every_solve, r_open, residual_tol)
        self._chk(lib().pe_hip_set_options(self._h, C.byref(o)))

    def set_knob(self, name, value):
        """one of the PHY_ENGINE_HIP_* tuning knobs for THIS engine; takes effect at the next analysis"""
        self._chk(lib().pe_hip_set_knob(self._h, name.encode(), int(value)))

    def get_knob(self, name):
        v, s = C.c_int(0), C.c_int(0)
        self._chk(lib().pe_hip_get_knob(self._h, name.encode(), C.byref(v), C.byref(s)))
        return v.value if s.value else None

    def set_digital_drives(self, nodes, volts):
        n = np.ascontiguousarray(nodes, dtype=np.int32)
        v = np.ascontiguousarray(volts, dtype=np.float64)
        self._chk(lib().pe_hip_set_digital_drives(self._h, len(n), _ip(n), _dp(v)))

    def load(self, n_nodes, n_branches, tables, batch=1):
        arr = (DeviceTable * max(1, len(tables)))()
        self._keep = list(tables)
        for i, (kind, nodes, branch, par, batched) in enumerate(tables):
            arr[i] = DeviceTable(kind, len(nodes), _ip(nodes), None if branch is None else _ip(branch), _dp(par), batched)
        self._chk(lib().pe_hip_load_circuit(self._h, int(n_nodes), int(n_branches), int(batch), len(tables), arr))
        self.rows = n_nodes + n_branches
        self.batch = batch

    def load_deck(self, deck, batch=1, overrides=None, n_drives=0):
        n_nodes, n_br, tables = deck_tables(deck, batch, overrides, n_drives)
        self.load(n_nodes, n_br, tables, batch)

    def info(self):
        i = Info()
        self._chk(lib().pe_hip_get_info(self._h, C.byref(i)))
        return i.asdict()

    def reset(self):
        self._chk(lib().pe_hip_reset(self._h))

    def analyze_tr(self, dt, nsteps, check=True):
        st = RunStats()
        rc = lib().pe_hip_analyze_tr(self._h, float(dt), int(nsteps), C.byref(st))
        if check:
            self._chk(rc)
        d = st.asdict()
        d["rc"] = rc
        return d

    def analyze_dc(self, mode=MODE_DC, check=True):
        st = RunStats()
        rc = lib().pe_hip_analyze_dc(self._h, int(mode), C.byref(st))
        if check:
            self._chk(rc)
        d = st.asdict()
        d["rc"] = rc
        return d

    def solution(self, first=0, count=None):
        count = self.batch - first if count is None else count
        x = np.empty((count, self.rows))
        self._chk(lib().pe_hip_get_solution(self._h, first, count, _dp(x)))
        return x

    def safety_net(self):
        """{'refined': solves repaired by refinement, 'rematched': symbolic re-analyses, 'careful': host-driven schedule forced}."""
        a, b, c = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
        self._chk(lib().pe_hip_get_safety_net_counters(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"refined": a.value, "rematched": b.value, "careful": bool(c.value)}

    def measure_hbm_ceiling(self, nbytes=1 << 31, reps=5):
        """GB/s of a device-to-device stream copy on this engine's GPU (read + written bytes / HIP-event time)."""
        g = C.c_double(0.0)
        self._chk(lib().pe_hip_measure_hbm_ceiling(self._h, C.c_size_t(nbytes), int(reps), C.byref(g)))
        return g.value

    def sweep_statistics(self):
        """[4][rows]: sum, sum of squares, min, max of the current solution over this engine's instances (computed on the device)."""
        out = np.empty((4, self.rows))
        self._chk(lib().pe_hip_sweep_statistics(self._h, _dp(out)))
        return out

    def set_solution(self, x, first=0):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1, self.rows)
        self._chk(lib().pe_hip_set_solution(self._h, first, len(x), _dp(x)))

    def state(self):
        st = np.empty(self.batch, dtype=np.int32)
        steps = np.empty(self.batch, dtype=np.int64)
        iters = np.empty(self.batch, dtype=np.int64)
        t = np.empty(self.batch)
        self._chk(lib().pe_hip_get_instance_state(self._h, 0, self.batch, _ip(st), steps.ctypes.data_as(C.POINTER(C.c_longlong)),
                                                  iters.ctypes.data_as(C.POINTER(C.c_longlong)), _dp(t)))
        return {"status": st, "steps": steps, "iters": iters, "t": t}

    def newton_trace(self, capacity=1 << 16):
        buf = np.zeros(capacity, dtype=np.int32)
        n = C.c_int()
        self._chk(lib().pe_hip_get_newton_trace(self._h, capacity, _ip(buf), C.byref(n)))
        return buf[:min(n.value, capacity)].copy()

    def matrix(self, instance=0):
        info = self.info()
        rp = np.empty(self.rows + 1, dtype=np.int32)
        ci = np.empty(info["nnz_a"], dtype=np.int32)
        va = np.empty(info["nnz_a"])
        rhs = np.empty(self.rows)
        self._chk(lib().pe_hip_get_matrix(self._h, instance, _ip(rp), _ip(ci), _dp(va), _dp(rhs)))
        return rp, ci, va, rhs

    def phase_clocks(self, instance=0):
        """In-kernel phase times of one instance since reset(), in microseconds (see pe_hip_get_phase_clocks)."""
        t = np.zeros(8, dtype=np.int64)
        fn = lib().pe_hip_get_phase_clocks
        fn.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
        self._chk(fn(self._h, instance, t.ctypes.data_as(C.POINTER(C.c_longlong))))
        names = ["eval_stamp", "lu_wave", "lu_coop", "backward_coop", "newton", "backward", "coop_asm", "coop_piv"]
        return {n: float(t[i]) / 100.0 for i, n in enumerate(names)}

    def checkpoint(self) -> bytes:
        """Device-resident simulation state of every instance as one blob (pe_hip_checkpoint_save)."""
        n = C.c_size_t()
        lib().pe_hip_checkpoint_size.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        self._chk(lib().pe_hip_checkpoint_size(self._h, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        lib().pe_hip_checkpoint_save.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self._chk(lib().pe_hip_checkpoint_save(self._h, buf, n.value))
        return buf.raw

    def restore(self, blob: bytes):
        lib().pe_hip_checkpoint_load.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self._chk(lib().pe_hip_checkpoint_load(self._h, blob, len(blob)))

    def analyze_ac(self, omega, check=True):
        """Small-signal AC at `omega` rad/s; returns the complex solution [batch][rows]."""
        st = RunStats()
        fn = lib().pe_hip_analyze_ac
        fn.argtypes = [C.c_void_p, C.c_double, C.POINTER(RunStats)]
        rc = fn(self._h, float(omega), C.byref(st))
        if check:
            self._chk(rc)
        re = np.empty((self.batch, self.rows))
        im = np.empty((self.batch, self.rows))
        if rc == 0:
            g = lib().pe_hip_get_solution_ac
            g.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            self._chk(g(self._h, 0, self.batch, _dp(re), _dp(im)))
        return re + 1j * im, rc

    def phase_clocks_coop(self, instance=0):
        """Per-layout breakdown of the cooperative fronts (see pe_hip_get_phase_clocks_ex), microseconds / counts."""
        t = np.zeros(48, dtype=np.int64)
        n = C.c_int()
        fn = lib().pe_hip_get_phase_clocks_ex
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
        self._chk(fn(self._h, instance, 48, t.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n)))
        out = {}
        for L, name in enumerate(("whole", "panel", "chain", "wave0")):
            q = t[8 + 6 * L: 14 + 6 * L] if L < 3 else t[32:38]
            out[name] = {"asm": q[0] / 100.0, "piv": q[1] / 100.0, "schur": q[2] / 100.0, "store": q[3] / 100.0, "fronts": int(q[4]), "sum_m2": int(q[5])}
        out["wave0"]["asm_own"] = t[38] / 100.0
        out["part_us"] = [float(v) / 100.0 for v in t[40:48]]
        out["wave_phase_us"] = [float(v) / 100.0 for v in t[26:32]]
        return out

    def update_param(self, kind, index, column, values):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64))
        self._chk(lib().pe_hip_update_param(self._h, kind, index, column, _dp(v), 1 if len(v) > 1 else 0))

    def solve_csr(self, n, row_ptr, col_ind, values, b, copy_pattern=True):
        rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
        ci = np.ascontiguousarray(col_ind, dtype=np.int32)
        va = np.ascontiguousarray(values, dtype=np.float64)
        bb = np.ascontiguousarray(b, dtype=np.float64)
        x = np.empty(n)
        tm = Timings()
        self._chk(lib().pe_hip_solve_csr_real(self._h, n, len(ci), _ip(rp), _ip(ci), _dp(va), _dp(bb), _dp(x), 1 if copy_pattern else 0,
                                              C.byref(tm)))
        return x, {k: getattr(tm, k) for k, _ in tm._fields_}

    def solve_csr_complex(self, n, row_ptr, col_ind, values, b, copy_pattern=True):
        """pe_hip_solve_csr_complex: complex128 arrays travel as the interleaved (re, im) doubles std::complex<double> is."""
        rp = np.ascontiguousarray(row_ptr, dtype=np.int32)
        ci = np.ascontiguousarray(col_ind, dtype=np.int32)
        va = np.ascontiguousarray(values, dtype=np.complex128)
        bb = np.ascontiguousarray(b, dtype=np.complex128)
        x = np.empty(n, dtype=np.complex128)
        tm = Timings()
        self._chk(lib().pe_hip_solve_csr_complex(self._h, n, len(ci), _ip(rp), _ip(ci), va.ctypes.data_as(C.POINTER(C.c_double)),
                                                 bb.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double)),
                                                 1 if copy_pattern else 0, C.byref(tm)))
        return x, {k: getattr(tm, k) for k, _ in tm._fields_}


def build_id():
    """16 hex digits identifying the loaded library build (pe_hip_build_id)."""
    return lib().pe_hip_build_id().decode()


class Sweep:
    """Monte-Carlo sweep over the devices of `device_mask` (pe_hip_sweep_*): contiguous instance blocks per device, one engine each."""

    def __init__(self, device_mask=1):
        l = lib()
        l.pe_hip_sweep_last_error.restype = C.c_char_p
        l.pe_hip_sweep_last_error.argtypes = [C.c_void_p]
        for name in ("pe_hip_sweep_destroy", "pe_hip_sweep_devices", "pe_hip_sweep_reset"):
            getattr(l, name).argtypes = [C.c_void_p]
        l.pe_hip_sweep_destroy.restype = None
        l.pe_hip_sweep_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        l.pe_hip_sweep_load_circuit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DeviceTable)]
        l.pe_hip_sweep_run.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(RunStats)]
        l.pe_hip_sweep_operating_point.argtypes = [C.c_void_p, C.c_int, C.POINTER(RunStats)]
        l.pe_hip_sweep_reduce.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        l.pe_hip_sweep_get_solution.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        l.pe_hip_sweep_shard.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        self._h = C.c_void_p()
        rc = l.pe_hip_sweep_create(C.c_uint(int(device_mask)), C.byref(self._h))
        if rc != 0:
            raise PeHipError(rc, (l.pe_hip_sweep_last_error(None) or b"").decode())
        self.rows = 0
        self.batch = 0

    def close(self):
        if self._h:
            lib().pe_hip_sweep_destroy(self._h)
            self._h = C.c_void_p()

    def _chk(self, rc):
        if rc != 0:
            raise PeHipError(rc, (lib().pe_hip_sweep_last_error(self._h) or b"").decode())

    def set_options(self, g_min=0.0, refactor_every_solve=1, residual_tol=0.0):
        o = Options(0.0, 0.0, 0.0, 0.0, g_min, 0, refactor_every_solve, 0.0, residual_tol)
        self._chk(lib().pe_hip_sweep_set_options(self._h, C.byref(o)))

    def load_deck(self, deck, batch, overrides=None):
        n_nodes, n_br, tables = deck_tables(deck, batch, overrides, 0)
        arr = (DeviceTable * max(1, len(tables)))()
        self._keep = list(tables)
        for i, (kind, nodes, branch, par, batched) in enumerate(tables):
            arr[i] = DeviceTable(kind, len(nodes), _ip(nodes), None if branch is None else _ip(branch), _dp(par), batched)
        self._chk(lib().pe_hip_sweep_load_circuit(self._h, int(n_nodes), int(n_br), int(batch), len(tables), arr))
        self.rows, self.batch = n_nodes + n_br, batch

    def shards(self):
        out = []
        for i in range(lib().pe_hip_sweep_devices(self._h)):
            d, f, c = C.c_int(), C.c_int(), C.c_int()
            self._chk(lib().pe_hip_sweep_shard(self._h, i, C.byref(d), C.byref(f), C.byref(c)))
            out.append((d.value, f.value, c.value))
        return out

    def reset(self):
        self._chk(lib().pe_hip_sweep_reset(self._h))

    def run(self, dt, nsteps):
        st = RunStats()
        self._chk(lib().pe_hip_sweep_run(self._h, float(dt), int(nsteps), C.byref(st)))
        return st.asdict()

    def reduce(self):
        out = np.empty((4, self.rows))
        self._chk(lib().pe_hip_sweep_reduce(self._h, _dp(out)))
        return out

    def solution(self, first=0, count=None):
        count = self.batch - first if count is None else count
        x = np.empty((count, self.rows))
        self._chk(lib().pe_hip_sweep_get_solution(self._h, int(first), int(count), _dp(x)))
        return x
