#!/usr/bin/env python3
"""Developer tool: timeline of the workgroups of the dominant kernel (k_m2_factor_parts) in one launch of the 1024-instance
sweep -- start / end of every workgroup, concurrency over time, durations by XCD and by CU (rounds, tail, placement).
Reads the per-instance timeline slots 48.. of pe_hip_get_phase_clocks_ex."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1024"))
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), False)   # linear: every instance is active in every launch
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]})
eng.reset(); eng.analyze_tr(1e-10, 2); eng.reset(); st = eng.analyze_tr(1e-10, 3)
print(f"kernel, HIP events: {st['dominant_ms'] / st['dominant_launches']:.3f} ms per launch")
fn = pe.ffi.lib().pe_hip_get_phase_clocks_ex
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
nparts = min(4, eng.info()["n_parts"]) if hasattr(eng, "info") else 4
T = np.zeros((B, nparts, 2)); H = np.zeros((B, nparts), dtype=np.int64)
for b in range(B):
    t = np.zeros(64, dtype=np.int64); n = C.c_int()
    fn(eng._h, b, 64, t.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n))
    q = t[48:48 + 3 * nparts].reshape(nparts, 3)
    T[b] = q[:, :2]; H[b] = q[:, 2]
T = (T - T[:, :, 0].min()) / 100.0   # us
D = T[:, :, 1] - T[:, :, 0]
span = T[:, :, 1].max()
print(f"span of the launch {span:.0f} us; sum of workgroup time / (span x 1024 slots) = {D.sum() / (span * 1024):.3f}")
for part in range(nparts):
    s, e = T[:, part, 0], T[:, part, 1]
    print(f"part {part}: start min/med/max {s.min():.0f}/{np.median(s):.0f}/{s.max():.0f}  end {e.min():.0f}/{np.median(e):.0f}/{e.max():.0f}  duration {D[:, part].min():.0f}/{np.median(D[:, part]):.0f}/{D[:, part].max():.0f}")
xcc = (H >> 32) & 15; hw = H & 0xffffffff
cu = (hw >> 8) & 15; se = (hw >> 13) & 7
for x in np.unique(xcc):
    m = xcc == x
    print(f"XCD {x}: {m.sum()} workgroups, mean duration {D[m].mean():.0f} us, last end {T[:, :, 1][m].max():.0f}")
key = xcc * 1000 + se * 100 + cu
means = np.array([D[key == k].mean() for k in np.unique(key)])
print(f"{len(means)} CUs: mean workgroup duration per CU min/median/max {means.min():.0f}/{np.median(means):.0f}/{means.max():.0f} us")
ev = sorted([(x, 1) for x in T[:, :, 0].ravel()] + [(x, -1) for x in T[:, :, 1].ravel()])
cur, k = 0, 0
for q in np.linspace(0, span, 17)[1:]:
    while k < len(ev) and ev[k][0] <= q:
        cur += ev[k][1]; k += 1
    print(f"  t = {q:6.0f} us: {cur} workgroups running")
eng.close()
