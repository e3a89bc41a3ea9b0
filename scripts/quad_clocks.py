#!/usr/bin/env python3
"""Developer tool: in-kernel phase clocks of the lane-group kernel (PHY_ENGINE_HIP_QUAD=33), list 0 of the first quad."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PHY_ENGINE_HIP_QUAD"] = os.environ.get("PHY_ENGINE_HIP_QUAD", "33")
import numpy as np
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1024"))
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 2); eng.reset()
st = eng.analyze_tr(1e-10, 8)
t = np.zeros(96, dtype=np.int64); n = C.c_int()
fn = pe.ffi.lib().pe_hip_get_phase_clocks_ex
fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
for inst in (0, 512):
    fn(eng._h, inst, 96, t.ctypes.data_as(C.POINTER(C.c_longlong)), C.byref(n))
    q = t[64:70]; nf = max(1, int(q[5]))
    print(f"instance {inst}: fronts {nf} us/front: header {q[0]/100/nf:.2f} assembly {q[1]/100/nf:.2f} elimination {q[2]/100/nf:.2f} stores {q[3]/100/nf:.2f} whole {q[4]/100/nf:.2f}; launches {st['dominant_launches']}, {st['dominant_ms']/st['dominant_launches']:.3f} ms per factor launch pair")
