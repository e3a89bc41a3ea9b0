#!/usr/bin/env python3
"""Developer probe: why does the sweep's exchange step (pe_hip_sweep_statistics) take milliseconds in bench.py?  Times consecutive calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1024"))
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 3)
for k in range(4):
    t0 = time.perf_counter(); s = eng.sweep_statistics(); t1 = time.perf_counter()
    print(f"call {k}: {(t1 - t0) * 1e3:.3f} ms", flush=True)
t0 = time.perf_counter(); x = eng.solution(); t1 = time.perf_counter()
print(f"solution() D2H of {x.nbytes / 1e6:.0f} MB: {(t1 - t0) * 1e3:.2f} ms")
