#!/usr/bin/env python3
"""Developer tool: one M10k-NL sweep run (env B, STEPS) for profiling under rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1024"))
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 2)
st = eng.analyze_tr(1e-10, int(os.environ.get("STEPS", "6")))
print(st)
