#!/usr/bin/env python3
"""Developer tool: spread of Newton iterations per instance over the bench's 22 steps (tail of the resident kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1024"))
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 2)
i0 = eng.state()["iters"].copy()
st = eng.analyze_tr(1e-10, 20)
it = eng.state()["iters"] - i0
print(f"iterations per instance over 20 steps: mean {it.mean():.2f} min {it.min()} max {it.max()}  -> tail efficiency mean/max = {it.mean()/it.max():.3f}; gpu_ms {st['gpu_ms']:.1f}")
print("histogram", np.bincount(it)[it.min():])
