#!/usr/bin/env python3
"""Quick on-GPU parity + timing sweep over the golden cases (developer tool; the judged tests are tests/ -m gpu)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from parity_common import *

names = sys.argv[1:] or ["rc_step", "diode_op", "bridge_c2", "mesh32_lin", "mesh32_nl", "mesh100_lin_seed2"]
for name in names:
    meta, gx, deck = golden(name)
    eng = pe.ffi.Engine()
    t = time.time()
    snaps, trace, fail = run_engine_case(eng, meta, deck)
    el = time.time() - t
    n = min(len(snaps), len(gx))
    err = max_err(snaps[:n, 0, :], gx[:n], 1e-9, 1e-7) if n else -1
    gi = np.array(meta["newton_iters"]); k = min(len(gi), len(trace))
    print(f"{name}: err={err:.3g} snaps={len(snaps)}/{len(gx)} iters_equal={np.array_equal(gi[:k], trace[:k])} fail={fail}/{meta['fail_step']} wall={el:.2f}s", flush=True)
    eng.close()
