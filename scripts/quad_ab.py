#!/usr/bin/env python3
"""Developer tool: A/B of the lane-group kernel of the wave fronts (PHY_ENGINE_HIP_QUAD=1 / 0) on the M10k-NL sweep:
ms per launch of the factor kernels (HIP events), steps/s, Newton counts, and the difference of the two solutions."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import numpy as np
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1024")); W = int(os.environ.get("MESH", "100")); STEPS = int(os.environ.get("STEPS", "20"))
deck, r, c = pe.deck.rc_mesh_params(W, W, list(range(1, B + 1)), True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 2)
eng.reset()
t0 = time.perf_counter(); st = eng.analyze_tr(1e-10, STEPS); wall = time.perf_counter() - t0
x = eng.solution(0, min(B, 8))
np.save(os.environ["OUT"], x)
i = eng.info()
print(json.dumps({"quad": os.environ.get("PHY_ENGINE_HIP_QUAD"), "B": B, "steps_per_s": st["steps"] / wall, "iters": st["newton_iters"], "gpu_ms": st["gpu_ms"],
                  "dominant_ms_per_launch": st["dominant_ms"] / max(1, st["dominant_launches"]), "launches": st["dominant_launches"], "n_failed": st["n_failed"],
                  "fronts": i["n_fronts"], "stored": i["nnz_lu_stored"], "trace": [int(v) for v in eng.newton_trace()[:STEPS]]}), flush=True)
''' % ROOT
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
outs = {}
for rep in range(int(os.environ.get("REPS", "1"))):
    for quad in os.environ.get("ARMS", "1,0").split(","):
        out = os.path.join(ROOT, "gpurun_out", f"quad_ab_{quad}.npy")
        env = dict(os.environ, OUT=out)
        if "PHY_ENGINE_HIP_QUAD" not in os.environ or len(os.environ.get("ARMS", "1,0").split(",")) > 1:
            env["PHY_ENGINE_HIP_QUAD"] = quad
        subprocess.run([sys.executable, "-c", child], env=env, timeout=600)
        outs[quad] = out
if len(outs) == 2:
    import numpy as np
    a, b = (np.load(p) for p in outs.values())
    print("max |quad - per-instance| over the first instances:", float(np.max(np.abs(a - b))), "max |x|", float(np.max(np.abs(b))))
