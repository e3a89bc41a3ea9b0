#!/usr/bin/env python3
"""Developer tool: single-instance M10k timing in multi-workgroup mode vs parts / cut."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys, time
sys.path.insert(0, %r)
import pe_load
pe = pe_load.load()
B = int(os.environ.get("B", "1"))
nl = os.environ.get("NL", "1") == "1"
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), nl)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 3)
t0 = time.perf_counter(); st = eng.analyze_tr(1e-10, 30); wall = time.perf_counter() - t0
i = eng.info()
print(f"parts={os.environ.get('PHY_ENGINE_HIP_PARTS')} cut={os.environ.get('PHY_ENGINE_HIP_PART_CUT_X10')} B={B} {'NL' if nl else 'lin'}: {st['gpu_ms']/st['newton_iters']*B:.3f} ms/iter gpu, wall {wall*1e3/st['newton_iters']*B:.3f} ms/iter, "
      f"{st['steps']/wall:.1f} steps/s wall, fronts={i['n_fronts']}", flush=True)
''' % ROOT
for cfg in os.environ.get("CFGS", "1:15,8:15,16:15,32:15,16:10,16:30").split(","):
    parts, cut = cfg.split(":")
    env = dict(os.environ, PHY_ENGINE_HIP_PARTS=parts, PHY_ENGINE_HIP_PART_CUT_X10=cut)
    subprocess.run([sys.executable, "-c", child], env=env, timeout=300)
