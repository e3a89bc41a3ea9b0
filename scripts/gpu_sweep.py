#!/usr/bin/env python3
"""Developer tool: aggregate throughput of the M10k-NL sweep vs workgroup geometry (env knobs) and batch size."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = r'''
import os, sys
sys.path.insert(0, %r)
import pe_load
pe = pe_load.load()
B = int(os.environ["B"])
deck, r, c = pe.deck.rc_mesh_params(100, 100, list(range(1, B + 1)), True)
eng = pe.ffi.Engine(); eng.set_options(g_min=0.0)
eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]}); eng.reset()
eng.analyze_tr(1e-10, 2)
st = eng.analyze_tr(1e-10, 8)
i = eng.info()
print(f"B={B} waves={os.environ.get('PHY_ENGINE_HIP_WAVES')} wave_m={os.environ.get('PHY_ENGINE_HIP_WAVE_M')} maxp={os.environ.get('PHY_ENGINE_HIP_MAX_PIVOTS')} absorb={os.environ.get('PHY_ENGINE_HIP_ABSORB_M')} leaf={os.environ.get('PHY_ENGINE_HIP_ND_LEAF')} wave_p={os.environ.get('PHY_ENGINE_HIP_WAVE_P')}: "
      f"{st['newton_iters']/st['gpu_ms']*1e3:.0f} iters/s, {st['steps']/st['gpu_ms']*1e3:.0f} steps/s, fronts={i['n_fronts']} flops={i['factor_flops']/1e6:.1f}M", flush=True)
''' % ROOT
configs = [tuple(int(v) for v in c.split(":")) for c in os.environ.get("CONFIGS", "8:48:48,8:32:32,4:32:32").split(",")]
for B in [int(x) for x in os.environ.get("BATCHES", "128,256,512").split(",")]:
    for cfg in configs:
        w, wm, mp = cfg[:3]
        ab = cfg[3] if len(cfg) > 3 else min(32, wm)
        leaf = cfg[4] if len(cfg) > 4 else 24
        wp = cfg[5] if len(cfg) > 5 else 24
        env = dict(os.environ, B=str(B), PHY_ENGINE_HIP_WAVES=str(w), PHY_ENGINE_HIP_WAVE_M=str(wm), PHY_ENGINE_HIP_MAX_PIVOTS=str(mp),
                   PHY_ENGINE_HIP_ABSORB_M=str(ab), PHY_ENGINE_HIP_ND_LEAF=str(leaf), PHY_ENGINE_HIP_WAVE_P=str(wp))
        subprocess.run([sys.executable, "-c", child], env=env, timeout=300)
