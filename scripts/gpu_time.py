#!/usr/bin/env python3
"""Developer tool: GPU time per Newton iteration + in-kernel phase breakdown on the M10k mesh (B = 1 and B = N)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pe_load
pe = pe_load.load()
W = int(os.environ.get("MESH", "100"))
for nonlinear in ((True,) if os.environ.get("NLONLY") else (False, True)):
    for B in [int(x) for x in os.environ.get("BATCHES", "1,128").split(",")]:
        seeds = list(range(1, B + 1))
        deck, r, c = pe.deck.rc_mesh_params(W, W, seeds, nonlinear)
        eng = pe.ffi.Engine()
        eng.set_options(g_min=0.0)
        eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]})
        eng.reset()
        eng.analyze_tr(1e-10, 2)
        eng.reset()
        st = eng.analyze_tr(1e-10, 20)
        it = st["newton_iters"] / B
        ph = eng.phase_clocks(0)
        tot = sum(ph.values())
        print(f"mesh{W} {'NL' if nonlinear else 'lin'} B={B}: {st['gpu_ms']/20:.3f} ms/step, {st['gpu_ms']/it:.3f} ms/iter(instance 0 stream), iters/step={it/20:.2f} "
              f"agg iters/s={st['newton_iters']/st['gpu_ms']*1e3:.0f} dominant kernel {st['dominant_ms']/max(1,st['dominant_launches']):.3f} ms/launch | phases us/iter: " + " ".join(f"{k}={v/it:.0f}" for k, v in ph.items()) + f" sum={tot/it:.0f}", flush=True)
        if os.environ.get("COOP") == "1":
            pc = eng.phase_clocks_coop(0)
            print("    wave phase per wavefront us/iter: " + " ".join(f"{v/it:.0f}" for v in pc.pop("wave_phase_us")), flush=True)
            print("    factorisation time per part us/iter: " + " ".join(f"{v/it:.0f}" for v in pc.pop("part_us")), flush=True)
            for name, q in pc.items():
                print(f"    coop {name}: fronts/iter={q['fronts']/it:.1f} sum_m2/iter={q['sum_m2']/it:.0f} us/iter: asm={q['asm']/it:.0f} piv={q['piv']/it:.0f} schur={q['schur']/it:.0f} store={q['store']/it:.0f}" + (f" (asm: zero+own A={q['asm_own']/it:.0f})" if "asm_own" in q else ""), flush=True)
        eng.close()
