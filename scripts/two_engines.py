#!/usr/bin/env python3
"""Developer tool: G engines x (1024 / G) instances of M10k-NL driven from G host threads (own stream each) vs one engine."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pe_load
pe = pe_load.load()
TOTAL, STEPS = int(os.environ.get("TOTAL", "1024")), int(os.environ.get("STEPS", "20"))
for G in [int(x) for x in os.environ.get("GROUPS", "1,2,4").split(",")]:
    per = TOTAL // G
    engs = []
    for g in range(G):
        seeds = list(range(1 + g * per, 1 + (g + 1) * per))
        deck, r, c = pe.deck.rc_mesh_params(100, 100, seeds, True)
        e = pe.ffi.Engine(); e.set_options(g_min=0.0)
        e.load_deck(deck, batch=per, overrides={"R": r[:, :, None], "C": c[:, :, None]})
        e.reset(); e.analyze_tr(1e-10, 2); e.reset()
        engs.append(e)
    out = [None] * G
    def run(g): out[g] = engs[g].analyze_tr(1e-10, STEPS)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
    [t.start() for t in th]; [t.join() for t in th]
    el = time.perf_counter() - t0
    its = sum(o["newton_iters"] for o in out)
    print(f"groups={G} x {per}: wall {el*1e3:.1f} ms for {STEPS} steps -> {TOTAL*STEPS/el:.0f} instance-steps/s, {its/el:.0f} Newton it/s; per-engine gpu_ms {[round(o['gpu_ms'],1) for o in out]}", flush=True)
    [e.close() for e in engs]
