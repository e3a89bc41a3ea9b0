"""Shared helpers of the parity tests: load golden fixtures (made by scripts/make_golden.py from the real
reference), run a case through the C ABI, compare."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import pe_load  # noqa: E402

pe = pe_load.load()
GOLD = os.path.join(ROOT, "tests", "golden")
MODES = {"OP": pe.ffi.MODE_OP, "DC": pe.ffi.MODE_DC, "TROP": pe.ffi.MODE_TROP}


def golden(name):
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    width = meta["rows"] * (2 if meta["analysis"] in ("AC", "ACOP") else 1)
    x = np.fromfile(os.path.join(GOLD, name + ".bin")).reshape(-1, width) if meta["rows"] else np.zeros((0, 0))
    dp = os.path.join(GOLD, name + ".deck")
    if os.path.exists(dp):
        deck = pe.deck.Deck.read(dp)
    else:
        deck = getattr(pe.deck, meta["recipe"]["fn"])(**meta["recipe"]["kwargs"])
    return meta, x, deck


def run_engine_case(eng, meta, deck, batch=1, overrides=None):
    """Runs the golden's analysis; returns (snapshots [nsnap][batch][rows], newton trace of instance 0, fail_step)."""
    eng.set_options(g_min=meta["gmin"], r_open=meta.get("r_open", 0.0))
    eng.load_deck(deck, batch=batch, overrides=overrides)
    eng.reset()
    snaps = []
    fail = -1
    if meta["analysis"] in ("DC", "OP"):
        st = eng.analyze_dc(MODES[meta["analysis"]], check=False)
        if st["rc"] != 0:
            fail = 0
        snaps.append(eng.solution())
    else:
        done = 0
        want = list(meta["snap_steps"])
        if meta["analysis"] == "TROP":
            st = eng.analyze_dc(MODES["TROP"], check=False)
            if st["rc"] != 0:
                fail = 0
            if 0 in want:
                snaps.append(eng.solution())
                want.remove(0)
        for s in want + ([meta["steps"]] if (not want or want[-1] != meta["steps"]) else []):
            if fail >= 0:
                break
            st = eng.analyze_tr(meta["dt"], s - done, check=False)
            done = s
            if st["rc"] != 0:
                fail = int(eng.state()["steps"][0]) + 1
                break
            if s in meta["snap_steps"]:
                snaps.append(eng.solution())
    return np.array(snaps), eng.newton_trace(), fail


def max_err(a, b, atol, rtol):
    """max over entries of |a-b| / (atol + rtol*|b|): <= 1 passes."""
    return float(np.max(np.abs(a - b) / (atol + rtol * np.abs(b)))) if a.size else 0.0


def run_ac_case(eng, meta, deck):
    """AC / ACOP golden: the operating point first when the circuit is non-linear (or ACOP), then one AC solve per omega.
    Returns complex snapshots [n_omega][rows] of instance 0."""
    eng.set_options(g_min=meta["gmin"], r_open=meta.get("r_open", 0.0))
    eng.load_deck(deck)
    eng.reset()
    if meta["analysis"] == "ACOP" or deck.has_nonlinear():
        eng.analyze_dc(MODES["OP"])
    out = []
    for w in meta["omegas"]:
        x, rc = eng.analyze_ac(w)
        assert rc == 0
        out.append(x[0])
    return np.array(out)


def golden_complex(meta, gx):
    n = meta["rows"]
    return gx[:, :n] + 1j * gx[:, n:]


def adversarial_pivot_sweep():
    """Decks of the residual-safety-net test: ONE topology, two instances whose values call for different pivot orders.
    A 1 V source drives a resistive ladder (12 series arms, shunts to ground) through a switch in the middle.
      instance 0  switch closed (branch diagonal D = 0), every resistor 1 kOhm
      instance 1  switch open (D = -r_open = -1e12) and the right half of the ladder scaled by 1e6 (1 GOhm beside 1 kOhm)
    The static pivot order is matched on instance 0's values.  Returns (deck0, deck1, batched overrides)."""
    n = 12

    def build(closed, scale):
        d = pe.deck.Deck()
        d.n_nodes = n + 2
        d.add("VDC", (1, 0), 1.0)
        for i in range(1, n + 1):
            r = 1000.0 * (scale if i > n // 2 else 1.0)
            a, b = i, i + 1
            if i == n // 2:
                d.add("SW", (a, n + 2), 1.0 if closed else 0.0)   # the switch sits in series with this arm
                a = n + 2
            d.add("R", (a, b), r)
            d.add("R", (b, 0), 2.0 * r)
        return d

    d0, d1 = build(True, 1.0), build(False, 1e6)
    rr = np.array([[p[0] for k, _, p in d.devices if k == "R"] for d in (d0, d1)])[:, :, None]
    sw = np.array([[p[0] for k, _, p in d.devices if k == "SW"] for d in (d0, d1)])[:, :, None]
    return d0, d1, {"R": rr, "SW": sw}


def make(*args):
    """`make <args>` under an exclusive lock: the test modules build shared artefacts (tests/emu libraries, tests/cpp programs) from
    their fixtures, and under pytest-xdist several workers reach those fixtures at once -- two makes rebuilding the same library is how
    a worker ends up loading a half-written .so.  Serialised, the second make finds its targets up to date."""
    import fcntl
    import subprocess
    os.makedirs(os.path.join(ROOT, "tests", "cpp", "_build_emu"), exist_ok=True)
    with open(os.path.join(ROOT, "tests", "cpp", "_build_emu", ".make.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return subprocess.run(["make", *map(str, args)], check=True, capture_output=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
