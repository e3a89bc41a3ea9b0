#!/usr/bin/env python3
"""Generate tests/golden/* by running the REAL reference (oracle/_ref/ref_driver, built by
`make -C oracle ref` from /root/reference/include) on decks produced by phy-engine_amd/deck.py.

Runs only in the build container (the reference does not exist on the GPU box).  Output per case:
    tests/golden/<case>.json   meta: analysis, dt, steps, gmin, snap_steps, newton_iters, fail_step, deck recipe
    tests/golden/<case>.bin    float64 snapshots [len(snap_steps)][rows]
    tests/golden/<case>.deck   the deck text (small cases only; meshes are re-generated from their seed)
"""
import concurrent.futures as cf
import importlib.util
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pe_deck", os.path.join(ROOT, "phy-engine_amd", "deck.py"))
deck = importlib.util.module_from_spec(spec)
sys.modules["pe_deck"] = deck
spec.loader.exec_module(deck)

DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
GOLD = os.path.join(ROOT, "tests", "golden")

S4 = "1,10,100,1000"
EVERY100 = ",".join(str(i) for i in range(100, 4001, 100))

# name: (recipe (fn, kwargs), analysis, dt, steps, gmin, snaps, keep_deck)
CASES = {
    "rc_step": (("rc_step", {}), "TR", 1e-8, 100, 0.0, "1,50,100", True),
    "rl_step": (("rl_step", {}), "TR", 1e-8, 100, 0.0, "1,50,100", True),
    "rlc_series_vl": (("rlc_series_vl", {}), "TR", 1e-6, 200, 0.0, "1,100,200", True),
    "rlc_series_vl_trop": (("rlc_series_vl", {}), "TROP", 1e-6, 50, 0.0, "0,1,50", True),
    "divider_dc": (("divider_dc", {}), "DC", 0.0, 0, 0.0, "", True),
    "diode_op": (("diode_op", {}), "OP", 0.0, 0, 0.0, "", True),
    "floating_rc_dc": (("floating_rc", {}), "DC", 0.0, 0, 0.0, "", True),
    "bridge_c2": (("bridge_rectifier", {}), "TR", 1e-5, 4000, 1e-12, EVERY100, True),
    "bridge_gmin0_fail": (("bridge_rectifier", {}), "TR", 1e-4, 100, 0.0, "50", True),
    "ladder_c1": (("resistor_ladder", {"n": 1000, "merges": 100, "seed": 1}), "DC", 0.0, 0, 0.0, "", False),
    "mesh32_lin": (("rc_mesh", {"W": 32, "H": 32, "seed": 1, "nonlinear": False}), "TR", 1e-10, 1000, 0.0, S4, False),
    "mesh32_nl": (("rc_mesh", {"W": 32, "H": 32, "seed": 1, "nonlinear": True}), "TR", 1e-10, 1000, 0.0, S4, False),
    "mesh100_lin": (("rc_mesh", {"W": 100, "H": 100, "seed": 1, "nonlinear": False}), "TR", 1e-10, 1000, 0.0, S4, False),
    "mesh100_nl": (("rc_mesh", {"W": 100, "H": 100, "seed": 1, "nonlinear": True}), "TR", 1e-10, 1000, 0.0, S4, False),
}
for sd in range(2, 9):
    CASES[f"mesh32_lin_seed{sd}"] = (("rc_mesh", {"W": 32, "H": 32, "seed": sd, "nonlinear": False}), "TR", 1e-10, 100, 0.0, "100", False)
    CASES[f"mesh32_nl_seed{sd}"] = (("rc_mesh", {"W": 32, "H": 32, "seed": sd, "nonlinear": True}), "TR", 1e-10, 100, 0.0, "100", False)
CASES["mesh100_lin_seed2"] = (("rc_mesh", {"W": 100, "H": 100, "seed": 2, "nonlinear": False}), "TR", 1e-10, 100, 0.0, "100", False)
# config C5: exact per-instance parity of the M10k-NL sweep for seeds 1..8 (seed 1 = mesh100_nl)
for sd in range(2, 9):
    CASES[f"mesh100_nl_seed{sd}"] = (("rc_mesh", {"W": 100, "H": 100, "seed": sd, "nonlinear": True}), "TR", 1e-10, 100, 0.0, "10,100", False)


# the reference's own model tests (DC) + transient variants of the same devices
for nm in ("vccs_dc", "vcvs_gain", "cccs_dc", "ccvs_dc", "op_amp_follower", "transformer_ratio", "generator_dc"):
    CASES[nm] = ((nm, {}), "DC", 0.0, 0, 0.0, "", True)
CASES["switch_open_dc"] = (("switch_divider", {"closed": False}), "DC", 0.0, 0, 0.0, "", True)
CASES["switch_closed_dc"] = (("switch_divider", {"closed": True}), "DC", 0.0, 0, 0.0, "", True)
CASES["switch_open_ropen1e6_dc"] = (("switch_divider", {"closed": False}), "DC", 0.0, 0, 0.0, "", True)
R_OPEN = {"switch_open_ropen1e6_dc": 1e6}
CASES["generators_tr"] = (("generators_tr", {}), "TR", 5e-6, 400, 0.0, "1,7,50,133,200,399,400", True)
CASES["generators_trop"] = (("generators_tr", {}), "TROP", 5e-6, 20, 0.0, "0,1,20", True)
CASES["iac_rc_tr"] = (("iac_rc", {}), "TR", 1e-5, 300, 0.0, "1,100,300", True)
CASES["iac_rc_dc"] = (("iac_rc", {}), "DC", 0.0, 0, 0.0, "", True)
CASES["iac_rc_trop"] = (("iac_rc", {}), "TROP", 1e-5, 10, 0.0, "0,1,10", True)
CASES["coupled_l_k0_tr"] = (("coupled_inductors_tr", {"k": 0.0}), "TR", 1e-5, 10, 0.0, "1,10", True)
CASES["coupled_l_k09_tr"] = (("coupled_inductors_tr", {"k": 0.9}), "TR", 1e-5, 100, 0.0, "1,10,100", True)
CASES["coupled_l_k09_trop"] = (("coupled_inductors_tr", {"k": 0.9}), "TROP", 1e-5, 20, 0.0, "0,1,20", True)
CASES["coupled_l_dc"] = (("coupled_inductors_tr", {"k": 0.5}), "DC", 0.0, 0, 0.0, "", True)
CASES["controlled_mix_tr"] = (("controlled_mix", {}), "TR", 1e-6, 300, 0.0, "1,10,100,300", True)


# three-pin non-linear devices (no dedicated test in the reference besides the loader coverage: circuits are ours)
CASES["nmos_cutoff_dc"] = (("nmos_common_source", {"vg": 0.5}), "DC", 0.0, 0, 0.0, "", True)
CASES["nmos_sat_dc"] = (("nmos_common_source", {"vg": 2.0}), "DC", 0.0, 0, 0.0, "", True)
CASES["nmos_triode_op"] = (("nmos_common_source", {"vg": 4.5}), "OP", 0.0, 0, 0.0, "", True)
CASES["cmos_inverter_tr"] = (("cmos_inverter_tr", {}), "TR", 1e-7, 250, 1e-12, "1,3,10,50,55,100,105,250", True)
CASES["bjt_npn_ce_dc_fail"] = (("bjt_common_emitter", {"pnp": False}), "DC", 0.0, 0, 0.0, "", True)
CASES["bjt_pnp_ce_op_fail"] = (("bjt_common_emitter", {"pnp": True}), "OP", 0.0, 0, 0.0, "", True)
CASES["bjt_amp_tr"] = (("bjt_amp_tr", {}), "TR", 1e-6, 300, 0.0, "1,10,100,300", True)
CASES["bjt_amp_trop_fail"] = (("bjt_amp_tr", {}), "TROP", 1e-6, 50, 0.0, "0,1,50", True)   # cold-start OP of an unlimited exp: the reference gives up


CASES["center_tap_ratio"] = (("center_tap_ratio", {}), "DC", 0.0, 0, 0.0, "", True)
CASES["relay_ramp_tr"] = (("relay_ramp", {}), "TR", 1e-4, 200, 0.0, "1,30,62,63,64,70,100,130,137,138,139,150,200", True)


# small-signal AC: snapshots are [Re x ; Im x] per frequency point
AC_OMEGAS = {"ac_rc_lowpass": [10.0, 1000.0, 1e5], "ac_rlc_diode_acop": [0.0, 1e3, 2e4, 1e6], "ac_linear_mix": [100.0, 6283.185307179586, 1e5],
             "ac_nmos_amp": [1e2, 6.283185307179586e4, 1e7]}
CASES["ac_rc_lowpass"] = (("ac_rc_lowpass", {}), "AC", 0.0, 0, 0.0, "", True)
CASES["ac_rlc_diode_acop"] = (("ac_rlc_diode", {}), "ACOP", 0.0, 0, 0.0, "", True)
CASES["ac_linear_mix"] = (("ac_linear_mix", {}), "AC", 0.0, 0, 0.0, "", True)
CASES["ac_nmos_amp"] = (("ac_nmos_amp", {}), "AC", 0.0, 0, 0.0, "", True)


def tt_diode_deck():
    """test/0004.solver/pn_junction_tt_tr.cpp: VDC 0.7 + VAC 0.1 (omega*dt = pi/2) across a tt=1e-9 diode."""
    import math
    d = deck.Deck()
    d.n_nodes = 2
    d.add("VDC", (1, 0), 0.7)
    d.add("VAC", (2, 1), 0.1, math.pi / (2.0 * 1e-8), 0.0)
    d.add("D", (2, 0), 1e-14, 1.0, 0.0, 2.0, 27.0, 1e-3, 40.0, 1.0, 1.0, 1e-9)
    return d


deck.tt_diode = tt_diode_deck
CASES["pn_tt_tr"] = (("tt_diode", {}), "TR", 1e-8, 2, 0.0, "1,2", True)


def run_case(name):
    (fn, kw), analysis, dt, steps, gmin, snaps, keep = CASES[name]
    d = getattr(deck, fn)(**kw)
    with tempfile.TemporaryDirectory() as tmp:
        dp = os.path.join(tmp, name + ".deck")
        d.write(dp)
        out = os.path.join(tmp, name)
        cmd = [DRIVER, dp, "--analysis", analysis, "--gmin", repr(gmin), "--out", out]
        if analysis in ("TR", "TROP"):
            cmd += ["--dt", repr(dt), "--steps", str(steps), "--snap", snaps]
        if name in R_OPEN:
            cmd += ["--ropen", repr(R_OPEN[name])]
        if name in AC_OMEGAS:
            cmd += ["--omegas", ",".join(repr(w) for w in AC_OMEGAS[name])]
        if (d.rows <= 2000 or name in ("mesh100_lin", "mesh100_nl")) and analysis not in ("AC", "ACOP"):  # (the restated Newton counter == circult::analyze(), also on the largest cases)
            cmd += ["--check-analyze"]
        subprocess.run(cmd, check=True)
        meta = json.load(open(out + ".json"))
        meta["recipe"] = {"fn": fn, "kwargs": kw}
        if name in R_OPEN:
            meta["r_open"] = R_OPEN[name]
        if name in AC_OMEGAS:
            meta["omegas"] = AC_OMEGAS[name]
        meta["generator"] = "scripts/make_golden.py via oracle/_ref/ref_driver (real reference)"
        json.dump(meta, open(os.path.join(GOLD, name + ".json"), "w"))
        os.replace(out + ".bin", os.path.join(GOLD, name + ".bin"))
        if keep:
            d.write(os.path.join(GOLD, name + ".deck"))
    return name, meta["rows"], meta["fail_step"], meta.get("analyze_bit_equal")


def sweep_statistics(name="mesh100_nl_stats32", W=100, seeds=range(1, 33), steps=10):
    """Config C5: per-row {sum, sum of squares, min, max} of the solution after `steps` steps over the instances seed = 1..32
    of the M10k-NL sweep, each run by the REAL reference -- the 32-instance CPU subset SURVEY.md 8(d) compares the sweep's
    statistics with.  tests/golden/<name>.bin = float64 [4][rows]."""
    import numpy as np

    def one(sd):
        d = deck.rc_mesh(W, W, sd, True)
        with tempfile.TemporaryDirectory() as tmp:
            dp = os.path.join(tmp, "m.deck")
            d.write(dp)
            out = os.path.join(tmp, "m")
            subprocess.run([DRIVER, dp, "--analysis", "TR", "--gmin", "0.0", "--out", out, "--dt", repr(1e-10), "--steps", str(steps), "--snap", str(steps)], check=True)
            meta = json.load(open(out + ".json"))
            return np.fromfile(out + ".bin").reshape(-1, meta["rows"])[-1], meta["newton_iters"]

    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(one, seeds))
    x = np.stack([r[0] for r in res])
    stats = np.stack([x.sum(axis=0), (x * x).sum(axis=0), x.min(axis=0), x.max(axis=0)])
    stats.tofile(os.path.join(GOLD, name + ".bin"))
    json.dump({"rows": int(x.shape[1]), "seeds": list(seeds), "steps": steps, "dt": 1e-10, "mesh": W, "layout": "[sum, sum of squares, min, max][rows]",
               "newton_iters_total": [int(sum(r[1])) for r in res],
               "generator": "scripts/make_golden.py stats via oracle/_ref/ref_driver (real reference), one run per seed"}, open(os.path.join(GOLD, name + ".json"), "w"))
    return name, x.shape


# goldens that are the plain output of a reference-linked program (oracle/Makefile): name -> (binary, file)
PROGRAMS = {"adc": ("ref_adc", "adc_c4.json"), "digital": ("ref_digital", "digital_blocks.json")}


def penl_fixtures():
    """tests/golden/penl/: PE-NL containers WRITTEN BY THE REAL REFERENCE (oracle/_ref/ref_penl = tests/cpp/penl_tool.cpp compiled against
    the reference's pe_nl_fileformat.h and its vendored LevelDB) + the canonical dump the reference prints for each.  ref_reopened/ is a
    database directory the reference's LevelDB has opened a second time: its recovery turned the write-ahead log into a sorted table."""
    import shutil
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_penl")
    out = os.path.join(GOLD, "penl")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    def run(*a):
        return subprocess.run([exe, *a], capture_output=True, text=True, check=True, cwd=out).stdout
    open(os.path.join(out, "ref_full.dump"), "w").write(run("save", "ref_full.penl", "full", "file", "solve"))
    open(os.path.join(out, "ref_struct.dump"), "w").write(run("dump", "ref_struct.penl") if run("save", "ref_struct.penl", "structure", "file") else "")
    run("save", "ref_ck.penl", "runtime", "file", "solve")
    open(os.path.join(out, "ref_zoo.dump"), "w").write(run("dump", "ref_zoo.penl") if run("save", "ref_zoo.penl", "structure", "file", "zoo") else "")
    run("save", "ref_dir", "full", "dir", "solve")
    run("save", "ref_reopened", "structure", "dir", "zoo")
    run("dump", "ref_reopened")  # second open: LevelDB recovery writes 00000N.ldb and a new MANIFEST
    for d in ("ref_dir", "ref_reopened"):
        for junk in ("LOG", "LOG.old", "LOCK"):
            p = os.path.join(out, d, junk)
            if os.path.exists(p):
                os.remove(p)
    open(os.path.join(out, "schema.txt"), "w").write(subprocess.run([exe, "schema"], capture_output=True, text=True, check=True).stdout)
    return sorted(os.listdir(out))


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    os.makedirs(GOLD, exist_ok=True)
    if "penl" in names:
        print(penl_fixtures(), flush=True)
        names = [n for n in names if n != "penl"]
        if not names:
            sys.exit(0)
    for n in [n for n in names if n in PROGRAMS]:
        exe, fn = PROGRAMS[n]
        out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", exe)], capture_output=True, text=True, check=True).stdout
        json.loads(out)
        open(os.path.join(GOLD, fn), "w").write(out)
        print((n, fn, len(out)), flush=True)
    if "stats" in names:
        print(sweep_statistics(), flush=True)
    names = [n for n in names if n not in PROGRAMS and n != "stats"]
    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        for r in ex.map(run_case, names):
            print(r, flush=True)
