"""The multi-device sweep entry point of the C ABI (pe_hip_sweep_*) and bench.py's N > 1 path, on the CPU:
two 'devices' of the host emulation library (tests/emu: test infrastructure) and two gloo ranks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from parity_common import make

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "emu", "libpe_hip_emu.so")


@pytest.fixture(scope="module")
def emu():
    make("-C", os.path.join(ROOT, "tests", "emu"))
    return EMU


CHILD = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import pe_load
pe = pe_load.load()
B, W = 7, 12
deck, r, c = pe.deck.rc_mesh_params(W, W, list(range(1, B + 1)), True)
ov = {"R": r[:, :, None], "C": c[:, :, None]}
out = {}
for mask in (1, 3):
    s = pe.ffi.Sweep(mask)
    s.set_options(g_min=0.0)
    s.load_deck(deck, B, ov)
    s.reset()
    st = s.run(1e-10, 5)
    out[str(mask)] = {"shards": s.shards(), "stats": s.reduce().tolist(), "x": s.solution().tolist(), "steps": st["steps"], "iters": st["newton_iters"], "failed": st["n_failed"]}
    s.close()
try:
    pe.ffi.Sweep(1 << 5)
    out["bad_mask"] = "accepted"
except pe.ffi.PeHipError as e:
    out["bad_mask"] = str(e)
print(json.dumps(out))
''' % ROOT


def test_sweep_over_two_emulated_devices_matches_one(emu):
    env = dict(os.environ, PE_HIP_LIB=emu, PE_EMU_DEVICES="2")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    one, two = out["1"], out["3"]
    assert one["shards"] == [[0, 0, 7]]
    assert two["shards"] == [[0, 0, 4], [1, 4, 3]]  # contiguous blocks of ceil(7 / 2)
    assert one["steps"] == two["steps"] == 7 * 5 and one["iters"] == two["iters"] and one["failed"] == two["failed"] == 0
    a, b = np.array(one["stats"]), np.array(two["stats"])
    assert np.array_equal(a[2:], b[2:])                          # min / max: bit for bit
    assert np.allclose(a[:2], b[:2], rtol=1e-12, atol=1e-300)    # sums: the blocks are added in another order
    assert np.array_equal(np.array(one["x"]), np.array(two["x"]))  # every instance's solution is the same whichever device ran it
    assert "visible" in out["bad_mask"]


def test_bench_world2_gloo_statistics_equal_world1(emu):
    """bench.py end to end on a small mesh: 2 gloo ranks (each with its own engine and its block of instances) against 1 rank."""
    env = dict(os.environ, PE_HIP_LIB=emu, PE_BENCH_BACKEND="gloo", PE_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    args = ["--mesh", "16", "--instances", "6", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-single"]
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    l1 = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29731",
                         os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-2000:]
    l2 = json.loads([l for l in r2.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert l1["n_gpus"] == 1 and l2["n_gpus"] == 2 and l2["scaling"] == "strong"
    # the line reports the ranks that really joined, names the build of the library, and times the exchange step's all-reduce on its own
    assert l1["ranks_seen"] == 1 and l2["ranks_seen"] == 2 and l2["collective_backend"] == "gloo" and l1["allreduce_ms"] == 0.0 and l2["allreduce_ms"] > 0.0
    assert l1["build_id"] == l2["build_id"] and l1["build_id"]
    # --gpus N without N ranks is refused instead of printing a mislabelled line
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args, env=env, capture_output=True, text=True, timeout=600)
    assert r3.returncode != 0 and "rank(s) joined" in r3.stderr and not [l for l in r3.stdout.splitlines() if l.startswith("{")]
    assert l2["config"]["instances_total"] == 6 and l2["config"]["instances_rank0"] == 3
    assert l1["newton_iters_per_step"] == pytest.approx(l2["newton_iters_per_step"])
    assert l2["stats_checksum"] == pytest.approx(l1["stats_checksum"], rel=1e-12)
