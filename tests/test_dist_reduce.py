"""CPU, world_size 2, gloo: the only collective of the sweep (final per-node statistics reduction)."""
import os
import subprocess
import sys

import pytest

from parity_common import ROOT

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["PE_ROOT"])
import bench
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rng = np.random.default_rng(7)
full = rng.standard_normal((7, 37))                       # 7 instances (ragged: 4 + 3), 37 rows
lo, hi = bench.shard(len(full), world, rank)              # contiguous blocks of ceil(7 / 2) instances per rank (bench.py --gpus N)
assert (lo, hi) == ((0, 4) if rank == 0 else (4, 7))
# [sum, sum2] -> one all-reduce(SUM); [-min, max] -> one all-reduce(MAX): two packed collectives in all
calls = []
real = dist.all_reduce
def counted(t, op=dist.ReduceOp.SUM, **kw):
    calls.append(op)
    return real(t, op=op, **kw)
dist.all_reduce = counted
out = bench.reduce_statistics(bench.node_statistics(full[lo:hi]), dist, torch.device("cpu"))
dist.all_reduce = real
assert calls == [dist.ReduceOp.SUM, dist.ReduceOp.MAX], calls
ref = bench.node_statistics(full)
assert np.allclose(out, ref, rtol=0, atol=1e-12), (out - ref)
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sweep_reduction_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, PE_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", "29631", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("ok") == 2


def test_bench_copies_pmc_traffic_only_from_the_same_library_build(tmp_path):
    """roofline.traffic is an OFFLINE figure (rocprofv3 --pmc passes, scripts/pmc_traffic.py).  Round 3 copied it whenever batch / mesh /
    kernel matched; now it also has to carry the build id of the library that prints the line (pe_hip_build_id), else the line says
    traffic null + traffic_stale and names the build the newest summary belongs to."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    base = {"instances_per_gpu": 1024, "nonlinear": True, "mesh": 100, "kernel": "k_m2_factor_quads + k_m2_factor_parts", "hbm_bytes_per_launch": 13.0e9, "note": "n"}
    (tmp_path / "r03_pmc_traffic.json").write_text(json.dumps(base))                                   # round 3: no build id at all
    (tmp_path / "r04_pmc_traffic.json").write_text(json.dumps(dict(base, build_id="aaaa", hbm_bytes_per_launch=12.0e9)))
    (tmp_path / "r04x_pmc_traffic.json").write_text(json.dumps(dict(base, build_id="bbbb", instances_per_gpu=128)))   # another configuration
    args = (1024, True, 100, "k_m2_factor_quads + k_m2_factor_parts<4>", 10.0e9)
    hit = bench.select_traffic(str(tmp_path), "aaaa", *args)
    assert hit["traffic"] == 12.0e9 and hit["traffic_over_algorithmic"] == pytest.approx(1.2) and "aaaa" in hit["traffic_source"] and "traffic_stale" not in hit
    stale = bench.select_traffic(str(tmp_path), "cccc", *args)
    assert stale["traffic"] is None and stale["traffic_stale"] is True and "aaaa" in stale["traffic_source"] and "cccc" in stale["traffic_source"]
    assert bench.select_traffic(str(tmp_path), "aaaa", 256, True, 100, args[3], 1.0) == {}                # nothing measured for that configuration
    assert bench.select_traffic(str(tmp_path / "missing"), "aaaa", *args) == {}
