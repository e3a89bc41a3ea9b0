#!/usr/bin/env python3
"""bench.py -- transient hot path on the synthetic 10k-node RC mesh (BASELINE.json metric).

Workload (config C5 of SURVEY.md 8d = BASELINE.json configs[4], the largest single-GPU configuration; at N = 1 the whole sweep
sits on the one GPU): M10k-NL = 100x100 RC mesh, R 1k +-5%, C 1p +-5%, 1249 clamp diodes, VAC 2 V / 100 MHz through 50 ohm,
dt = 1e-10 s, **1024 Monte-Carlo instances in total** (seed = global instance index + 1) sharing one symbolic analysis.
`--gpus N` shards them in contiguous blocks of ceil(1024 / N) per rank (the chunk rule of the reference's only multi-GPU code,
src/pe_synth_cuda_u64_cones.cu:1894-1904): STRONG scaling, no data-path collective; the only exchange is the final reduction
of per-node statistics (two packed RCCL all-reduces of 160 KB each), outside the timed region and reported as reduce_ms.
`--batch B` overrides the per-GPU instance count (then every rank holds B instances: weak scaling, reported as such).
A "step" is one transient time step of every instance of a rank (companion update -> Newton{device eval, MNA gather,
multifrontal LU, triangular solves, convergence test}).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One JSON line on rank 0.  value = instance-steps per second over all GPUs.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import pe_load  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
HBM_GUIDE_COPY_GBS = 6290.0  # ... and the float4 device-to-device copy the guide measured on this part (MI355X_MICROARCH.md:36,296)


def bytes_per_iteration(info):
    """Algorithmic bytes of one Newton iteration, SURVEY.md 8(d), with this engine's own F = nnz(L)+nnz(U)."""
    F = info["nnz_lu"]
    N = info["rows"]
    nnzA = info["nnz_a"]
    b_stamp = 16 * info["n_r"] + 24 * info["n_c"] + 40 * info["n_l"] + 72 * info["n_d"] + 24 * info["n_v"] + 8 * nnzA + 8 * N
    b_factor = 12 * nnzA + 20 * F
    b_solve = 12 * F + 16 * N
    b_newton = 24 * N
    return {"stamp": b_stamp, "factor": b_factor, "solve": b_solve, "newton": b_newton, "iter": b_stamp + b_factor + b_solve + b_newton,
            "companion_per_step": 40 * info["n_c"] + 48 * info["n_l"] + 24 * info["n_d"]}


def node_statistics(x):
    """Per-row {sum v, sum v^2, min, max} over instances (numpy restatement of pe_hip_sweep_statistics; tests only)."""
    return np.stack([x.sum(axis=0), (x * x).sum(axis=0), x.min(axis=0), x.max(axis=0)])


def reduce_statistics(local, dist=None, device=None):
    """The sweep's one exchange step: local = [4][rows] {sum, sum of squares, min, max}.  Packed as
    [sum, sum2] -> one all-reduce(SUM) and [-min, max] -> one all-reduce(MAX)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    import torch
    t = torch.from_numpy(np.ascontiguousarray(local)).to(device or "cpu")
    sums = t[:2].contiguous()
    ext = torch.stack([-t[2], t[3]]).contiguous()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(ext, op=dist.ReduceOp.MAX)
    return torch.cat([sums, -ext[:1], ext[1:]]).cpu().numpy()


def select_traffic(profiles_dir, build_id, B, nonlinear, W, kernel, algorithmic_bytes_per_launch):
    """roofline.traffic from the committed PMC summaries (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.py): the newest one
    measured for this configuration (instances per GPU, mesh, variant, kernel) counts -- and only when it was measured on THIS build of
    the library (pe_hip_build_id: hash of its sources and flags).  A figure from another build is reported as stale, never carried over."""
    out = {}
    try:
        names = sorted((f for f in os.listdir(profiles_dir) if f.endswith("_pmc_traffic.json")), reverse=True)
    except OSError:
        return out
    for tp in names:
        try:
            tj = json.load(open(os.path.join(profiles_dir, tp)))
        except Exception:
            continue
        if not (tj.get("instances_per_gpu") == B and tj.get("nonlinear") == nonlinear and tj.get("mesh") == W and
                tj.get("kernel", "").split("<")[0] == kernel.split("<")[0]):
            continue
        if build_id and tj.get("build_id") == build_id:
            return {"traffic": tj["hbm_bytes_per_launch"], "traffic_over_algorithmic": tj["hbm_bytes_per_launch"] / algorithmic_bytes_per_launch,
                    "traffic_source": f"profiles/{tp} (build_id {tj['build_id']}): " + tj.get("note", "")}
        if "traffic_stale" not in out:
            out = {"traffic": None, "traffic_stale": True,
                   "traffic_source": f"none for build_id {build_id}: the newest summary for this configuration, profiles/{tp}, is of build "
                                     f"{tj.get('build_id', 'unknown (before round 4)')} ({tj['hbm_bytes_per_launch']:.4g} B per launch there)"}
    return out


def shard(total, world, rank):
    """Contiguous blocks of ceil(total / world) instances per rank (the last ranks may get fewer, or none)."""
    chunk = -(-total // world)
    lo = min(total, rank * chunk)
    return lo, min(total, lo + chunk)


def measured_hbm_ceiling(eng):
    """On-box achievable HBM bandwidth (report the fraction of the spec AND of the measured ceiling): a
    device-to-device stream copy of 2 GiB by the engine's own copy kernel (pe_hip_measure_hbm_ceiling)."""
    try:
        return eng.measure_hbm_ceiling(1 << 31, 5)
    except Exception as e:  # (reported, never fatal: the ceiling is an extra)
        return f"unavailable: {type(e).__name__}: {e}"


def cpu_baseline(deck, dt, nonlinear, budget_steps):
    """Reference CPU path on this host's cores: the real reference binary when it travelled with the repo
    (oracle/_ref/ref_driver, kind 'reference'), else the numpy/scipy restatement (kind 'port').  1 core."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if os.path.exists(drv):
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            dp = os.path.join(tmp, "bench.deck")
            deck.write(dp)
            out = subprocess.run([drv, dp, "--bench", "--dt", repr(dt), "--steps", str(budget_steps), "--warmup", "1"], capture_output=True, text=True,
                                 timeout=600)
            if out.returncode == 0:
                r = json.loads(out.stdout.strip().splitlines()[-1])
                return {"value": r["steps_per_s"], "unit": "steps/s", "cores": 1, "kind": "reference",
                        "newton_iters_per_s": r["newton_iters_per_s"],
                        "sample": f"{budget_steps} TR steps of instance seed=1 (same deck), real reference binary (Eigen SparseLU, complex), 1 thread"}
    orc = pe_load.load_oracle()
    o = orc.Oracle(deck)
    o.analyze_tr(dt, 1)
    t0 = time.time()
    n = max(2, budget_steps // 2)
    o.analyze_tr(dt, n)
    el = time.time() - t0
    it = sum(o.newton_iters[1:])
    return {"value": n / el, "unit": "steps/s", "cores": 1, "kind": "port", "newton_iters_per_s": it / el,
            "sample": f"{n} TR steps of instance seed=1, oracle/pe_oracle.py (scipy SuperLU, refactor every solve), 1 thread"}


def cpu_baseline_all_cores(W, dt, nonlinear, budget_steps, pe):
    """the sweep on the CPU is one circuit per thread on all host cores -- one reference process per core,
    instance seeds 1..cores, started together; aggregate = sum of the per-process steps / the slowest process's wall time.
    Reported next to the 1-core baseline (an extra object, never the measured path)."""
    import concurrent.futures as cf
    import tempfile
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(drv):
        return None
    cores = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for k in range(cores):
            dp = os.path.join(tmp, f"i{k}.deck")
            pe.deck.rc_mesh(W, W, k + 1, nonlinear).write(dp)
            paths.append(dp)

        def run(dp):
            t0 = time.time()
            out = subprocess.run([drv, dp, "--bench", "--dt", repr(dt), "--steps", str(budget_steps), "--warmup", "1"], capture_output=True, text=True,
                                 timeout=900)
            el = time.time() - t0
            return (json.loads(out.stdout.strip().splitlines()[-1]) if out.returncode == 0 else None), el

        t0 = time.time()
        with cf.ThreadPoolExecutor(max_workers=cores) as ex:
            res = list(ex.map(run, paths))
        wall = time.time() - t0
    ok = [r for r, _ in res if r]
    if len(ok) != cores:
        return None
    # each process times its own TR loop (netlist build / warm-up excluded): aggregate over the slowest one's loop time
    loop_s = max(budget_steps / r["steps_per_s"] for r in ok)
    return {"value": cores * budget_steps / loop_s, "unit": "instance-steps/s", "cores": cores, "kind": "reference",
            "newton_iters_per_s": sum(r["newton_iters_per_s"] / r["steps_per_s"] for r in ok) * budget_steps / loop_s,
            "sample": f"{cores} reference processes in parallel (one per host core), {budget_steps} TR steps each, seeds 1..{cores}; wall incl. start-up {wall:.1f} s"}


def single_circuit_numbers(pe, W, dt, device):
    """One M10k instance (no batch): NL and linear, full refactorisation per solve like the reference; and the linear
    circuit with the factors reused while dt is unchanged (legitimate for a linear circuit, SURVEY.md 8d -- flagged)."""
    out = {}
    for key, nonlinear, refac, steps in (("nl_steps_per_s", True, 1, 30), ("linear_steps_per_s", False, 1, 60), ("linear_reuse_factor_steps_per_s", False, 0, 200)):
        eng = pe.ffi.Engine(device=device)
        eng.set_options(g_min=0.0, refactor_every_solve=refac)
        eng.load_deck(pe.deck.rc_mesh(W, W, 1, nonlinear))
        eng.reset()
        eng.analyze_tr(dt, 10)   # (throw-away run: GPU clocks, see shard_rates), then the window of rounds 1-3: 3 warm-up steps + `steps`
        eng.reset()
        eng.analyze_tr(dt, 3)
        st = eng.analyze_tr(dt, steps)
        out[key] = st["steps"] / (st["gpu_ms"] * 1e-3)
        if nonlinear:
            out["nl_newton_iters_per_s"] = st["newton_iters"] / (st["gpu_ms"] * 1e-3)
        eng.close()
    out["note"] = ("one circuit spread over parts x workgroups + top levels (multi-workgroup schedule, one launch per phase, host Newton "
                   "loop): latency-bound; linear_reuse_factor skips B_factor (stamp + triangular solves only)")
    return out


def shard_rates(pe, W, dt, nonlinear, device, total=1024):
    """What ONE GPU delivers on the per-GPU share of the sweep at 8 / 4 / 2 GPUs (contiguous blocks of ceil(total / N) instances):
    instance-steps/s at 128, 256 and 512 instances, each with its own launch geometry.  Outside the timed region; lets a reader
    project the aggregate (N x rate(total / N)) against the 1-GPU rate without an 8-GPU node -- a projection, not a measurement."""
    out = {}
    for n_gpus in (8, 4, 2):
        B = -(-total // n_gpus)
        try:
            deck, r, c = pe.deck.rc_mesh_params(W, W, list(range(1, B + 1)), nonlinear)
            eng = pe.ffi.Engine(device=device)
            eng.set_options(g_min=0.0)
            eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]})
            eng.reset()
            # (a throw-away run first: these measurements are 70-400 ms long and start right after a second of host-only work -- the symbolic
            #  analysis -- during which the GPU clocks fall back; without it the 128-instance figure moved by 7 % between runs of one library.
            #  Then the SAME window of the transient as the headline measurement: 2 warm-up steps + steps 3..22 -- the Newton iterations per
            #  step change along the transient, another window would not be comparable with `value`)
            eng.analyze_tr(dt, 10)
            eng.reset()
            eng.analyze_tr(dt, 2)
            t0 = time.perf_counter()
            st = eng.analyze_tr(dt, 20)
            el = time.perf_counter() - t0
            out[str(B)] = {"instance_steps_per_s": st["steps"] / el, "gpu_ms_per_step": st["gpu_ms"] / 20, "n_parts": eng.info()["n_parts"],
                           "newton_iters_per_step": st["newton_iters"] / max(1, st["steps"])}
            eng.close()
        except Exception as e:
            out[str(B)] = {"error": str(e)}
    return out


def other_config_numbers(pe, device):
    """steps/s of config C2 (diode-bridge transient, a 4-row circuit: pure launch/latency figure) and samples/s of
    config C4 (flash-ADC mixed signal through the C++ plug-in API: tests/cpp/adc_flash, when it has been built)."""
    out = {}
    try:
        eng = pe.ffi.Engine(device=device)
        eng.set_options(g_min=1e-12)
        eng.load_deck(pe.deck.bridge_rectifier())
        eng.reset()
        eng.analyze_tr(1e-5, 100)
        st = eng.analyze_tr(1e-5, 3900)
        out["c2_bridge_steps_per_s"] = st["steps"] / (st["gpu_ms"] * 1e-3)
        out["c2_bridge_newton_iters_per_s"] = st["newton_iters"] / (st["gpu_ms"] * 1e-3)
        eng.close()
    except Exception as e:
        out["c2_error"] = str(e)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "adc_flash")
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
            j = json.loads(r.stdout)
            if "samples_per_s" in j:
                out["c4_adc_samples_per_s"] = j["samples_per_s"]
                out["c4_note"] = j.get("timing_note", "")
        except Exception as e:
            out["c4_error"] = str(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--instances", type=int, default=1024, help="Monte-Carlo instances of the whole sweep (config C5), sharded over the GPUs")
    ap.add_argument("--batch", type=int, default=0, help="override: instances PER GPU (weak scaling); 0 = shard --instances (strong scaling)")
    ap.add_argument("--mesh", type=int, default=100)
    ap.add_argument("--linear", action="store_true", help="VDC-driven linear variant (no diodes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the extra single-circuit measurement")
    ap.add_argument("--cpu-steps", type=int, default=60)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    tdev = None
    # rehearsal knobs (not used by the driver): several ranks on ONE GPU over gloo, to exercise the N>1 code path on a 1-GPU box
    backend = os.environ.get("PE_BENCH_BACKEND", "nccl")
    device_index = int(os.environ.get("PE_BENCH_DEVICE", local_rank))
    if world > 1 or os.environ.get("PE_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(device_index)
            tdev = torch.device("cuda", device_index)
            dist.init_process_group("nccl", device_id=tdev)
        else:
            tdev = torch.device("cpu")
            dist.init_process_group(backend)

    # the line reports n_gpus = the ranks that really joined: a launch that asked for --gpus N and got another world size is refused
    # (a silent N = 1 run labelled n_gpus: 8 would be the worst kind of scaling number)
    ranks_seen = dist.get_world_size() if dist is not None else 1
    if ranks_seen != world or (args.gpus != ranks_seen and not os.environ.get("PE_BENCH_ALLOW_GPUS_MISMATCH")):
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but {ranks_seen} rank(s) joined (WORLD_SIZE={world}): launch with "
                         f"python -m torch.distributed.run --nproc-per-node {args.gpus} ... bench.py --gpus {args.gpus}")

    pe = pe_load.load()
    W = args.mesh
    nonlinear = not args.linear
    dt = 1e-10
    if args.batch > 0:
        lo, hi = rank * args.batch, (rank + 1) * args.batch
        total, scaling = world * args.batch, "weak"
    else:
        lo, hi = shard(args.instances, world, rank)
        total, scaling = args.instances, "strong"
    if hi <= lo:
        raise SystemExit(f"rank {rank}: no instances to run ({total} instances over {world} ranks)")
    B = hi - lo
    seeds = [lo + k + 1 for k in range(B)]
    deck, r, c = pe.deck.rc_mesh_params(W, W, seeds, nonlinear)
    eng = pe.ffi.Engine(device=device_index)
    eng.set_options(g_min=0.0)
    eng.load_deck(deck, batch=B, overrides={"R": r[:, :, None], "C": c[:, :, None]})
    eng.reset()

    def barrier():
        if dist is not None:
            import torch
            if tdev.type == "cuda":
                torch.cuda.synchronize()
            dist.barrier()
            if tdev.type == "cuda":
                torch.cuda.synchronize()

    st_w = eng.analyze_tr(dt, args.warmup) if args.warmup > 0 else None
    info = eng.info()
    barrier()
    t0 = time.perf_counter()
    st = eng.analyze_tr(dt, args.steps)  # synchronises the engine's stream before returning
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([el, st["gpu_ms"]], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = torch.tensor([st["steps"], st["newton_iters"]], dtype=torch.float64, device=tdev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        el, gpu_ms = float(tmax[0]), float(tmax[1])
        steps_total, iters_total = float(tot[0]), float(tot[1])
    else:
        gpu_ms = st["gpu_ms"]
        steps_total, iters_total = float(st["steps"]), float(st["newton_iters"])

    # the one exchange step of the sweep: per-node statistics at t_end (device kernel + 320 KB D2H), then the two packed all-reduces --
    # timed separately: allreduce_ms is what RCCL over xGMI adds at N > 1 (0 on one rank)
    t1 = time.perf_counter()
    local_stats = eng.sweep_statistics()
    t2 = time.perf_counter()
    stats = reduce_statistics(local_stats, dist, tdev)
    if dist is not None and tdev.type == "cuda":
        import torch
        torch.cuda.synchronize()
    t3 = time.perf_counter()
    reduce_ms = (t3 - t1) * 1e3
    allreduce_ms = (t3 - t2) * 1e3 if ranks_seen > 1 else 0.0

    if rank == 0:
        bpi = bytes_per_iteration(info)
        # Whole hot path: the launches of the timed region processed (local) newton_iters iterations + steps companion updates;
        # duration from HIP events on the engine's stream.
        split = info.get("n_parts", 1) > 1 or st["dominant_launches"] != st["n_launches"]
        local_bytes = bpi["iter"] * st["newton_iters"] + bpi["companion_per_step"] * st["steps"]
        if split and nonlinear and st["newton_iters"] > st["steps"]:
            # Split schedule, non-linear circuit: the Newton iterations after the first of a time point stamp only what depends on x
            # (pe_front.hpp stamp_dynamic_chunk) -- counted as the junction state (72 B) + one matrix slot and one right-hand-side
            # entry (16 B) per junction instead of B_stamp: bytes the path does not move are not credited to it.
            local_bytes -= (bpi["stamp"] - 88 * info["n_d"]) * (st["newton_iters"] - st["steps"])
        path_achieved = local_bytes / (st["gpu_ms"] * 1e-3) / 1e9
        if split:
            # Split schedule (one launch per phase): the dominant kernel is k_m2_factor_parts -- assembly + LU of every front below
            # the top levels with the right-hand side carried along (fused forward substitution).  Its algorithmic bytes per
            # Newton iteration: the factor bytes and the forward half of the solve bytes, scaled by the share of the factor
            # entries those fronts hold.  Duration: HIP events recorded around that launch alone (pe_kernels.hip m2_iteration).
            share = 1.0 - info["nnz_lu_stored_top"] / max(1, info["nnz_lu_stored"])
            dom_bytes_iter = share * (bpi["factor"] + 0.5 * bpi["solve"])
            # ... of which the bytes these kernels really move: the forward substitution is fused into the factorisation (the
            # right-hand side rides along as one more column of every front), so the factor panels are NOT read a second time --
            # the 1/2 B_solve of the SURVEY formula is credit for traffic that does not exist (as the un-moved stamp bytes above)
            dom_bytes_iter_moved = share * bpi["factor"]
            kernel = "k_m2_factor_parts<%d>" % (4 if info["n_wavefronts"] <= 4 else 2)
            if info.get("n_quad_fronts", 0) > 0:
                # round 3: the wave fronts that qualify run four instances per wavefront in their own launch right before -- the
                # HIP events bracket the pair, the bytes are those of the same fronts as before
                kernel = "k_m2_factor_quads + " + kernel
        else:
            share = 1.0
            dom_bytes_iter = dom_bytes_iter_moved = bpi["iter"]
            kernel = "k_tr_steps"
        dom_ms = st["dominant_ms"]
        dom_launches = max(1, st["dominant_launches"])
        dom_bytes = dom_bytes_iter * st["newton_iters"] + (0 if split else bpi["companion_per_step"] * st["steps"])
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        achieved_moved = (dom_bytes_iter_moved * st["newton_iters"] + (0 if split else bpi["companion_per_step"] * st["steps"])) / (dom_ms * 1e-3) / 1e9
        line = {
            "metric": "transient steps/sec (+ Newton iters/sec), 10k-node RC mesh",
            "value": steps_total / el,
            "unit": "instance-steps/s",
            "n_gpus": ranks_seen,
            "ranks_seen": ranks_seen,
            "collective_backend": (backend if dist is not None else None),
            "build_id": pe.ffi.build_id(),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "newton_iters_per_s": iters_total / el,
            "newton_iters_per_step": iters_total / max(1.0, steps_total),
            "config": {"workload": f"M10k{'-NL' if nonlinear else ''}: {W}x{W} RC mesh Monte-Carlo sweep, {total} instances in total, dt=1e-10, "
                                   f"{'1249 diodes + VAC 2V 100MHz' if nonlinear else 'VDC 1V'} (BASELINE.json configs[4]; configs[2] = one instance: single_circuit)",
                       "rows": info["rows"], "nnz_a": info["nnz_a"], "nnz_lu": info["nnz_lu"], "instances_total": total, "instances_rank0": B,
                       "parallelism": f"contiguous blocks of ceil({total}/{world}) instances per GPU, no data-path collective, final statistics all-reduce"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "frac_moved": achieved_moved / HBM_PEAK_GBS, "achieved_moved": achieved_moved,
                         "frac_note": "frac: SURVEY.md 8d bytes (share x (B_factor + 1/2 B_solve)); frac_moved: without the 1/2 B_solve the fused forward substitution never moves",
                         "traffic": None, "kernel": kernel,
                         "launches": dom_launches, "avg_launch_ms": dom_ms / dom_launches, "bytes_per_launch": dom_bytes / dom_launches,
                         "bytes_per_newton_iter": dom_bytes_iter, "share_of_factor_entries": share,
                         "time_share_of_hot_path": dom_ms / st["gpu_ms"],
                         "whole_path": {"achieved": path_achieved, "frac": path_achieved / HBM_PEAK_GBS, "bytes_per_newton_iter": local_bytes / max(1, st["newton_iters"]),
                                        "gpu_ms": st["gpu_ms"], "schedule": "split: k_m2_eval/stamp/winit/factor_parts/factor_top x levels/"
                                        "solve_top x levels/backward_parts/finish per Newton iteration (stamp: x-dependent slots only after the first iteration of a time point)" if split else "resident k_tr_steps"}},
            "reduce_ms": reduce_ms,
            "allreduce_ms": allreduce_ms,
            "stats_checksum": float(np.sum(stats[0])),
            "engine": dict({k: info[k] for k in ("n_fronts", "max_front", "tree_depth", "nnz_lu_stored", "factor_flops", "bytes_per_instance")},
                           stored_over_structural=info["nnz_lu_stored"] / max(1, info["nnz_lu"]), n_wave_fronts=info.get("n_wave_fronts"),
                           n_quad_fronts=info.get("n_quad_fronts")),
        }
        ceiling = measured_hbm_ceiling(eng) if world == 1 else None
        if isinstance(ceiling, float):
            line["roofline"]["measured_ceiling"] = {"GBps": ceiling, "frac_of_measured": achieved / ceiling,
                                                    "guide_copy_GBps": HBM_GUIDE_COPY_GBS, "frac_of_guide_copy": achieved / HBM_GUIDE_COPY_GBS,
                                                    "how": "device-to-device stream copy of 2 GiB by the engine's copy kernel (contiguous chunk per workgroup, 8 x 16 B "
                                                           "non-temporal loads in flight per lane; read + write bytes / HIP-event time), this run; guide_copy: the float4 "
                                                           "copy of MI355X_MICROARCH.md, not reached by any copy shape tried on these boxes (scripts/copy_sweep.hip)"}
        elif ceiling:
            line["roofline"]["measured_ceiling"] = {"GBps": None, "how": ceiling}
        # HBM bytes of the dominant kernel from the PMC counters: OFFLINE figure (rocprofv3 cannot profile the process that prints
        # this line) -- two separate --pmc passes of this same command, corrected with the factors calibrated on this engine's
        # access shapes (scripts/hbm_calib.hip); copied from the committed summary only when it was measured for this configuration
        line["roofline"].update(select_traffic(os.path.join(ROOT, "profiles"), line["build_id"], B, nonlinear, W, kernel, dom_bytes / dom_launches))
        if world == 1 and not args.no_single:
            # extra (outside the timed region): ONE M10k circuit on the GPU -- the latency-bound case of config C3
            try:
                line["single_circuit"] = single_circuit_numbers(pe, W, dt, device_index)
            except Exception as e:
                line["single_circuit"] = {"error": str(e)}
            line["other_configs"] = other_config_numbers(pe, device_index)
            if scaling == "strong" and total == 1024:
                sr = shard_rates(pe, W, dt, nonlinear, device_index, total)
                line["shard_rates"] = sr
                r128 = sr.get("128", {}).get("instance_steps_per_s")
                if r128:
                    line["shard_rates"]["projected_8gpu_over_1gpu"] = 8.0 * r128 / line["value"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(deck, dt, nonlinear, args.cpu_steps)
                line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
                allc = cpu_baseline_all_cores(W, dt, nonlinear, max(10, args.cpu_steps // 3), pe)
                if allc:
                    line["cpu_baseline_all_cores"] = allc
                    line["speedup_vs_cpu_all_cores"] = line["value"] / allc["value"]
            except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU line
                line["cpu_baseline"] = {"value": None, "unit": "steps/s", "cores": 1, "kind": "reference", "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    eng.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
