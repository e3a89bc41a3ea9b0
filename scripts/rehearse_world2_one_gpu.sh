# Rehearsal of the N > 1 path of bench.py on a ONE-GPU box (the 8-GPU scaling run is the driver's): two ranks, one GPU, gloo for the
# exchange step (RCCL needs one device per rank), launched by torch.distributed.run BEFORE anything touches the GPU in the launcher.
# Checks: both ranks join (ranks_seen 2), contiguous blocks of ceil(1024 / 2) instances, the statistics checksum of the two-rank run
# equals the one-rank run's, allreduce_ms is reported.   gpurun -- 'bash scripts/rehearse_world2_one_gpu.sh r04'
TAG=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
export MASTER_ADDR=127.0.0.1 PE_BENCH_BACKEND=gloo PE_BENCH_DEVICE=0 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29741 bench.py --gpus 2 --steps 10 --warmup 2 \
    --no-cpu-baseline --no-single > $O/${TAG}_world2_one_gpu.json 2> $O/${TAG}_world2_one_gpu.err &&
timeout -k 10 300 python bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-single > $O/${TAG}_world1_same_box.json 2> $O/${TAG}_world1_same_box.err &&
python3 - <<PY
import json
two = json.loads([l for l in open("$O/${TAG}_world2_one_gpu.json") if l.startswith("{")][-1])
one = json.loads([l for l in open("$O/${TAG}_world1_same_box.json") if l.startswith("{")][-1])
assert two["ranks_seen"] == 2 and two["n_gpus"] == 2 and two["config"]["instances_rank0"] == 512, two
assert abs(two["stats_checksum"] - one["stats_checksum"]) <= 1e-9 * abs(one["stats_checksum"]), (two["stats_checksum"], one["stats_checksum"])
print("world 2 on one GPU: ranks_seen", two["ranks_seen"], "value", round(two["value"]), "allreduce_ms", round(two["allreduce_ms"], 3), "| world 1:", round(one["value"]), "checksums equal")
PY
