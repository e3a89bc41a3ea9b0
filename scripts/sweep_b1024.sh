export BATCHES=1024
CONFIGS=4:35:48:35:24:32,4:35:32:35:24:32,4:35:48:35:16:32,4:35:48:35:32:32,4:35:48:24:24:32,4:35:48:35:24:16,4:30:48:30:24:30,4:35:64:35:24:32 timeout -k 10 600 python scripts/gpu_sweep.py 2>&1 | tail -8
for cx in 5 15 20; do PHY_ENGINE_HIP_CUT_X10=$cx CONFIGS=4:35:48:35:24:32 timeout -k 10 200 python scripts/gpu_sweep.py 2>&1 | tail -1 | sed "s/^/cut=$cx /"; done
