# developer tool: default launch policy (no overrides) across batch sizes on the M10k mesh, then the resident kernel for comparison
set -e
python scripts/gpu_check.py > gpurun_out/check.log 2>&1 && tail -3 gpurun_out/check.log
BATCHES=1,16,64,128,256,512,1024 timeout -k 10 500 python scripts/gpu_time.py 2>&1 | cut -c1-150 | tee gpurun_out/auto.log
PHY_ENGINE_HIP_SPLIT=0 BATCHES=1,64,256,1024 timeout -k 10 400 python scripts/gpu_time.py 2>&1 | cut -c1-150 | tee gpurun_out/resident.log
