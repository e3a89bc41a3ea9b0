export B=1024
PHY_ENGINE_HIP_SPLIT=0 CFGS=1:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -1
PHY_ENGINE_HIP_SPLIT=1 CFGS=1:10,2:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -2
export B=256
PHY_ENGINE_HIP_SPLIT=0 CFGS=1:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -1
PHY_ENGINE_HIP_SPLIT=1 CFGS=1:10,2:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -2
