export B=1024
CFGS=2:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -1
PHY_ENGINE_HIP_WAVE_P=32 PHY_ENGINE_HIP_MAX_PIVOTS=48 CFGS=2:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -1 | sed 's/^/wp32 maxp48: /'
PHY_ENGINE_HIP_WAVE_P=16 PHY_ENGINE_HIP_MAX_PIVOTS=48 CFGS=2:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -1 | sed 's/^/wp16 maxp48: /'
PHY_ENGINE_HIP_WAVE_P=12 PHY_ENGINE_HIP_MAX_PIVOTS=24 CFGS=2:10,4:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -2 | sed 's/^/wp12 maxp24: /'
CFGS=1:10,4:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -2
