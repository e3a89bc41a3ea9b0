# developer tool: workgroup size x residency x parts of the split schedule at 1024 instances
export B=1024
run() { echo -n "$1: "; env $1 CFGS=${2:-4:10} timeout -k 10 200 python scripts/gpu_m2.py 2>&1 | tail -1; }
run "X=default"
run "PHY_ENGINE_HIP_WAVES=2 PHY_ENGINE_HIP_RESIDENT=8" 4:10
run "PHY_ENGINE_HIP_WAVES=2 PHY_ENGINE_HIP_RESIDENT=8" 8:10
run "PHY_ENGINE_HIP_WAVES=2 PHY_ENGINE_HIP_RESIDENT=8" 6:10
run "PHY_ENGINE_HIP_WAVES=3 PHY_ENGINE_HIP_RESIDENT=5" 5:10
run "PHY_ENGINE_HIP_WAVES=8 PHY_ENGINE_HIP_RESIDENT=2" 2:10
