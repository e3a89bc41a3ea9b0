export B=1024
run() { echo -n "$1: "; env $1 CFGS=${2:-4:10} timeout -k 10 200 python scripts/gpu_m2.py 2>&1 | tail -1; }
run "PHY_ENGINE_HIP_ND_LEAF=8"
run "PHY_ENGINE_HIP_ND_LEAF=10"
run "PHY_ENGINE_HIP_ND_LEAF=12"
run "PHY_ENGINE_HIP_ND_LEAF=12 PHY_ENGINE_HIP_ABSORB_M=30"
run "PHY_ENGINE_HIP_ND_LEAF=12 PHY_ENGINE_HIP_MAX_PIVOTS=40"
export B=128
run "X=default128" 8:10
run "PHY_ENGINE_HIP_ND_LEAF=12" 8:10
