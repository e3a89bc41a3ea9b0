# developer tool (round 2): one-knob-at-a-time sweep around the split-schedule geometry at 1024 instances (M10k-NL) after the
# round-2 kernel work lowered the per-front overhead (the best amalgamation / block sizes may have moved)
export B=1024
run() { echo -n "$1: "; env $1 CFGS=${2:-4:10} timeout -k 10 200 python scripts/gpu_m2.py 2>&1 | tail -1; }
run "X=default"
run "PHY_ENGINE_HIP_MAX_PIVOTS=24"
run "PHY_ENGINE_HIP_MAX_PIVOTS=16"
run "PHY_ENGINE_HIP_MAX_PIVOTS=40"
run "PHY_ENGINE_HIP_WAVE_P=8"
run "PHY_ENGINE_HIP_WAVE_P=12"
run "PHY_ENGINE_HIP_WAVE_P=24"
run "PHY_ENGINE_HIP_ABSORB_M=24"
run "PHY_ENGINE_HIP_ABSORB_M=28"
run "PHY_ENGINE_HIP_ND_LEAF=12"
run "PHY_ENGINE_HIP_ND_LEAF=16"
run "PHY_ENGINE_HIP_ND_LEAF=32"
run "PHY_ENGINE_HIP_ND_LEAF=48"
run "PHY_ENGINE_HIP_WAVE_M=30"
run "PHY_ENGINE_HIP_CUT_X10=5"
run "PHY_ENGINE_HIP_CUT_X10=20"
run "PHY_ENGINE_HIP_PART_CUT_X10=8" 4:8
run "PHY_ENGINE_HIP_PART_CUT_X10=15" 4:15
run "X=parts8" 8:10
