# developer tool (round 2): single M10k-NL circuit -- parts per instance / pivots per front with the round-2 kernels
export B=1
run() { echo -n "B=1 $1 [$2]: "; env $1 CFGS=$2 timeout -k 10 200 python scripts/gpu_m2.py 2>&1 | tail -1 | cut -c1-120; }
run "X=default" 48:10
run "X=p" 8:10
run "X=p" 16:10
run "X=p" 24:10
run "X=p" 32:10
run "X=p" 64:10
run "PHY_ENGINE_HIP_MAX_PIVOTS=32" 48:10
run "PHY_ENGINE_HIP_MAX_PIVOTS=24" 48:10
run "PHY_ENGINE_HIP_MAX_PIVOTS=64" 48:10
run "PHY_ENGINE_HIP_ND_LEAF=24" 48:10
run "X=cut" 48:15
run "X=cut" 48:7
