# developer tool: single-circuit knobs (M10k-NL, one instance)
export B=${B:-1}
run() { echo -n "$1: "; env $1 CFGS=${2:-48:10} timeout -k 10 200 python scripts/gpu_m2.py 2>&1 | tail -1; }
run "X=default"
run "PHY_ENGINE_HIP_MAX_PIVOTS=64"
run "PHY_ENGINE_HIP_MAX_PIVOTS=32"
run "X=default" 32:10
run "X=default" 64:10
run "PHY_ENGINE_HIP_MAX_PIVOTS=64" 64:10
run "X=default" 40:10
