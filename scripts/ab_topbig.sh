# developer tool: top fronts regrouped against a CU's LDS (pe_engine_policy.cpp regroup_wide_top): knob TOP_BIG 0 = off, 2 = whole-CU levels
# only, 1 = + half-CU levels (default); TOP_MAX_PIVOTS 48 / 64 -- interleaved on one box
for rep in 1 2 3; do
for v in "1 48" "0 48" "2 48" "1 64"; do
  set -- $v
  PHY_ENGINE_HIP_TOP_BIG=$1 PHY_ENGINE_HIP_TOP_MAX_PIVOTS=$2 BATCHES=${BATCHES:-1,128,256} timeout -k 10 400 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-120 | sed "s/^/top_big=$1 max_pivots=$2: /"
done; done
