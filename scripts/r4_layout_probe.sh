# round 4: per-layout phase clocks of the cooperative fronts under other workgroup shapes (does the whole-front layout of a larger LDS share pay
# for the fronts of order 72..100?).   gpurun -- 'bash scripts/r4_layout_probe.sh'
R=$GRAFT_REPO_ROOT
cd $R
run() { echo "== $*"; env "$@" NLONLY=1 COOP=1 BATCHES=${B:-1024} timeout -k 10 300 python scripts/gpu_time.py 2>&1 | cut -c1-330; }
run X=0
run PHY_ENGINE_HIP_RESIDENT=2
run PHY_ENGINE_HIP_WAVES=8 PHY_ENGINE_HIP_RESIDENT=2
run PHY_ENGINE_HIP_WAVES=8 PHY_ENGINE_HIP_RESIDENT=2 PHY_ENGINE_HIP_QUAD=0
