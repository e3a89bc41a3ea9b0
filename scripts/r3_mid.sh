# developer tool (round 3): the MID-front launch on / off (PHY_ENGINE_HIP_MID): parity spot check + kernel durations at 1 024 instances
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_mid
rm -rf $O; mkdir -p $O
for K in 0 1; do
PHY_ENGINE_HIP_MID=$K python3 $R/scripts/gpu_check.py 2>&1 | tail -2
PHY_ENGINE_HIP_MID=$K BATCHES=1024 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$K -- python3 $R/scripts/gpu_time.py > $O/run_k$K.log 2>&1
grep " NL " $O/run_k$K.log | cut -c1-150
grep -h "factor_quads\|factor_mid\|factor_parts\|backward" $O/k$K/*/*kernel_stats.csv | cut -c1-110
done
