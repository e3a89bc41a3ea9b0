# developer tool (round 2): parts per instance at the per-GPU batch sizes of the strong-scaling run (1024 / N instances per GPU)
run() { echo -n "B=$B $1 [$2]: "; env $1 CFGS=$2 timeout -k 10 200 python scripts/gpu_m2.py 2>&1 | tail -1 | cut -c1-120; }
export B=128
run "X=default" 8:10
run "X=p" 4:10
run "X=p" 6:10
run "X=p" 12:10
run "X=p" 16:10
run "PHY_ENGINE_HIP_MAX_PIVOTS=24" 8:10
export B=256
run "X=default" 4:10
run "X=p" 8:10
run "X=p" 6:10
export B=512
run "X=default" 4:10
run "X=p" 8:10
run "X=p" 2:10
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_b128; mkdir -p $R/gpurun_out/prof_b128
BATCHES=128 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b128 -- python3 $R/scripts/gpu_time.py > $R/gpurun_out/prof_b128/run.log 2>&1
cat $R/gpurun_out/prof_b128/*/*kernel_stats.csv | cut -c1-130 | head -14
