# developer tool: kernel-level statistics of the single-circuit case (one M10k-NL instance)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_b1
rm -rf $O; mkdir -p $O
BATCHES=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/gpu_time.py > $O/run.log 2>&1
cat $O/*/*kernel_stats.csv | cut -c1-150
grep " NL " $O/run.log | cut -c1-200
