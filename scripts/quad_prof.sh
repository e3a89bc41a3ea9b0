# developer tool (GPU box): per-kernel times of one M10k-NL sweep with / without the lane-group kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for q in 1 0; do
  export PHY_ENGINE_HIP_QUAD=$q
  rm -rf $O/quadprof_$q; B=${B:-1024} STEPS=8 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/quadprof_$q -- python3 $R/scripts/one_sweep.py > $O/quadprof_$q.log 2>&1
  f=$(ls $O/quadprof_$q/*/*kernel_stats.csv | head -1); echo "== quad=$q"; cut -d, -f1-8 $f | head -12
done
