# developer tool (GPU box): time of k_m2_factor_quads with parts of the front skipped (PHY_ENGINE_HIP_QUAD bits: 2 no elimination, 4 no stores, 8 no children)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for q in ${KNOBS:-1 3 5 9 7 15}; do
  export PHY_ENGINE_HIP_QUAD=$q
  rm -rf $O/quadknob_$q; B=${B:-1024} STEPS=4 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/quadknob_$q -- python3 $R/scripts/one_sweep.py > $O/quadknob_$q.log 2>&1
  f=$(ls $O/quadknob_$q/*/*kernel_stats.csv | head -1); echo -n "quad=$q: "; grep -E "factor_quads|factor_mid" $f | cut -d, -f2-4,6,7
done
