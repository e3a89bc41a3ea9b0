# developer tool (GPU box): kernel times of library variants phy-engine_amd/libpe_hip_<v>.so (V="w3 w4 ..."), PHY_ENGINE_HIP_QUAD knobs K
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp && export TMPDIR=/tmp
for v in ${V:-w3 w4}; do
  export PE_HIP_LIB=$R/phy-engine_amd/libpe_hip_$v.so
  for q in ${K:-1}; do
    export PHY_ENGINE_HIP_QUAD=$q
    rm -rf $O/quadvar_${v}_$q
    B=${B:-1024} STEPS=${STEPS:-6} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/quadvar_${v}_$q -- python3 $R/scripts/one_sweep.py > $O/quadvar_${v}_$q.log 2>&1
    f=$(ls $O/quadvar_${v}_$q/*/*kernel_stats.csv | head -1); echo "== $v quad=$q"; grep -E "factor_quads|factor_parts|backward_parts" $f | cut -d, -f1-4,6,7
  done
  PHY_ENGINE_HIP_QUAD=33 timeout -k 10 300 python3 $R/scripts/quad_clocks.py 2>&1 | tail -2
done
