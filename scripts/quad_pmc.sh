# developer tool (GPU box): SQ instruction counters of the factor kernels for library variants V (phy-engine_amd/libpe_hip_<v>.so)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for v in ${V:-head w3}; do
  export PE_HIP_LIB=$R/phy-engine_amd/libpe_hip_$v.so
  O=$R/gpurun_out/quadpmc_$v; rm -rf $O; mkdir -p $O
  B=1024 STEPS=4 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $O/p2 -- python3 $R/scripts/one_sweep.py > $O/p2.log 2>&1
  B=1024 STEPS=4 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d $O/p1 -- python3 $R/scripts/one_sweep.py > $O/p1.log 2>&1
  echo "== $v"; python3 $R/scripts/pmc_sum.py $O | grep -A17 -E "factor_quads" 
done
