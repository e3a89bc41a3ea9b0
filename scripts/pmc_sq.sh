# developer tool: SQ counter passes on the default bench (run on the GPU box); summary per kernel by scripts/pmc_sum.py
# Every pass keeps its log under a time-stamped name (round 2: a pass that died with a segmentation fault at 12:32 had its p1.log
# overwritten by the 17:40 rerun -- profiles/README.md "the 12:32 rocprofv3 segfault"); a failing pass no longer stops the others.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_sq
TS=$(date +%Y%m%d_%H%M%S)
rm -rf $O/p1 $O/p2 $O/p3; mkdir -p $O
B="python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-single"
pass() { # name, counters...
  n=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- $B > $O/${n}_$TS.log 2>&1
  rc=$?; echo "pass $n rc=$rc (log: gpurun_out/pmc_sq/${n}_$TS.log)"; [ $rc -ne 0 ] && tail -5 $O/${n}_$TS.log
}
pass p1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES
pass p2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass p3 SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python3 $R/scripts/pmc_sum.py $O
