# developer tool: memory-pipeline PMC passes on the default bench (run on the GPU box); <= 2 counters per block per pass
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mem
mkdir -p $O
B="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single"
i=0
for set in "TA_TA_BUSY_sum TA_FLAT_WAVEFRONTS_sum GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "TCC_HIT_sum TCC_MISS_sum" "SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  echo "pass $i: $set" >> $O/progress.txt
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -- $B > $O/p$i.log 2>&1 || echo "pass $i failed" >> $O/progress.txt
done
cat $O/progress.txt
