timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/tb_parity.log 2>&1; tail -2 gpurun_out/tb_parity.log
for rep in 1 2; do for mp in 64 48 32; do
  PHY_ENGINE_HIP_TOP_MAX_PIVOTS=$mp BATCHES=1,128 timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-120 | sed "s/^/top_max_pivots=$mp: /"
done; done
