# developer tool: results of an iteration published into pinned host memory and polled (1) against copy + stream synchronisation (0)
for B in 1 128 1024; do for rep in 1 2 3; do for v in 0 1; do
  echo -n "B=$B publish=$v: "; PHY_ENGINE_HIP_PUBLISH=$v BATCHES=$B timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-110
done; done; done
