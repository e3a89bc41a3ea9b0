for rep in 1 2; do
for v in 1 0; do
  PHY_ENGINE_HIP_TOP_BIG=$v BATCHES=16,32,64 timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-125 | sed "s/^/top_big=$v: /"
done; done
