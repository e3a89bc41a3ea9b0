for rep in 1 2; do
for v in 1 3 4; do
  PHY_ENGINE_HIP_TOP_BIG=$v BATCHES=512,1024 timeout -k 10 500 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-120 | sed "s/^/top_big=$v: /"
done; done
