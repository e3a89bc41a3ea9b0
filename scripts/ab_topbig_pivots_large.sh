for rep in 1 2 3; do
for mp in 64 48 40; do
  PHY_ENGINE_HIP_TOP_MAX_PIVOTS=$mp BATCHES=512,1024 timeout -k 10 500 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-160 | sed "s/^/top_max_pivots=$mp: /"
done; done
