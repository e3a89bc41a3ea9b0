for rep in 1 2 3; do
for v in 1280 2048 4096; do
  PHY_ENGINE_HIP_TOP_HALF_WGS=$v BATCHES=512,1024 timeout -k 10 500 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-160 | sed "s/^/half_wgs=$v: /"
done; done
