for rep in 1 2 3; do
for v in 2048 1280; do
  PHY_ENGINE_HIP_TOP_HALF_WGS=$v BATCHES=128,256 timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-125 | sed "s/^/half_wgs=$v: /"
done; done
