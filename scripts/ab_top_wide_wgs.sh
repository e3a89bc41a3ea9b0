for rep in 1 2 3; do
for v in 320 512 256; do
  PHY_ENGINE_HIP_TOP_WIDE_WGS=$v BATCHES=128,256 timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-125 | sed "s/^/wide_wgs=$v: /"
done; done
