for rep in 1 2 3 4; do
for v in "4 2048" "8 4096"; do
  set -- $v
  PHY_ENGINE_HIP_PARTS=$1 PHY_ENGINE_HIP_TOP_HALF_WGS=$2 BATCHES=1024 timeout -k 10 500 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-160 | sed "s/^/parts=$1 half_wgs=$2: /"
done; done
