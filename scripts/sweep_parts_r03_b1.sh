for rep in 1 2; do
for p in 24 32 48 64; do
  PHY_ENGINE_HIP_PARTS=$p BATCHES=1 timeout -k 10 200 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-125 | sed "s/^/parts=$p: /"
done; done
