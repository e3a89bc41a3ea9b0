for rep in 1 2; do
for v in "QUAD=1" "QUAD=0" "QUAD_BACK=0" "ABSORB_M=32 PHY_ENGINE_HIP_RELAX_SMALL=4"; do
  env $(echo $v | sed 's/^/PHY_ENGINE_HIP_/') BATCHES=128,256 timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-125 | sed "s/^/$v: /"
done; done
