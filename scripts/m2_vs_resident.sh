# developer tool: split-launch schedule (parts 1 / 2 / 4) vs the resident kernel (PHY_ENGINE_HIP_SPLIT=0) at several batch sizes
for B in 1024 512 256 2048; do
  export B
  PHY_ENGINE_HIP_SPLIT=0 CFGS=1:10 timeout -k 10 300 python scripts/gpu_m2.py 2>&1 | tail -1 | sed 's/^/resident /'
  CFGS=1:10,2:10,4:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -3
done
