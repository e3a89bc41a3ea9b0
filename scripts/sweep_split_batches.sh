# developer tool: the 4-wavefront / 4-workgroups-per-CU geometry (PHY_ENGINE_HIP_GEOMETRY_BATCH=1024) for smaller batches in the split schedule
export PHY_ENGINE_HIP_GEOMETRY_BATCH=1024
B=256 CFGS=2:10,4:10,8:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -3
B=128 CFGS=4:10,8:10,16:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -3
B=64 CFGS=8:10,16:10,32:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -3
unset PHY_ENGINE_HIP_GEOMETRY_BATCH
B=128 CFGS=2:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -1 | sed 's/^/own geometry: /'
B=64 CFGS=4:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -1 | sed 's/^/own geometry: /'
