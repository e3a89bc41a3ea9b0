# developer tool: part_cut sweep of the split schedule under the default geometry policy
B=1024 CFGS=4:8,4:10,4:12,4:15 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -4
B=256 CFGS=4:8,4:10,4:15 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -3
B=128 CFGS=8:8,8:10,8:15 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -3
B=16 CFGS=16:8,16:10,16:15,32:10 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -4
B=1 CFGS=48:10,48:15,32:10,32:15 timeout -k 10 400 python scripts/gpu_m2.py 2>&1 | tail -4
