# developer tool: parity spot check + 1024-instance timing with phase breakdown
python scripts/gpu_check.py > gpurun_out/check.log 2>&1 && tail -3 gpurun_out/check.log &&
COOP=1 BATCHES=1024 timeout -k 10 400 python scripts/gpu_time.py > gpurun_out/phases_split.log 2>&1; tail -7 gpurun_out/phases_split.log
