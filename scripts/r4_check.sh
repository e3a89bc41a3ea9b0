# round 4: GPU suite + LDS out-of-allocation probe + default bench -> gpurun_out/<tag>_*
TAG=${1:-r04b}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd $R
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $O/${TAG}_gpu_tests.log 2>&1; tail -3 $O/${TAG}_gpu_tests.log
(cd scripts && timeout -k 10 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe lds_oob_probe.hip && timeout -k 10 60 /tmp/lds_probe && timeout -k 10 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_probe2 lds_oob_probe2.hip && timeout -k 10 60 /tmp/lds_probe2) 2>&1 | grep workgroups > $O/${TAG}_lds_probe.log; cat $O/${TAG}_lds_probe.log
timeout -k 10 500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err && python3 -c "
import json,sys
d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms_per_step',d['ms_per_step'],'pair_ms',d['roofline']['avg_launch_ms'],'frac',d['roofline']['frac'],'single',d['single_circuit']['nl_steps_per_s'],'shard',{k:(v['instance_steps_per_s'] if isinstance(v,dict) else v) for k,v in d['shard_rates'].items()})"
