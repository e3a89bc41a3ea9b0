# round 4: knob A/B on one box: lines "B knob=value ..." from $1 (file) each run REPS times, interleaved.   gpurun -- 'REPS=2 bash scripts/r4_knobs.sh scripts/r4_knobs_a.txt'
R=$GRAFT_REPO_ROOT
cd $R
for rep in $(seq 1 ${REPS:-2}); do
while read -r B KN; do
  [ -z "$B" ] && continue
  envs=""; for kv in $KN; do envs="$envs PHY_ENGINE_HIP_$kv"; done
  echo -n "B=$B $KN: "; env $envs NLONLY=1 BATCHES=$B timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | sed 's/.*: \([0-9.]*\) ms\/step, \([0-9.]*\) ms\/iter.*dominant kernel \([0-9.]*\) ms.*/\1 ms\/step \2 ms\/iter pair \3/'
done < $1
done
