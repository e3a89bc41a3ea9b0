# developer tool (GPU box): steps/s and ms per factor launch pair for environment settings, one per line of $1 (e.g. "PHY_ENGINE_HIP_PARTS=8")
R=$GRAFT_REPO_ROOT
while read -r cfg; do
  [ -z "$cfg" ] && continue
  echo -n "$cfg: "; env $cfg B=${B:-1024} ARMS=${ARMS:-1} timeout -k 10 300 python3 $R/scripts/quad_ab.py 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print(f\"{d['steps_per_s']:.0f} steps/s, {d['dominant_ms_per_launch']:.3f} ms/pair, gpu {d['gpu_ms']:.1f} ms, fronts {d['fronts']}\")
"
done < $1
