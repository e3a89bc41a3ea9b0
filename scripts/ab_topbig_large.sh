# developer tool: the regrouped top (knob TOP_BIG) on / off at the larger sweep sizes, interleaved on one box
for rep in 1 2 3; do
for v in 1 0; do
  PHY_ENGINE_HIP_TOP_BIG=$v BATCHES=${BATCHES:-512,1024} timeout -k 10 500 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-160 | sed "s/^/top_big=$v: /"
done; done
