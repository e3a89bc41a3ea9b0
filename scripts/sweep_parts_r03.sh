# developer tool: parts per instance re-swept after the top fronts were regrouped (the top got cheaper: does the cut move?)
for rep in 1 2; do
for b in 128 256; do
for p in ${PARTS_LIST:-8 12 16 24 32}; do
  PHY_ENGINE_HIP_PARTS=$p BATCHES=$b timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-125 | sed "s/^/parts=$p: /"
done; done; done
