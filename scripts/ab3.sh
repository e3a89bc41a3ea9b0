# developer tool: three-way A/B of library builds on one box, interleaved (default build, and the variants named in LIBS)
for rep in 1 2 3; do
for v in default ${LIBS:-ilp mc}; do
  if [ $v = default ]; then unset PE_HIP_LIB; else export PE_HIP_LIB=$GRAFT_REPO_ROOT/phy-engine_amd/libpe_hip_$v.so; fi
  echo -n "$v: "; BATCHES=${BATCHES:-1024} timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-200
done; done
