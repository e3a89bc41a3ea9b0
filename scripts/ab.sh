# developer tool: A/B of two builds of the library on the same box (phy-engine_amd/libpe_hip_A.so vs libpe_hip.so), interleaved
for rep in 1 2; do
for v in A B; do
  if [ $v = A ]; then export PE_HIP_LIB=$GRAFT_REPO_ROOT/phy-engine_amd/libpe_hip_A.so; else unset PE_HIP_LIB; fi
  echo -n "$v: "; BATCHES=${BATCHES:-1024} timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-200
done; done
