# round 4: A/B of two library builds on one box (phy-engine_amd/libpe_hip_A.so = the build before the change), interleaved, at several sweep
# sizes: ms per Newton iteration / ms of the dominant pair.   gpurun -- 'BATCHES="1024 128 1" REPS=3 bash scripts/r4_ab.sh'
R=$GRAFT_REPO_ROOT
cd $R
for B in ${BATCHES:-1024 128 1}; do
for rep in $(seq 1 ${REPS:-2}); do
for v in A B; do
  if [ $v = A ]; then export PE_HIP_LIB=$R/phy-engine_amd/libpe_hip_A.so; else unset PE_HIP_LIB; fi
  echo -n "B=$B $v: "; NLONLY=1 BATCHES=$B timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | sed 's/.*: \([0-9.]*\) ms\/step, \([0-9.]*\) ms\/iter.*dominant kernel \([0-9.]*\) ms.*/\1 ms\/step \2 ms\/iter pair \3/'
done; done; done
