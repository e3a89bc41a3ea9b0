for rep in 1 2; do
for v in A c8 c8_norsm c4 c4_norsm; do
  unset PE_HIP_LIB PHY_ENGINE_HIP_QUAD
  case $v in
    A) export PE_HIP_LIB=$GRAFT_REPO_ROOT/phy-engine_amd/libpe_hip_A.so;;
    c8) export PE_HIP_LIB=$GRAFT_REPO_ROOT/phy-engine_amd/libpe_hip_c8.so;;
    c8_norsm) export PE_HIP_LIB=$GRAFT_REPO_ROOT/phy-engine_amd/libpe_hip_c8.so PHY_ENGINE_HIP_QUAD=65;;
    c4) ;;
    c4_norsm) export PHY_ENGINE_HIP_QUAD=65;;
  esac
  echo -n "$v: "; BATCHES=1024 timeout -k 10 300 python scripts/gpu_time.py 2>&1 | grep " NL " | cut -c1-150
done; done
