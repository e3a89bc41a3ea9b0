# developer tool (round 3): G concurrent engines on one GPU for the per-GPU shares of the sweep (128 / 256 / 512 instances)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_groups.log
: > $O
for T in 128 256 512; do
  echo "== total $T, geometry by own batch" >> $O
  TOTAL=$T GROUPS=1,2,4 timeout -k 10 200 python3 $R/scripts/two_engines.py >> $O 2>&1 || exit 1
  echo "== total $T, geometry of the total batch" >> $O
  PHY_ENGINE_HIP_GEOMETRY_BATCH=$T TOTAL=$T GROUPS=2,4 timeout -k 10 200 python3 $R/scripts/two_engines.py >> $O 2>&1 || exit 1
done
cat $O
