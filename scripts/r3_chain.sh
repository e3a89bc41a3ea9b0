# developer tool (round 3): durations of the top-level launches with / without the LDS-continued chain links (PHY_ENGINE_HIP_TOP_CHAIN_LDS)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_chain
rm -rf $O; mkdir -p $O
for B in 128 1; do for K in 0 1; do
PHY_ENGINE_HIP_TOP_CHAIN_LDS=$K BATCHES=$B timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b${B}_k$K -- python3 $R/scripts/gpu_time.py > $O/run_b${B}_k$K.log 2>&1 &&
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/b${B}_k$K/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tops = [r for r in rows[-60:] if "factor_top" in r["Kernel_Name"] or "solve_top" in r["Kernel_Name"]]
print("B=$B chain_lds=$K:", " ".join("%s%.0f" % ("F" if "factor" in r["Kernel_Name"] else "s", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in tops[-30:]))
PY
done; done
