# developer tool: kernel-level statistics at 128 instances (the per-GPU share of the sweep at 8 GPUs) + per-launch durations of the top levels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_b128
rm -rf $O; mkdir -p $O
BATCHES=128 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/scripts/gpu_time.py > $O/run.log 2>&1
cat $O/*/*kernel_stats.csv | cut -c1-150
grep " NL " $O/run.log | cut -c1-200
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$O/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 dispatches of the run: one Newton iteration in launch order
for r in rows[-44:]:
    print(r["Kernel_Name"][:60].ljust(60), r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
