# developer tool (round 3): phase clocks at 1 024 instances + one Newton iteration's launches in order at 1 024 and 128 instances
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3_probe
rm -rf $O; mkdir -p $O
BATCHES=1024 COOP=1 timeout -k 10 300 python3 $R/scripts/gpu_time.py > $O/clocks.log 2>&1 && grep -A12 " NL " $O/clocks.log | cut -c1-400 &&
for B in 1024 128; do
BATCHES=$B timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$B -- python3 $R/scripts/gpu_time.py > $O/run_b$B.log 2>&1 &&
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/b$B/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("B=$B: last launches in order")
prev_end = None
for r in rows[-48:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    print(r["Kernel_Name"][:50].ljust(50), r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Workgroup_Size_X", r.get("Workgroup_Size")), "%.1f us (gap %.1f)" % ((e - s) / 1e3, gap))
PY
done
