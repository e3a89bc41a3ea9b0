# developer tool: idle time BETWEEN the launches of a Newton iteration (kernel trace timestamps) for the single circuit and the
# 128-instance share -- what a captured launch sequence (hipGraph) could win at most.   usage: gpurun -- bash scripts/launch_gaps.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in ${BATCHES_LIST:-1 128}; do
O=$R/gpurun_out/gaps_b$B
rm -rf $O; mkdir -p $O
BATCHES=$B timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/scripts/gpu_time.py > $O/run.log 2>&1 || exit 1
grep " NL " $O/run.log | cut -c1-160
python3 - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("$O/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the iterations of the last (NL, timed) run: split at k_m2_eval
rows = rows[-2000:]
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("pe::k_m2_eval") or "k_m2_eval" in r["Kernel_Name"]]
its = [(a, b) for a, b in zip(starts[:-1], starts[1:])][-30:]
tot_busy = tot_gap = tot_head = n = 0
for a, b in its:
    seg = rows[a:b]
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = sum(max(0, int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) for i in range(len(seg) - 1))
    head = int(rows[b]["Start_Timestamp"]) - int(seg[-1]["End_Timestamp"])  # host decision + first launch of the next iteration
    tot_busy += busy; tot_gap += gaps; tot_head += head; n += 1
print(f"B=$B: {n} iterations, launches/iteration {len(rows[its[-1][0]:its[-1][1]])}: kernels {tot_busy/n/1e3:.1f} us, gaps between launches inside an iteration {tot_gap/n/1e3:.1f} us, iteration-to-iteration gap {tot_head/n/1e3:.1f} us")
a, b = its[-1]
prev = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"   {r['Kernel_Name'][:56].ljust(56)} {(e - s)/1e3:8.1f} us   gap before {((s - prev)/1e3 if prev else 0):6.1f} us")
    prev = e
PY
done
