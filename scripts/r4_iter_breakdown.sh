# round 4: per-kernel breakdown of ONE Newton iteration (kernel-trace time stamps, last iteration that runs every launch at full population = first
# iteration of the last time point) for a list of configurations "B:knob=value,knob=value".   gpurun -- 'bash scripts/r4_iter_breakdown.sh "1024: 1024:PARTS=8,TOP_HALF_WGS=4096"'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for cfg in $1; do
i=$((i+1))
B=${cfg%%:*}; KN=${cfg#*:}
O=$R/gpurun_out/r4_iter_$i
rm -rf $O; mkdir -p $O
envs=""
for kv in $(echo $KN | tr ',' ' '); do [ -n "$kv" ] && export PHY_ENGINE_HIP_$kv && envs="$envs $kv"; done
NLONLY=1 BATCHES=$B timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/scripts/gpu_time.py > $O/run.log 2>&1 || { tail -5 $O/run.log; exit 1; }
for kv in $(echo $KN | tr ',' ' '); do [ -n "$kv" ] && unset PHY_ENGINE_HIP_${kv%%=*}; done
echo "== B=$B $envs"; grep " NL " $O/run.log | cut -c1-170
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("$O/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "pe::k_" in r["Kernel_Name"]][-6000:]
starts = [i for i, r in enumerate(rows) if "k_m2_eval" in r["Kernel_Name"]]
its = list(zip(starts[:-1], starts[1:]))
# the longest of the last 12 iterations = a first iteration of a time point (everything active, full stamp)
best = max(its[-12:], key=lambda ab: int(rows[ab[1]]["Start_Timestamp"]) - int(rows[ab[0]]["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows[best[0]:best[1]]:
    k = r["Kernel_Name"].replace("pe::", "").split("(")[0][:40]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(k, [0, 0.0, []]); a[0] += 1; a[1] += d; a[2].append(round(d))
tot = sum(a[1] for a in agg.values()); wall = (int(rows[best[1]]["Start_Timestamp"]) - int(rows[best[0]]["Start_Timestamp"])) / 1e3
print(f"   one full iteration: {best[1]-best[0]} launches, kernels {tot:.0f} us, wall {wall:.0f} us")
for k, a in agg.items(): print(f"   {k.ljust(40)} x{a[0]:2d} {a[1]:8.1f} us  {a[2] if a[0] > 1 else ''}")
PY
done
