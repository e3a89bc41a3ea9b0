#!/usr/bin/env python3
"""Developer tool: per-kernel sums of the rocprofv3 --pmc passes under a directory (counter_collection.csv files)."""
import csv, glob, os, sys
from collections import defaultdict
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
        seen.add((k, row["Dispatch_Id"]))
    for k, _ in seen:
        cnt[k] = max(cnt[k], sum(1 for kk, _ in seen if kk == k))
for k in sorted(tot, key=lambda k: -tot[k].get("SQ_BUSY_CYCLES", 0.0)):
    if not k.startswith("pe::"):
        continue
    print(f"{k}  dispatches={cnt[k]}")
    for c in sorted(tot[k]):
        print(f"    {c:32s} {tot[k][c]:.6g}")
