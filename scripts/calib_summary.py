#!/usr/bin/env python3
"""Developer tool: profiles/r02_hbm_calib.json from one run of scripts/calib.sh (gpurun_out/calib_*): the on-box HBM ceiling of the
stream kernels and the FETCH_SIZE / WRITE_SIZE factors for this engine's access shapes (bytes moved / bytes the counter reports).

  python scripts/calib_summary.py [gpurun_out]"""
import csv, glob, json, os, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out")
bare = [json.loads(l) for l in open(os.path.join(src, "calib_bare.jsonl")) if l.startswith('{"kernel"')]


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                out.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch, write = per_kernel("calib_fetch", "FETCH_SIZE"), per_kernel("calib_write", "WRITE_SIZE")
rows = []
for b in bare:
    k = b["kernel"]
    row = dict(b)
    if b["bytes_read"] and fetch.get(k):
        row["FETCH_SIZE_bytes"] = fetch[k]
        row["fetch_factor"] = b["bytes_read"] / fetch[k]
    if b["bytes_written"] and write.get(k):
        row["WRITE_SIZE_bytes"] = write[k]
        row["write_factor"] = b["bytes_written"] / write[k]
    rows.append(row)
out = {
    "device": "MI355X (gfx950), ROCm 7.2, one gpurun box", "program": "scripts/hbm_calib.hip (4 GiB per buffer)",
    "ceiling_GBps": {r["kernel"]: r["GBps"] for r in rows},
    "kernels": rows,
    "fetch_factor_for_this_engine": next(r["fetch_factor"] for r in rows if r["kernel"] == "k_seg128"),
    "write_factor_for_this_engine": next(r["write_factor"] for r in rows if r["kernel"] == "k_wseg128"),
    "note": "FETCH_SIZE reports half of the bytes read for every read shape this engine uses (16 B/lane stream, 8 B/lane stream, 8 B/lane in "
            "128-byte quarter-wave segments = the update-matrix tile loads): factor 2.0, as MI355X_MICROARCH.md states for 16 B/lane; "
            "WRITE_SIZE is exact for all three store shapes.  scripts/pmc_traffic.py applies these factors.",
}
json.dump(out, open(os.path.join(root, "profiles", "r02_hbm_calib.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
