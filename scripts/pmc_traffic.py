#!/usr/bin/env python3
"""Developer tool: profiles/<tag>_pmc_traffic.json from the two rocprofv3 PMC passes of scripts/profile_round.sh.

  TAG=r03 python scripts/pmc_traffic.py gpurun_out/<tag>_fetch gpurun_out/<tag>_write gpurun_out/<tag>_bench.json [kernel substring]

The dominant "kernel" may be a PAIR of launches (round 3: "k_m2_factor_quads + k_m2_factor_parts"): the counters of both are summed
and divided by the number of pairs (= dispatches of the last one).

Sums FETCH_SIZE / WRITE_SIZE (KiB) over the dispatches of the dominant kernel and divides by their count: HBM bytes per
launch, the same normalisation as roofline.achieved in bench.py.  FETCH_SIZE / WRITE_SIZE (KiB) are converted with the factors
calibrated on this engine's access shapes (profiles/r02_hbm_calib.json from scripts/hbm_calib.hip: FETCH x 2.0, WRITE x 1.0)."""
import csv, glob, json, os, sys

fetch_dir, write_dir, bench_json = sys.argv[1:4]
line = json.loads(open(bench_json).read().strip().splitlines()[-1])
kernel = sys.argv[4] if len(sys.argv) > 4 else line["roofline"]["kernel"].split("<")[0]
kernels = [k.strip() for k in kernel.split("+")]
TAG = os.environ.get("TAG", "r04")


def newest(d):
    """gpurun merges every call's output into gpurun_out/: only the newest pass of a directory counts."""
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    return [max(fs, key=os.path.getmtime)] if fs else []


def total(d, counter):
    s, n = 0.0, 0
    for f in newest(d):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            if any(k in row["Kernel_Name"] for k in kernels):
                s += float(row["Counter_Value"])
            if kernels[-1] in row["Kernel_Name"]:
                n += 1
    return s, n


f, nf = total(fetch_dir, "FETCH_SIZE")
w, nw = total(write_dir, "WRITE_SIZE")
assert nf and nf == nw, (nf, nw)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cal = json.load(open(os.path.join(root, "profiles", "r02_hbm_calib.json")))
ff, wf = cal["fetch_factor_for_this_engine"], cal["write_factor_for_this_engine"]
out = {
    "build_id": line.get("build_id"),  # the library the passes ran (bench.py copies roofline.traffic only on a match with its own)
    "instances_per_gpu": line["config"]["instances_rank0"], "nonlinear": "-NL" in line["config"]["workload"], "mesh": 100,
    "kernel": kernel, "dispatches": nf, "FETCH_SIZE_KiB_per_launch": f / nf, "WRITE_SIZE_KiB_per_launch": w / nw,
    "fetch_factor": ff, "write_factor": wf,
    "read_bytes_per_launch": f / nf * 1024.0 * ff, "written_bytes_per_launch": w / nw * 1024.0 * wf,
    "hbm_bytes_per_launch": f / nf * 1024.0 * ff + w / nw * 1024.0 * wf,
    "algorithmic_bytes_per_launch": line["roofline"]["bytes_per_launch"],
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes of `python3 bench.py --no-cpu-baseline --no-single`, averaged over "
            "every dispatch of the kernel (warm-up included: same work per launch up to the converged instances); KiB x 1024 x the factor calibrated "
            "for this engine's access shapes on the same kind of box (profiles/r02_hbm_calib.json: FETCH_SIZE reports half of the bytes read)",
}
out["traffic_over_algorithmic"] = out["hbm_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
json.dump(out, open(os.path.join(root, "profiles", TAG + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))


def slim(d, counter, dst):
    """profiles/<tag>_pmc_<counter>.csv: the engine's kernels only, the columns the summary uses."""
    cols = ["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]
    with open(dst, "w", newline="") as o:
        wr = csv.writer(o)
        wr.writerow(cols)
        for f in newest(d):
            for row in csv.DictReader(open(f)):
                if "pe::k_" in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    wr.writerow([row[c] for c in cols])


slim(fetch_dir, "FETCH_SIZE", os.path.join(root, "profiles", TAG + "_pmc_FETCH_SIZE.csv"))
slim(write_dir, "WRITE_SIZE", os.path.join(root, "profiles", TAG + "_pmc_WRITE_SIZE.csv"))
