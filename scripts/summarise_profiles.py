"""Reduce the rocprofv3 passes of scripts/collect_profiles.sh to one JSON (per-launch medians)."""
import csv
import glob
import json
import statistics
import sys

tag = sys.argv[1]
root = "gpurun_out"


def rows(pattern):
    out = []
    for f in glob.glob(f"{root}/{tag}_{pattern}", recursive=True):
        with open(f, newline="") as fh:
            out += list(csv.DictReader(fh))
    return out


summary = {"tag": tag, "kernels": {}}
for r in rows("stats/**/*kernel_stats.csv"):
    name = r["Name"]
    if "qt::" not in name:
        continue
    summary["kernels"].setdefault(name.split("(")[0], {}).update(
        calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3, min_us=float(r["MinNs"]) / 1e3,
        max_us=float(r["MaxNs"]) / 1e3)
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    per = {}
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if r["Counter_Name"] == counter and "qt::" in r["Kernel_Name"]:
            per.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        summary["kernels"].setdefault(k, {})[counter + "_KiB_median"] = statistics.median(v)
        summary["kernels"][k][counter + "_launches"] = len(v)
print(json.dumps(summary, indent=1))
