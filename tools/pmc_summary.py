#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_<tag>/ (tools/pmc_passes.sh) into one JSON: per-dispatch averages
of every counter for the classify kernel + the kernel-trace average duration."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
out = {"source": root}
for d in sorted(os.listdir(root)):
    p = os.path.join(root, d)
    if not os.path.isdir(p):
        continue
    for f in glob.glob(p + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "classify" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                out.setdefault("vgpr", r.get("VGPR_Count")); out.setdefault("sgpr", r.get("SGPR_Count"))
                out.setdefault("lds_block_size", r.get("LDS_Block_Size")); out.setdefault("grid", r.get("Grid_Size"))
                out.setdefault("workgroup", r.get("Workgroup_Size"))
        for k, v in agg.items():
            out[k] = {"dispatches": len(v), "avg": sum(v) / len(v)}
    for f in glob.glob(p + "/**/*_kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "classify" in r["Name"]:
                out["kernel_trace"] = {"name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                       "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
json.dump(out, sys.stdout, indent=1)
