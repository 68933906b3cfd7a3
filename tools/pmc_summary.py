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
# what the profile is tied to (bench.py refuses a profile of other kernel sources)
import hashlib
_repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _rel in ("kmer_id_amd/csrc/kid_kernels.hip.h", "kmer_id_amd/csrc/kid_api.hip", "kmer_id_amd/csrc/kid_common.h"):
    _h.update(open(os.path.join(_repo, _rel), "rb").read())
out["kernel_source_sha256_16"] = _h.hexdigest()[:16]
out["commit"] = os.environ.get("KID_COMMIT")  # (the GPU box has no .git: tools/pmc_traffic.sh is given the commit)
for d in sorted(os.listdir(root)):
    p = os.path.join(root, d)
    if not os.path.isdir(p):
        continue
    # two classify kernels run per batch (pair loop / general loops) and one of them returns at once:
    # the numbers are those of the kernel that did the work (largest values), the other is only named
    # rocprofv3 writes one CSV per PROCESS it saw: the one with the most classify dispatches is the bench process (a
    # stray child -- the nk10 leg -- would otherwise overwrite its numbers)
    best = None
    for f in glob.glob(p + "/**/*_counter_collection.csv", recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        meta = {}
        n_rows = 0
        for r in csv.DictReader(open(f)):
            if "classify" in r["Kernel_Name"]:
                per[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[r["Kernel_Name"]] = r
                n_rows += 1
        if per and (best is None or n_rows > best[0]):
            best = (n_rows, per, meta)
    if best:
        _, per, meta = best
        # several classify kernels may run per batch and all but one return at once: the numbers are those of the
        # kernel that did the work (largest values)
        name = max(per, key=lambda k: sum(sum(v) for v in per[k].values()))
        r = meta[name]
        out.setdefault("kernel", name)
        out.setdefault("vgpr", r.get("VGPR_Count")); out.setdefault("sgpr", r.get("SGPR_Count"))
        out.setdefault("lds_block_size", r.get("LDS_Block_Size")); out.setdefault("grid", r.get("Grid_Size"))
        out.setdefault("workgroup", r.get("Workgroup_Size"))
        for k, v in per[name].items():
            out[k] = {"dispatches": len(v), "avg": sum(v) / len(v)}
    for f in glob.glob(p + "/**/*_kernel_stats.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "classify" in r["Name"]]
        rows.sort(key=lambda r: -float(r["AverageNs"]))
        for j, r in enumerate(rows):
            out["kernel_trace" if j == 0 else "kernel_trace_idle_twin"] = {
                "name": r["Name"], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])}
json.dump(out, sys.stdout, indent=1)
