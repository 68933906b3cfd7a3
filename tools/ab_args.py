#!/usr/bin/env python3
"""A/B bench.py over argument variants, interleaved, inside one process tree (one GPU box):
   python tools/ab_args.py [--rounds R] [--lib path.so] "--log2-slots 30" "--log2-slots 29" ... [-- common bench.py args]
Each variant string is split on whitespace and appended to the bench.py command line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
common = []
if "--" in args:
    i = args.index("--")
    args, common = args[:i], args[i + 1:]
rounds = 2
lib = None
while args and args[0] in ("--rounds", "--lib"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    else:
        lib = os.path.abspath(args[1])
    args = args[2:]
env = dict(os.environ)
if lib:
    env["KMER_ID_AMD_LIB"] = lib
for rd in range(rounds):
    for v in args:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-reads", "0"] + common + v.split()
        p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out = p.stdout.decode().strip().splitlines()
        try:
            d = json.loads(out[-1]); r = d["roofline"]
            g = " ".join("%s=%s" % (k[9:], r[k]) for k in sorted(r) if k.startswith("gather16B") and k.endswith("per_s"))
            print("%-34s %7.1f Mpairs/s  kernel %.3f ms (dev clock %.3f)  %.1f Glookups/s  frac %.3f  cells/lookup %.4f  parity %s  %s" % (
                v, d["value"] / 1e6, r["avg_kernel_ms"], r.get("avg_kernel_ms_device_clock") or 0, r["lookups_per_s"] / 1e9, r["frac"],
                r["cells_read_per_launch"] / r["lookups_per_launch"], r.get("fullsize_parity_vs_reference_geometry"), g), flush=True)
        except Exception as e:
            print(v, "FAILED", e, out[-3:], p.stderr.decode()[-600:], flush=True)
