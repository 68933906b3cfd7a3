#!/usr/bin/env python3
"""A/B bench.py over environment variants: python tools/ab_env.py [--rounds R] "KID_GRID_MULT=4" "KID_GRID_MULT=8" ... [-- bench args]"""
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
if args and args[0] == "--rounds":
    rounds = int(args[1]); args = args[2:]
for rd in range(rounds):
    for v in args:
        env = dict(os.environ)
        for kv in v.split():
            k, _, val = kv.partition("=")
            env[k] = val
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-reads", "0", "--host-leg", "0", "--e2e-leg", "0"] + common,
                           env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out = p.stdout.decode().strip().splitlines()
        try:
            d = json.loads(out[-1]); r = d["roofline"]
            print("%-28s %7.1f Mpairs/s  kernel %.3f ms (dev clock %.3f)  frac %.3f" % (v, d["value"] / 1e6, r["avg_kernel_ms"], r.get("avg_kernel_ms_device_clock") or 0, r["frac"]), flush=True)
        except Exception as e:
            print(v, "FAILED", e, out[-2:], p.stderr.decode()[-400:], flush=True)
