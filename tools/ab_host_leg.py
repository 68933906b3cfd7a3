#!/usr/bin/env python3
"""A/B of bench.py's host-buffer leg (pinned host memory -> kid_classify_fixed_async) over libraries and environments:
   python tools/ab_host_leg.py "lib.so" "lib.so ENV=1" ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rd in range(2):
    for spec in sys.argv[1:]:
        parts = spec.split()
        env = dict(os.environ, KMER_ID_AMD_LIB=os.path.abspath(parts[0]))
        for kv in parts[1:]:
            k, _, v = kv.partition("="); env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-reads", "0", "--gather", "0", "--xcheck", "0", "--e2e-leg", "0", "--steps", "10", "--warmup", "3"],
                             env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip().splitlines()
        d = json.loads(out[-1]); h = d["host_buffer_path"]
        print("%-60s host path %.1f M pairs/s (%.1f GB/s h2d)  resident %.1f M pairs/s" % (spec[-60:], h["pairs_per_s"] / 1e6, h["GBps_h2d"], d["value"] / 1e6), flush=True)
