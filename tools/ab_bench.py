#!/usr/bin/env python3
"""A/B the classify kernel across alternative builds of the library:
   python tools/ab_bench.py kmer_id_amd/libkid_*.so  [-- extra bench.py args]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--")
    args, extra = args[:i], args[i + 1:]
for lib in args:
    env = dict(os.environ, KMER_ID_AMD_LIB=os.path.abspath(lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-reads", "0", "--gather", "0"] + extra,
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode().strip().splitlines()
    try:
        d = json.loads(out[-1]); r = d["roofline"]
        print("%-40s %7.1f Mpairs/s  kernel %.3f ms  %.2f Glookups/s  %.1f GB/s alg  cells/lookup %.3f" % (
            os.path.basename(lib), d["value"] / 1e6, r["avg_kernel_ms"], r["lookups_per_s"] / 1e9, r["achieved"],
            r["cells_read_per_launch"] / r["lookups_per_launch"]), flush=True)
    except Exception as e:
        print(lib, "FAILED", e, out[-3:])
