#!/usr/bin/env python3
"""Where does the host-buffer pipeline lose time?  Variants of bench.py's host leg on a small DB (the kernels' own time
hardly depends on it): copies only / no result download / everything."""
import os
import sys
import time
import ctypes as C
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import kmer_id_amd
from kmer_id_amd import KmerDB, PinnedBuffer, synth

parent, cnt = synth.load_taxonomy("bact10")
cum = synth.cumulative(synth.scaled_counts(cnt, 0.01))
keys, targets = synth.db_keys(cum)
db = KmerDB(keys, targets, parent, k=30, log2_slots=24)
n, L = 2_000_000, 150
pins = [PinnedBuffer(n * L) for _ in range(3)]
host = synth.reads(cum, parent, n, L)
for p in pins:
    p.array[:] = host
outs = [PinnedBuffer(n * 4) for _ in range(3)]
s = db.sample()
mode = os.environ.get("MODE", "all")
print(open("/proc/self/status").read().split("Cpus_allowed_list:")[1].split()[0], flush=True)
def run(k):
    t = []
    for i in range(k):
        t.append(s.classify_fixed_async(pins[i % 3].ptr, L, n, outs[i % 3].ptr if mode != "noout" else 0))
        if len(t) > 2:
            s.wait(t.pop(0))
    for x in t:
        s.wait(x)
run(3)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    run(12)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%s KID_DEBUG_NO_KERNELS=%s: %.1f ms per batch, %.1f GB/s, %.1f M pairs/s" % (
        mode, os.environ.get("KID_DEBUG_NO_KERNELS"), dt / 12 * 1e3, 12 * n * L / dt / 1e9, 12 * n / 2 / dt / 1e6), flush=True)
