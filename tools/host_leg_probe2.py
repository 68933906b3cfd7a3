#!/usr/bin/env python3
"""bench.py's host-buffer leg under the microscope: full DB, pins filled by DMA or by the CPU, end() inside or outside."""
import os, sys, time
import ctypes as C
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import kmer_id_amd
from kmer_id_amd import PinnedBuffer, synth
torch.cuda.set_device(0)
device = torch.device("cuda", 0)
scale = float(os.environ.get("SCALE", "1.0"))
db, parent, cum, build_s, host, dev = bench.build_db(device, scale, 30, False)
n = 2_000_000
batches = [bench.gen_reads(device, cum, parent, b * n, n) for b in range(3)]
lib = kmer_id_amd.load()
nbytes = n * 150
for fill in ("dma", "cpu"):
    pins = [PinnedBuffer(nbytes, 0) for _ in range(3)]
    outs = [PinnedBuffer(n * 4, 0) for _ in range(3)]
    for b_, pin in zip(batches, pins):
        if fill == "dma":
            kmer_id_amd._lib.check(lib.kid_dev_download(0, C.c_void_p(pin.ptr), C.c_void_p(b_.data_ptr()), nbytes))
        else:
            pin.array[:] = b_[:nbytes].cpu().numpy()
    s = db.sample()
    def run(k):
        t = []
        for i in range(k):
            t.append(s.classify_fixed_async(pins[i % 3].ptr, 150, n, outs[i % 3].ptr))
            if len(t) > 2:
                s.wait(t.pop(0))
        for x in t:
            s.wait(x)
    run(3); s.reset(); torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); run(12); t1 = time.perf_counter(); g, u = s.end(); t2 = time.perf_counter()
        print("fill=%s scale=%g: run %.1f ms/batch (%.1f GB/s), end() %.1f ms" % (fill, scale, (t1 - t0) / 12 * 1e3, 12 * nbytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3), flush=True)
    s.close()
    for p in pins + outs:
        p.close()
