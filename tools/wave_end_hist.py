#!/usr/bin/env python3
"""Development aid (library built with -DKID_ENDHIST): histogram of the times at which the waves of one launch run out of
reads (64 us bins from the launch's first workgroup start).
   KMER_ID_AMD_LIB=$PWD/kmer_id_amd/libkid_eh.so python tools/wave_end_hist.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kmer_id_amd
device = torch.device("cuda", 0)
db, parent, cum, _, _, _ = bench.build_db(device, 1.0, 30, False)
n_reads = int(os.environ.get("READS", 2_000_000))
batches = [bench.gen_reads(device, cum, parent, b * n_reads, n_reads) for b in range(2)]
out_final = torch.empty(n_reads, dtype=torch.int32, device=device)
s = db.sample()
lib = kmer_id_amd.load()
lib.kid_sample_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
lib.kid_sample_debug_counters.restype = ctypes.c_int
for it in range(6):
    s.classify_fixed_device(batches[it % 2].data_ptr(), 150, n_reads, d_out=out_final.data_ptr())
torch.cuda.synchronize()
s.reset()
reps = 4
for it in range(reps):
    s.classify_fixed_device(batches[it % 2].data_ptr(), 150, n_reads, d_out=out_final.data_ptr())
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 24)()
assert lib.kid_sample_debug_counters(s._h, out) == 0
v = list(out)
ms, nl = s.kernel_time_device()
print("kernel %.1f us (device clock, mean of %d launches); waves ending per 64-us bin (mean per launch):" % (ms * 1e3 / nl, nl))
tot = sum(v)
acc = 0
for i, x in enumerate(v):
    acc += x
    if x:
        print("  %4d-%4d us: %7.1f waves  (cumulative %5.1f %%)" % (64 * i, 64 * i + 64, x / reps, 100.0 * acc / tot))
import numpy as np
rec = np.zeros((8192, 4), np.uint32)
lib.kid_sample_debug_wave_records.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64]
lib.kid_sample_debug_wave_records.restype = ctypes.c_int
if lib.kid_sample_debug_wave_records(s._h, rec.ctypes.data_as(ctypes.c_void_p), 8192) == 0:
    reads, tend, tlast, nb = rec[:, 0].astype(float), rec[:, 1] / 100.0, rec[:, 2] / 100.0, rec[:, 3]
    print("per wave (last launch): reads min/mean/max %d/%.1f/%d; blocks mean %.1f" % (reads.min(), reads.mean(), reads.max(), nb.mean()))
    order = np.argsort(tend)
    for q in (0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 1.0):
        w = order[min(int(q * 8191), 8191)]
        print("  quantile %.2f: wave %5d (wib %d, block %4d)  end %.0f us, last block started %.0f us, reads %d, blocks %d" % (
            q, w, w % 8, w // 8, tend[w], tlast[w], reads[w], nb[w]))
    wib = np.arange(8192) % 8
    for k in range(8):
        m = wib == k
        print("  wave-in-block %d: mean end %.0f us, mean reads %.1f" % (k, tend[m].mean(), reads[m].mean()))
    blk = np.arange(8192) // 8
    slot = blk // 256
    for k in range(4):
        m = slot == k
        print("  blocks %4d-%4d: mean end %.0f us, mean reads %.1f" % (k * 256, k * 256 + 255, tend[m].mean(), reads[m].mean()))
    xcd = blk % 8
    for k in range(8):
        m = xcd == k
        print("  blockIdx %% 8 = %d: mean end %.0f us, mean reads %.1f" % (k, tend[m].mean(), reads[m].mean()))
    last_dur = tend - tlast
    print("  duration of the last block: mean %.0f us, p90 %.0f us, max %.0f us" % (last_dur.mean(), np.percentile(last_dur, 90), last_dur.max()))
