#!/usr/bin/env python3
"""Development aid (library built with -DKID_WAVEPROF): how long the waves of a 2 M-read launch take for their 1st, 2nd,
3rd 64 reads and the rest, and how far apart they finish.
   KMER_ID_AMD_LIB=$PWD/kmer_id_amd/libkid_wprof.so python tools/wave_progress.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import kmer_id_amd
device = torch.device("cuda", 0)
db, parent, cum, _, _, _ = bench.build_db(device, 1.0, 30, False)
n_reads = int(os.environ.get("READS", 2_000_000))
reads = bench.gen_reads(device, cum, parent, 0, n_reads)
out_final = torch.empty(n_reads, dtype=torch.int32, device=device)
s = db.sample()
lib = kmer_id_amd.load()
lib.kid_sample_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
lib.kid_sample_debug_counters.restype = ctypes.c_int
for it in range(3):
    s.classify_fixed_device(reads.data_ptr(), 150, n_reads, d_out=out_final.data_ptr())
torch.cuda.synchronize()
s.reset()
import ctypes as C
# stats[14] is a minimum: arm it
s.classify_fixed_device(reads.data_ptr(), 150, n_reads, d_out=out_final.data_ptr())
torch.cuda.synchronize()
out = (ctypes.c_uint64 * 24)()
assert lib.kid_sample_debug_counters(s._h, out) == 0
v = list(out)
nw = 8192
tick = 0.01  # us per tick (100 MHz)
print("reads per wave %.1f" % (n_reads / nw))
for q in range(4):
    print("block %d (reads %d..%d of a wave): mean %.1f us" % (q, 64 * q, 64 * q + 63, v[q] / nw * tick))
print("rest: mean %.1f us" % (v[4] / nw * tick))
print("mean wave loop time %.1f us; first wave through at %.1f us, last at %.1f us (kernel clock from the first workgroup's start)" % (
    v[5] / nw * tick, v[6] * tick if v[6] < 2**60 else -1, v[7] * tick))
print("mean wave start delay %.1f us, last start %.1f us" % (v[8] / nw * tick, v[9] * tick))
mean_end = None
ms, md = s.kernel_time_device()
print("device-clock kernel time %.1f us over %d launch(es)" % (ms * 1e3 / max(md, 1), md))
