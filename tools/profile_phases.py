#!/usr/bin/env python3
"""Development aid: where a wave of the classify kernel spends its cycles.  Needs a library built with
   -DKID_PROFILE -DKID_PAIRS=0 (the sequential read loop with s_memtime stamps around its phases):
   KMER_ID_AMD_LIB=$PWD/kmer_id_amd/libkid_prof.so python tools/profile_phases.py [read_len]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import kmer_id_amd  # noqa: E402

bench.READ_LEN = int(sys.argv[1]) if len(sys.argv) > 1 else 150
device = torch.device("cuda", 0)
db, parent, cum, _, _, _ = bench.build_db(device, 1.0, 30, False)
n_reads = 2_000_000
reads = bench.gen_reads(device, cum, parent, 0, n_reads)
out_final = torch.empty(n_reads, dtype=torch.int32, device=device)
s = db.sample()
iters = 3
for it in range(iters):
    s.classify_fixed_device(reads.data_ptr(), bench.READ_LEN, n_reads, d_out=out_final.data_ptr())
torch.cuda.synchronize()
lib = kmer_id_amd.load()
out = (ctypes.c_uint64 * 24)()
lib.kid_sample_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
lib.kid_sample_debug_counters.restype = ctypes.c_int
assert lib.kid_sample_debug_counters(s._h, out) == 0
pairs = os.environ.get("PAIRS", "0") == "1"
if pairs:  # -DKID_PROFILE build with the pair kernel
    names = ["wait words A", "stage+front A", "wait words B", "stage+front B", "wait headers A", "back+finish A", "words A' + wait headers B",
             "back+finish B + words B'", "tail"]
else:      # -DKID_PROFILE -DKID_PAIRS=0
    names = ["between reads", "wait words", "stage+front", "wait headers", "back half", "finish+prefetch", "", "", "tail"]
v = list(out)[:9]
tot = float(sum(v))
for n, x in zip(names, v):
    print("%-16s %6.2f %%  %8.0f ticks/read" % (n, 100.0 * x / tot, x / (iters * n_reads)))
print("resolver (contained in the phases above) %6.2f %%" % (100.0 * out[9] / tot))
print("total ticks/read %.0f (s_memtime ticks; 100 MHz constant clock on gfx9+)" % (tot / (iters * n_reads)))
