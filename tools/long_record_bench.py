#!/usr/bin/env python3
"""One very long record (a FASTA contig classified whole, kmer_read_vf6.cpp:803-861): the general loops give a record to
ONE wave, segment by segment.  Time for records of 0.1 .. 8 Mb, alone in their batch, on the bact10-synth DB (1 % scale)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmer_id_amd import KmerDB, synth
parent, cnt = synth.load_taxonomy("bact10")
cum = synth.cumulative(synth.scaled_counts(cnt, 0.01))
keys, targets = synth.db_keys(cum)
db = KmerDB(keys, targets, parent, k=30, log2_slots=24)
rng = np.random.default_rng(5)
for n_rec, L in ((1, 100_000), (1, 1_000_000), (1, 8_000_000), (8, 1_000_000), (64, 1_000_000), (512, 125_000)):
    bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_rec * L)]
    # a DB k-mer every ~2 kb so that the fold has something to do
    for p in range(1000, n_rec * L - 40, 2000):
        v = int(keys[rng.integers(0, keys.size)])
        bases[p:p + 30] = np.frombuffer("".join("ACGT"[(v >> (2 * (29 - i))) & 3] for i in range(30)).encode(), np.uint8)
    off = (np.arange(n_rec + 1, dtype=np.uint64) * np.uint64(L))
    s = db.sample()
    s.classify(bases, off)  # warm
    torch.cuda.synchronize()
    s.reset(); s.set_timing(True)
    for _ in range(3):
        s.classify(bases, off)
    ms, nl = s.kernel_time()
    st = s.stats()
    print("%4d record(s) of %9d bases: kernel %.2f ms per batch, %.2f G lookups/s" % (n_rec, L, ms / nl, st["lookups"] / 3 / (ms / nl) / 1e6), flush=True)
    s.close()
