#!/usr/bin/env python3
"""Development aid: classify-kernel time on reads whose every k-mer is in the DB (the lookup queue and
the resolver are the hot path then), next to reads of the same DB without any hit."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmer_id_amd import KmerDB, synth  # noqa: E402
from test_gpu_parity import _genome_db, K  # noqa: E402

read_len, n = 150, 2_000_000
parent, _ = synth.load_taxonomy("bact10")
rng = np.random.default_rng(3)
genomes, keys, targets = _genome_db(parent, 400, 20000, rng)
db = KmerDB(keys, targets, parent, k=K, log2_slots=int(os.environ.get("LOG2", "26")))
G = np.stack(genomes)
for name, frac in (("no hits", 0.0), ("25 % of the reads from the genomes", 0.25), ("all reads from the genomes", 1.0)):
    gi = rng.integers(0, G.shape[0], n); pos = rng.integers(0, G.shape[1] - read_len + 1, n)
    idx = pos[:, None] + np.arange(read_len)[None, :]
    bases = G[gi[:, None], idx]
    rnd = rng.random(n) >= frac
    bases[rnd] = rng.choice(np.frombuffer(b"ACGT", np.uint8), (int(rnd.sum()), read_len))
    d = torch.from_numpy(np.ascontiguousarray(bases).reshape(-1)).cuda()
    s = db.sample(); s.set_timing(True)
    for _ in range(6):
        s.classify_fixed_device(d.data_ptr(), read_len, n)
    ms, launches = s.kernel_time()
    st = s.stats()
    print("%-40s %.3f ms per 2 M reads, %.1f G lookups/s, hits per read %.1f, cells per lookup %.3f" % (name, ms / launches, st["lookups"] / 6 / (ms / launches) / 1e6, st["hits"] / st["reads"], st["probes"] / st["lookups"]))
    s.close()
