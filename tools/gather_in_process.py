#!/usr/bin/env python3
"""Random-line gather rates on the REAL table inside the bench process (kid_bench_gather) -- next to tools/gather_shape.hip,
which asks the same of a table of its own in a process of its own."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402

device = torch.device("cuda", 0)
db, parent, cum, _, _, _ = bench.build_db(device, 1.0, int(os.environ.get("LOG2_SLOTS", 30)), False)
for name, code, n in (("16-byte cells, 1 in flight", 1, 1 << 29), ("16-byte cells, 4 in flight", 4, 1 << 29),
                      ("lines, 64 per load, 4 in flight (asm)", 101, 1 << 29), ("lines, runs of 8, 4 in flight (asm)", 108, 1 << 29),
                      ("lines, 64 per load, a random cell (asm)", 111, 1 << 29), ("lines, 64 per load, cell 0, compiler's load", 121, 1 << 29),
                      ("lines, 64 per load, random cell, compiler's load", 131, 1 << 29)):
    for rep in range(2):
        ms, loads = db.gather_ceiling(n_loads=n, inflight=code, iters=3)
        print("%-44s %8.3f ms  %7.2f G lines/s" % (name, ms, loads / ms / 1e6), flush=True)
