#!/usr/bin/env python3
"""Host-to-device copy bandwidth from pinned memory on this box: one copy at a time, and the same bytes split over
2 / 4 streams (different SDMA queues), and through a kernel reading the pinned buffer directly (torch copy_ kernel)."""
import time
import torch

dev = torch.device("cuda", 0)
n = 300_000_000
src = torch.empty(n, dtype=torch.uint8).pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device=dev)
streams = [torch.cuda.Stream(dev) for _ in range(4)]
for ways in (1, 2, 4):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(4):
            step = n // ways
            for w in range(ways):
                with torch.cuda.stream(streams[w]):
                    dst[w * step:(w + 1) * step].copy_(src[w * step:(w + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("H2D pinned, %d stream(s): %.1f GB/s" % (ways, 4 * n / dt / 1e9), flush=True)
out = torch.empty(8_000_000, dtype=torch.uint8).pin_memory()
d2 = torch.empty(8_000_000, dtype=torch.uint8, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for it in range(20):
    out.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
print("D2H 8 MB: %.1f GB/s" % (20 * 8e6 / (time.perf_counter() - t0) / 1e9))
