"""Multi-GPU merge of one sample's counters: one process per GPU, reads sharded
over the ranks, the database replicated in every GPU's HBM.

  gcount (reads per target)          additive            -> all_reduce(SUM)
  ucount (distinct DB k-mers seen)   NOT additive        -> ranks exchange slices of the
        per-cell "seen" bitmap (all_to_all: every xGMI link carries one slice), OR them,
        count their own slice by target, and the partial counts ride in the same
        all_reduce as gcount.

The exchange itself is plain torch.distributed ("nccl" = RCCL on ROCm; "gloo" on
CPU for the tests); the classification path has no collective in it.
"""
import functools
import time

import numpy as np
import torch
import torch.distributed as dist


def _tick(device):
    if torch.device(device).type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


def merge_counts(gcount, seen, count_slice, group=None, force_collectives=False, timing=None):
    """gcount: int64[ntar] tensor; seen: uint8[nbytes] tensor (this rank's bitmap, nbytes a multiple of 16), both on the
    backend's device.  count_slice(byte_begin, byte_end, merged_slice_uint8) -> int64[ntar] tensor with the ucount
    contribution of that byte range of the bitmap (byte_end <= nbytes; an empty range is never asked for).
    timing: a dict that receives the seconds of the phases (all_to_all, or, count, all_reduce) on this rank.
    Returns (gcount_total, ucount_total), identical on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    nbytes = seen.numel()
    if world == 1 and not force_collectives:
        return gcount.clone(), count_slice(0, nbytes, seen)
    if nbytes % 16 != 0:
        raise ValueError("bitmap of %d bytes is not a whole number of 16-byte groups" % nbytes)
    # equal slices of whole 16-byte groups; the bitmap is padded with zero bytes up to world x slice (one bit per DB
    # entry: its size has nothing to do with the number of ranks)
    sl = -(-(nbytes // 16) // world) * 16
    send = seen
    if sl * world != nbytes:
        send = torch.zeros(sl * world, dtype=seen.dtype, device=seen.device)
        send[:nbytes] = seen
    recv = torch.empty_like(send)
    t0 = _tick(seen.device)
    dist.all_to_all_single(recv, send, group=group)  # chunk j of recv = slice `rank` of rank j's bitmap
    t1 = _tick(seen.device)
    merged = functools.reduce(torch.bitwise_or, recv.view(world, sl).unbind(0))
    t2 = _tick(seen.device)
    b0, b1 = min(rank * sl, nbytes), min((rank + 1) * sl, nbytes)
    if b1 > b0:
        part = count_slice(b0, b1, merged[:b1 - b0].contiguous())
    else:
        part = torch.zeros_like(gcount)
    t3 = _tick(seen.device)
    both = torch.stack([gcount, part.to(gcount.device)])
    dist.all_reduce(both, op=dist.ReduceOp.SUM, group=group)
    t4 = _tick(seen.device)
    if timing is not None:
        timing.update({"all_to_all_s": t1 - t0, "or_s": t2 - t1, "count_s": t3 - t2, "all_reduce_s": t4 - t3, "slice_bytes": sl})
    return both[0], both[1]


def merge_sample(sample, device, group=None, force_collectives=False, timing=None):
    """Sample-level wrapper for the HIP path: -> (gcount, ucount) numpy int64 arrays."""
    device = torch.device(device)
    t_begin = time.perf_counter()
    on_dev = device.type == "cuda"  # False: rehearsal over gloo, the exchange goes through host memory
    g = torch.from_numpy(sample.gcount()).to(device)
    nbytes = sample.seen_bytes()
    seen = torch.empty(nbytes, dtype=torch.uint8, device=device)
    sample.seen_export(0, nbytes, dst_ptr=seen.data_ptr(), on_device=on_dev)

    def count_slice(b0, b1, merged):
        if b1 - b0 != nbytes:  # fold the other ranks' bits of my slice into my bitmap, then count it
            sample.seen_or(b0, merged.data_ptr(), nbytes=b1 - b0, on_device=on_dev)
        return torch.from_numpy(sample.ucount_range(b0 * 8, b1 * 8)).to(device)

    t_export = time.perf_counter()
    gt, ut = merge_counts(g, seen, count_slice, group, force_collectives, timing)
    if timing is not None:
        timing["export_s"] = t_export - t_begin  # waits for the rank's classify kernels, then gcount + bitmap out of the library
        timing["total_s"] = time.perf_counter() - t_begin
    return gt.cpu().numpy().astype(np.int64), ut.cpu().numpy().astype(np.int64)
