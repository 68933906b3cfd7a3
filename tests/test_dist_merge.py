"""world_size-2 gloo test of the multi-GPU merge logic (host side, CPU tensors)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _popcount_by_target(bitmap_u8, byte0, values, ntar):
    bits = np.unpackbits(bitmap_u8, bitorder="little")
    slots = np.flatnonzero(bits) + byte0 * 8
    return np.bincount(values[slots], minlength=ntar).astype(np.int64)


def _worker(rank, world, port, ntar, nbytes, seed, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from kmer_id_amd.dist import merge_counts
    rng = np.random.default_rng(seed)
    values = rng.integers(2, ntar, nbytes * 8)                   # slot -> target (same on all ranks)
    r2 = np.random.default_rng(seed + 1 + rank)
    g = r2.integers(0, 1000, ntar).astype(np.int64)
    seen = np.packbits(r2.random(nbytes * 8) < 0.01, bitorder="little")

    def count_slice(b0, b1, merged):
        return torch.from_numpy(_popcount_by_target(merged.numpy(), b0, values, ntar))

    gt, ut = merge_counts(torch.from_numpy(g), torch.from_numpy(seen), count_slice)
    q.put((rank, g, seen, gt.numpy().copy(), ut.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,nbytes", [(2, 4096), (4, 4096), (2, 4096 + 16), (4, 16 * 13), (3, 16 * 2)])
def test_merge_counts_gloo(world, nbytes):
    """(the bitmap has one bit per DB entry: its size need not divide by the number of ranks, and may be smaller than it)"""
    ntar, seed = 97, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ntar, nbytes, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    res.sort(key=lambda t: t[0])
    values = np.random.default_rng(seed).integers(2, ntar, nbytes * 8)
    g_sum = sum(r[1] for r in res)
    seen_or = np.bitwise_or.reduce(np.stack([r[2] for r in res]))
    u_exp = _popcount_by_target(seen_or, 0, values, ntar)
    naive = sum(_popcount_by_target(r[2], 0, values, ntar).sum() for r in res)
    assert u_exp.sum() <= naive and (nbytes < 4096 or u_exp.sum() < naive)  # ucount is not additive
    for r in res:
        assert np.array_equal(r[3], g_sum)
        assert np.array_equal(r[4], u_exp)
