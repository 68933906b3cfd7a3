#!/usr/bin/env python3
"""A database shaped like the output of the reference's builder (kmer_build_vf6.cpp:460-640), next to a random-key
database of the same size: how the minimizer-localised table copes with CLUMPS.

The builder walks every genome and emits non-overlapping 30-mers (the next probe ends at least 31 bases after the last one,
:626) whose target -- the LCA of the genomes that contain the k-mer (:168-193) -- is more specific than the root.  Related
genomes are walked independently, so around conserved sequence the probes of different strains sit at DIFFERENT PHASES:
overlapping k-mers that share their minimizer, up to 15 keys on a line that holds 7.

Synthetic stand-in (the GenBank genomes are not in the repo): G genera x S species x T strains on the bact10 tree shape
(here: its own three-level tree), species = genus ancestor + 4 % substitutions, strain = species + 0.4 % substitutions +
a few short indels (they shift the walk's phase).  k-mer -> LCA target over all genomes, emission per genome as above
(minimum count rule :596-603 simplified to count >= 1, no entropy filter, <= 100 000 probes per target :41).

Reports, for the clumped DB and for random keys (same number of entries, same table size): kernel ms per 1 M pairs,
lookups/s, table cells read per lookup, lines with 7 entries and a chain, hits per read; for reads drawn from the
genomes (dense hits), and for random reads (no hits).
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmer_id_amd import KmerDB  # noqa: E402

K = 30
G, S, T = int(os.environ.get("G", 24)), int(os.environ.get("S", 5)), int(os.environ.get("T", 5))
GLEN = int(os.environ.get("GLEN", 200_000))
rng = np.random.default_rng(11)


def mutate(g, sub, n_indel):
    g = g.copy()
    m = rng.random(g.size) < sub
    g[m] = (g[m] + rng.integers(1, 4, int(m.sum()))) & 3
    for _ in range(n_indel):
        p = int(rng.integers(1000, g.size - 1000))
        if rng.random() < 0.5:
            g = np.concatenate([g[:p], rng.integers(0, 4, int(rng.integers(1, 4))).astype(np.uint8), g[p:]])
        else:
            g = np.concatenate([g[:p], g[p + int(rng.integers(1, 4)):]])
    return g


def kmers(codes):
    """canonical 2-bit keys of every 30-mer of a genome given as codes 0..3"""
    n = codes.size - K + 1
    f = np.zeros(n, np.uint64)
    r = np.zeros(n, np.uint64)
    for j in range(K):
        c = codes[j:j + n].astype(np.uint64)
        f = (f << np.uint64(2)) | c
        r = r | ((np.uint64(3) - c) << np.uint64(2 * j))
    return np.minimum(f, r)


CACHE = os.environ.get("CLUMPED_CACHE")  # (several library builds on the same database: the 3 minutes of numpy above once)
if CACHE and os.path.exists(CACHE):
    z = np.load(CACHE)
    keys, targets, parent, reads_g, reads_r, rkeys = z["keys"], z["targets"], z["parent"], z["reads_g"], z["reads_r"], z["rkeys"]
    n, L, n_reads = keys.size, 150, reads_g.shape[0]
    log2_slots = max(16, int(np.ceil(np.log2(n / 0.10))))
    dev = torch.device("cuda", 0)
    print("clumped DB from %s: %d probes" % (CACHE, n), flush=True)
else:
    t0 = time.time()
    # taxonomy: 1 = root, then genera, species, strains
    parent = [1, 1]
    genomes, strain_t, species_t, genus_t = [], [], [], []
    for gi in range(G):
        parent.append(1); gt = len(parent) - 1
        anc = rng.integers(0, 4, GLEN).astype(np.uint8)
        for si in range(S):
            parent.append(gt); st = len(parent) - 1
            sp = mutate(anc, 0.04, 0)
            for ti in range(T):
                parent.append(st); tt = len(parent) - 1
                genomes.append(mutate(sp, 0.004, 6))
                strain_t.append(tt); species_t.append(st); genus_t.append(gt)
    parent = np.array(parent, np.int32)
    ntar = parent.size
    # k-mer -> LCA
    keys_all = [kmers(g) for g in genomes]
    occ_key = np.concatenate(keys_all)
    occ_gen = np.concatenate([np.full(k_.size, i, np.int32) for i, k_ in enumerate(keys_all)])
    order = np.argsort(occ_key, kind="stable")
    sk = occ_key[order]
    sg = occ_gen[order]
    first = np.concatenate([[True], sk[1:] != sk[:-1]])
    starts = np.flatnonzero(first)
    ukeys = sk[starts]
    st_arr, sp_arr, ge_arr = np.array(strain_t)[sg], np.array(species_t)[sg], np.array(genus_t)[sg]


    def same(a):
        return np.minimum.reduceat(a, starts) == np.maximum.reduceat(a, starts)


    lca = np.where(same(st_arr), np.minimum.reduceat(st_arr, starts),
                   np.where(same(sp_arr), np.minimum.reduceat(sp_arr, starts),
                            np.where(same(ge_arr), np.minimum.reduceat(ge_arr, starts), 1))).astype(np.uint32)
    del occ_key, occ_gen, order, sk, sg, st_arr, sp_arr, ge_arr
    # emission: per genome, greedy non-overlapping walk over the positions whose k-mer has a target > 1
    pcount = np.zeros(ntar, np.int64)
    out_keys, out_targets = [], []
    for gi, kk in enumerate(keys_all):
        tg = lca[np.searchsorted(ukeys, kk)]
        elig = np.flatnonzero(tg > 1)
        p = 0
        pos = []
        while True:
            j = np.searchsorted(elig, p)
            if j >= elig.size:
                break
            q = int(elig[j])
            if pcount[tg[q]] < 100000:
                pos.append(q)
                pcount[tg[q]] += 1
                p = q + K + 1  # the next probe ENDS at least KSIZE + 1 later: minpos = gpos + KSIZE, test gpos > minpos
            else:
                p = q + 1
        pos = np.array(pos, np.int64)
        out_keys.append(kk[pos]); out_targets.append(tg[pos])
    keys = np.concatenate(out_keys)
    targets = np.concatenate(out_targets)
    n = keys.size
    nu = np.unique(keys).size
    print("clumped DB: %d genomes x %d bases, %d nodes, %d probes (%d distinct keys) in %.0f s" % (len(genomes), GLEN, ntar, n, nu, time.time() - t0), flush=True)
    log2_slots = max(16, int(np.ceil(np.log2(n / 0.10))))
    rkeys = rng.integers(0, 1 << 60, n, dtype=np.uint64)  # random keys (canonical or not: lookups of random reads never hit either way)
    dev = torch.device("cuda", 0)
    n_reads, L = 2_000_000, 150
    Gm = [g for g in genomes]
    gi = rng.integers(0, len(Gm), n_reads)
    reads_g = np.empty((n_reads, L), np.uint8)
    lut = np.frombuffer(b"ACGT", np.uint8)
    for i in range(len(Gm)):
        m = np.flatnonzero(gi == i)
        pos = rng.integers(0, Gm[i].size - L + 1, m.size)
        reads_g[m] = lut[Gm[i][pos[:, None] + np.arange(L)[None, :]]]
    reads_r = lut[rng.integers(0, 4, (n_reads, L))].astype(np.uint8)
    if CACHE:
        np.savez(CACHE, keys=keys, targets=targets, parent=parent, reads_g=reads_g, reads_r=reads_r, rkeys=rkeys)
for name, kk in (("clumped (builder-shaped)", keys), ("random keys", rkeys)):
    db = KmerDB(kk, targets, parent, k=K, log2_slots=log2_slots)
    info = db.info
    t_, p_ = db.lookup(kk[:: max(1, n // 200000)], with_probes=True)
    print("%s: %d entries, 2^%d cells (load %.3f), geometry %d; cells per lookup of a present key %.3f (max %d)" % (
        name, n, log2_slots, n / 2 ** log2_slots, info.geometry, p_.mean(), p_.max()), flush=True)
    for rname, rd in (("reads from the genomes", reads_g), ("random reads", reads_r)):
        d = torch.from_numpy(np.ascontiguousarray(rd).reshape(-1)).to(dev)
        s = db.sample(); s.set_timing(True)
        for _ in range(6):
            s.classify_fixed_device(d.data_ptr(), L, n_reads)
        ms, launches = s.kernel_time()
        st = s.stats()
        print("   %-24s %.3f ms per 1 M pairs, %6.1f G lookups/s, cells per lookup %.4f, hits per read %.2f" % (
            rname, ms / launches, st["lookups"] / 6 / (ms / launches) / 1e6, st["probes"] / st["lookups"], st["hits"] / st["reads"]), flush=True)
        s.close()
        del d
    db.close()
