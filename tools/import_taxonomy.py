#!/usr/bin/env python3
"""Derive the compact taxonomy fixtures the synthetic workloads need from the
reference's DATA files (not code): the parent array after all add_edge calls
(newkmer_10nx.cpp:973-983) and the per-target k-mer counts (refkey column 3).

Runs only where /root/reference exists (the build container); the outputs
kmer_id_amd/data/taxonomy_*.npz are committed.
"""
import os
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmer_id_amd", "data")


def load_parent(tree_path, ntar):
    parent = np.ones(ntar, np.int32)
    for line in open(tree_path):
        f = line.split()
        if len(f) >= 2:
            parent[int(f[1])] = int(f[0])
    return parent


def load_counts(refkey_path, ntar, col=2):
    cnt = np.zeros(ntar, np.int64)
    with open(refkey_path, encoding="latin-1") as fh:
        next(fh)
        for line in fh:
            f = line.rstrip("\r\n").split("\t")
            if len(f) > col and f[0].isdigit() and int(f[0]) < ntar:
                cnt[int(f[0])] = int(f[col])
    return cnt


def main():
    os.makedirs(OUT, exist_ok=True)
    # bact10: MAXTAR = 5982 (newkmer_10nx.cpp:45)
    parent = load_parent(os.path.join(REF, "b10", "btree_10.txt"), 5982)
    cnt = load_counts(os.path.join(REF, "b10", "refkey10.txt"), 5982)
    np.savez_compressed(os.path.join(OUT, "taxonomy_bact10.npz"), parent=parent, kmer_count=cnt)
    print("bact10: ntar", parent.size, "kmers", int(cnt.sum()))
    # mitochondria: num_targ = max target + 1 over mitochondria_data.txt (kmer_read_m3.cpp:1036-1044)
    tmax = 0
    for line in open(os.path.join(REF, "mitochondria_data.txt")):
        f = line.split()
        if len(f) >= 2:
            tmax = max(tmax, int(f[0]))
    ntar = tmax + 1
    parent = load_parent(os.path.join(REF, "mitochondria_tree.txt"), ntar)
    cnt = load_counts(os.path.join(REF, "mitochondria_refkey.txt"), ntar)
    np.savez_compressed(os.path.join(OUT, "taxonomy_mito.npz"), parent=parent, kmer_count=cnt)
    print("mito: ntar", ntar, "probes", int(cnt.sum()))


if __name__ == "__main__":
    main()
