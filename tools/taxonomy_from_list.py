#!/usr/bin/env python3
"""Taxonomy list -> target ids / tree, the way the reference's shipped DB files were derived
(the rule is inferred, the reference ships no code for it, and it is verified here
against the files it does ship): walking the rank columns left to right, every distinct path of
non-blank ranks gets the next id (from 2) at its first appearance; blank ranks are skipped; a node's
parent is the previous non-blank rank of the same row; top-rank nodes hang off root (no edge line).

  python tools/taxonomy_from_list.py            # verifies the rule on mitochondria_list.txt and writes
                                                # kmer_id_amd/data/taxonomy_fungal.npz from fung1_list_vf6.txt
Runs only where /root/reference exists; the .npz it writes is committed.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmer_id_amd", "data")


def derive(path, n_ranks, acc_col):
    ids, parent, strain_targets = {}, {1: 1}, []
    nxt = 2
    with open(path, encoding="latin-1") as fh:
        next(fh)
        for line in fh:
            f = line.rstrip("\r\n").split("\t")
            if len(f) <= acc_col or not f[acc_col].strip():
                continue
            path_key, prev = (), 1
            for r in range(n_ranks):
                name = f[r].strip()
                if not name:
                    continue
                path_key = path_key + (name,)
                if path_key not in ids:
                    ids[path_key] = nxt
                    parent[nxt] = prev
                    nxt += 1
                prev = ids[path_key]
            strain_targets.append((prev, f[acc_col].strip()))
    ntar = nxt
    par = np.ones(ntar, np.int32)
    for k, v in parent.items():
        par[k] = v
    return par, strain_targets


def main():
    # --- verify on the mitochondria files
    par, strains = derive(os.path.join(REF, "mitochondria_list.txt"), 6, 6)
    z = np.load(os.path.join(OUT, "taxonomy_mito.npz"))
    assert par.size == z["parent"].size, (par.size, z["parent"].size)
    assert np.array_equal(par, z["parent"]), "tree mismatch on mitochondria_list.txt"
    data = [l.split() for l in open(os.path.join(REF, "mitochondria_data.txt")) if l.strip()]
    assert [(int(a), b) for a, b in data] == strains, "strain targets mismatch on mitochondria_list.txt"
    print("rule verified on mitochondria_list.txt: %d nodes, %d strains" % (par.size, len(strains)))
    # --- fungal (vf6) taxonomy: only the list ships
    par, strains = derive(os.path.join(REF, "fung1_list_vf6.txt"), 8, 8)
    depth = np.zeros(par.size, np.int32)
    for i in range(2, par.size):
        d, z_ = 0, i
        while z_ != 1:
            z_ = par[z_]; d += 1
        depth[i] = d
    # synthetic k-mer counts: strains' own targets get 2000 each, inner nodes 300 ("~10 k probes / target, synthetic")
    cnt = np.zeros(par.size, np.int64)
    leaf = np.zeros(par.size, bool)
    for t, _ in strains:
        leaf[t] = True
    cnt[2:] = 300
    cnt[leaf] = 2000
    np.savez_compressed(os.path.join(OUT, "taxonomy_fungal.npz"), parent=par, kmer_count=cnt,
                        strain_target=np.array([t for t, _ in strains], np.int32))
    print("fungal: %d nodes, %d strains, depth max %d, synthetic k-mers %d" % (par.size, len(strains), depth.max(), cnt.sum()))


if __name__ == "__main__":
    main()
