#!/usr/bin/env python3
"""Taxonomy list -> target ids / tree, the way the reference's shipped DB files were derived
(the rule is inferred, the reference ships no code for it, and it is verified here
against the files it does ship): walking the rank columns left to right, every distinct path of
non-blank ranks gets the next id (from 2) at its first appearance; blank ranks are skipped; a node's
parent is the previous non-blank rank of the same row; top-rank nodes hang off root (no edge line).

  python tools/taxonomy_from_list.py            # verifies the rule on mitochondria_list.txt and writes
                                                # kmer_id_amd/data/taxonomy_fungal.npz from fung1_list_vf6.txt
                                                # (runs only where /root/reference exists; the .npz is committed)
  python tools/taxonomy_from_list.py LIST --ranks N --acc-col C --out PREFIX [--counts FILE.npy]
        # the converter proper: writes PREFIX_data.txt ("target<TAB>accession", list order: what kmer_read_vf6.cpp
        # :1059-1089 reads as <name>_data.txt), PREFIX_tree.txt ("parent<TAB>child" per edge: <name>_tree.txt) and
        # PREFIX_refkey.txt (the report scripts' key, in the layout of mitochondria_refkey.txt: header, then
        # target, name = ranks joined by '_', probe count, three simulation columns the reference's author filled
        # in by other means (0 here), number of strains under the node; CRLF like the reference's file)
"""
import os
import sys

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kmer_id_amd", "data")


def derive_named(path, n_ranks, acc_col):
    """-> parent[], [(target, accession)], names[target], strains_under[target]"""
    ids, parent, strain_targets = {}, {1: 1}, []
    names = {0: "none", 1: "root"}
    under = {}
    nxt = 2
    with open(path, encoding="latin-1") as fh:
        next(fh)
        for line in fh:
            f = line.rstrip("\r\n").split("\t")
            if len(f) <= acc_col or not f[acc_col].strip():
                continue
            path_key, prev, lineage = (), 1, []
            for r in range(n_ranks):
                name = f[r].strip()
                if not name:
                    continue
                path_key = path_key + (name,)
                if path_key not in ids:
                    ids[path_key] = nxt
                    parent[nxt] = prev
                    names[nxt] = "_".join(path_key)
                    nxt += 1
                prev = ids[path_key]
                lineage.append(prev)
            for t in lineage:
                under[t] = under.get(t, 0) + 1
            strain_targets.append((prev, f[acc_col].strip()))
    ntar = nxt
    par = np.ones(ntar, np.int32)
    for k, v in parent.items():
        par[k] = v
    return par, strain_targets, [names[i] for i in range(ntar)], [under.get(i, 0) for i in range(ntar)]


def write_db_files(list_path, n_ranks, acc_col, out_prefix, probe_counts=None):
    """The three text files of a DB (minus the probes): <prefix>_data.txt, _tree.txt, _refkey.txt."""
    par, strains, names, under = derive_named(list_path, n_ranks, acc_col)
    with open(out_prefix + "_data.txt", "w", newline="") as fh:
        for t, acc in strains:
            fh.write("%d\t%s\n" % (t, acc))
    with open(out_prefix + "_tree.txt", "w", newline="") as fh:
        edges = sorted((int(par[c]), c) for c in range(2, par.size) if par[c] != 1)  # top-rank nodes hang off root: no line
        for p_, c in edges:
            fh.write("%d\t%d\n" % (p_, c))
    with open(out_prefix + "_refkey.txt", "w", newline="") as fh:
        fh.write("target\tname\tprobe count\treads hit\treads tested\ttotal size\tstrains\r\n")
        for t in range(par.size):
            pc = int(probe_counts[t]) if probe_counts is not None and t < len(probe_counts) else 0
            fh.write("%d\t%s\t%d\t0\t0\t0\t%d\r\n" % (t, names[t], pc, under[t]))
    return par, strains, names, under


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
    if len(sys.argv) > 1:
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument("list")
        ap.add_argument("--ranks", type=int, required=True, help="number of taxonomy columns (mitochondria_list.txt: 6, fung1_list_vf6.txt: 8)")
        ap.add_argument("--acc-col", type=int, required=True, help="0-based column of the accession (mitochondria_list.txt: 6, fung1_list_vf6.txt: 8)")
        ap.add_argument("--out", required=True, help="output prefix")
        ap.add_argument("--counts", help=".npy of probe counts per target for the refkey")
        a = ap.parse_args()
        par, strains, names, under = write_db_files(a.list, a.ranks, a.acc_col, a.out, np.load(a.counts) if a.counts else None)
        print("%d nodes, %d strains -> %s_{data,tree,refkey}.txt" % (par.size, len(strains), a.out))
    else:
        main()
