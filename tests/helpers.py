"""Shared helpers for the test-suite (imports the oracle: test infrastructure)."""
import gzip
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from kmer_id_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

K = 30


def unpack_strings(data, off):
    raw = bytes(data)
    off = [int(x) for x in off]
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def parse_probes_text(text, k=K):
    """Host restatement (numpy-free, tiny inputs) of process_kmergz + process_kmer for tests that
    need the (key, target) list in file order; the C oracle has its own parser, this one is
    only used to hand the SAME entries to kid_db_build."""
    keys, targets = [], []
    parts = text.split(b"\n")
    lines = parts[:-1]  # the unterminated tail is dropped (newkmer_10nx.cpp:705-709)
    for line in lines:
        if line.endswith(b"\r"):
            line = line[:-1]
        if not line:
            continue
        f = line.replace(b",", b" ").split()
        if len(f) < 6:
            continue
        try:
            t = int(f[1]); int(f[2]); int(f[3]); int(f[5])
        except ValueError:
            continue
        if t < 0:
            continue
        cpos, key = 0, 0
        for ch in f[0]:
            c = {65: 0, 67: 1, 71: 2, 84: 3}.get(ch, -1)
            if c < 0:
                cpos, key = 0, 0
            else:
                key = ((key << 2) & ((1 << (2 * k)) - 1)) | c
                cpos += 1
            if cpos == k:
                keys.append(key)
                targets.append(t)
                cpos -= 1
    return np.array(keys, np.uint64), np.array(targets, np.uint32)


def small_db(scale, name="bact10", k=K, seed=synth.DB_SEED):
    parent, cnt = synth.load_taxonomy(name)
    cum = synth.cumulative(synth.scaled_counts(cnt, scale))
    keys, targets = synth.db_keys(cum, k, seed=seed)
    return parent, cum, keys, targets


def oracle_db(parent, keys, targets, log2_slots, k=K, max_probes=0, flags=0):
    db = ob.OracleDB(parent.size, k, log2_slots, max_probes, flags, parent=parent)
    db.add(keys, targets)
    return db


def concat_reads(seqs):
    data = np.frombuffer(b"".join(seqs), np.uint8).copy()
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return data, off
