"""Synthetic workloads: the real probes10.txt.gz is not
distributed with the reference, so every configuration runs on a seeded
synthetic DB laid over the reference's REAL taxonomy (tree + per-target k-mer
counts) and on seeded synthetic reads.  The generators themselves live in the
native library (kid_synth_*), one implementation for host and device.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
DB_SEED = 0xB10
READ_SEED = 0x5EED


def load_taxonomy(name="bact10"):
    """-> (parent int32[ntar], kmer_count int64[ntar]) of the reference's DB `name`."""
    z = np.load(os.path.join(DATA, "taxonomy_%s.npz" % name))
    return z["parent"].astype(np.int32), z["kmer_count"].astype(np.int64)


def scaled_counts(kmer_count, scale):
    """Shrink the per-target counts for small test DBs: every target that has k-mers keeps at least one."""
    if scale >= 1.0:
        return kmer_count.copy()
    c = np.floor(kmer_count * scale).astype(np.int64)
    c[(kmer_count > 0) & (c == 0)] = 1
    return c


def cumulative(kmer_count):
    cum = np.zeros(kmer_count.size + 1, np.uint64)
    cum[1:] = np.cumsum(kmer_count.astype(np.uint64))
    return cum


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def db_keys(cum, k=30, seed=DB_SEED, j0=0, n=None):
    """Host copy of the synthetic DB entries [j0, j0+n) in probes-file order."""
    lib = _lib.load()
    ntar = cum.size - 1
    if n is None:
        n = int(cum[-1]) - j0
    keys = np.empty(n, np.uint64)
    targets = np.empty(n, np.uint32)
    check(lib.kid_synth_db_keys_host(seed, k, _p(cum), ntar, j0, n, _p(keys), _p(targets)))
    return keys, targets


def reads(cum, parent, n_reads, read_len=150, k=30, db_seed=DB_SEED, read_seed=READ_SEED, r0=0):
    """Host copy of synthetic reads [r0, r0+n_reads): uint8[n_reads*read_len] (fixed-length layout)."""
    lib = _lib.load()
    out = np.empty(n_reads * read_len, np.uint8)
    parent = np.ascontiguousarray(parent, np.int32)
    check(lib.kid_synth_reads_host(db_seed, read_seed, k, _p(cum), _p(parent), parent.size, r0, n_reads, read_len, _p(out)))
    return out


def fixed_offsets(n_reads, read_len):
    return (np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(read_len)).astype(np.uint64)


def key_to_seq(key, k=30):
    return "".join("ACGT"[(int(key) >> (2 * (k - 1 - i))) & 3] for i in range(k))


def write_probes_gz(path, keys, targets, k=30):
    """The probes text format of kmer_build_vf6.cpp:625: SEQ,target,org,position,strand,count"""
    import gzip
    with gzip.open(path, "wt", compresslevel=6, newline="") as fh:
        for j, (key, t) in enumerate(zip(keys.tolist(), targets.tolist())):
            fh.write("%s,%d,%d,%d,F,1\n" % (key_to_seq(key, k), t, 0, j))


# ---------------------------------------------------------------- FASTQ text for the file-based runs
def _splitmix64_np(x):
    x = np.asarray(x, np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def qualities(n_reads, read_len, seed=0x9A1, r0=0):
    """Mixed PHRED+33 profile, uint8[n_reads, read_len]: mostly 'I'; ~30 % of reads
    have a decaying tail, ~10 % a poor head, ~5 % are noisy throughout, ~2 % are bad
    everywhere (process_qual drops those)."""
    r = np.arange(r0, r0 + n_reads, dtype=np.uint64)
    d = _splitmix64_np(np.uint64(seed) ^ (r * np.uint64(0xA24BAED4963EE407)))
    q = np.full((n_reads, read_len), ord("I"), np.uint8)
    pos = np.arange(read_len)[None, :]
    kind = (d & np.uint64(0xFF)).astype(np.int64)
    tail = ((d >> np.uint64(8)) % np.uint64(max(read_len // 2, 1))).astype(np.int64)[:, None]
    head = ((d >> np.uint64(24)) % np.uint64(max(read_len // 4, 1))).astype(np.int64)[:, None]
    noise = _splitmix64_np(d[:, None] + np.arange(read_len, dtype=np.uint64)[None, :])
    lowq = (ord("#") + (noise % np.uint64(16))).astype(np.uint8)          # '#'..'2'  (< '1' mostly)
    midq = (ord("+") + (noise % np.uint64(30))).astype(np.uint8)          # '+'..'H'
    has_tail = (kind < 77)[:, None]
    has_head = ((kind >= 77) & (kind < 103))[:, None]
    noisy = ((kind >= 103) & (kind < 116))[:, None]
    bad = ((kind >= 116) & (kind < 121))[:, None]
    q = np.where(has_tail & (pos >= read_len - tail), lowq, q)
    q = np.where(has_head & (pos < head), lowq, q)
    q = np.where(noisy, midq, q)
    q = np.where(bad, lowq, q)
    return q


def write_fastq_gz(path, bases, quals, read_len, names_prefix="@r", mate=1, eol="\n", final_newline=True):
    """bases: uint8[n*read_len]; quals: uint8[n, read_len]."""
    import gzip
    n = bases.size // read_len
    b = bases.reshape(n, read_len)
    chunks = []
    for i in range(n):
        chunks.append("%s%d/%d" % (names_prefix, i, mate))
        chunks.append(b[i].tobytes().decode("latin-1"))
        chunks.append("+")
        chunks.append(quals[i].tobytes().decode("latin-1"))
    text = eol.join(chunks) + (eol if final_newline else "")
    with gzip.open(path, "wb", compresslevel=6) as fh:
        fh.write(text.encode("latin-1"))
