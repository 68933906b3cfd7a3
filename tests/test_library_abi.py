"""The C-ABI library loads and exports every symbol include/kmer_id_amd.h (the boundary) and include/kmer_id_amd_bench.h
(bench / test helpers) declare;
host-only helpers work; compute entry points fail loudly without a GPU."""
import os
import re

import numpy as np
import pytest

import kmer_id_amd
from kmer_id_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("kmer_id_amd.h", "kmer_id_amd_bench.h"))
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kid_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = kmer_id_amd.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.PROTOTYPES, "no ctypes prototype for " + n
    assert sorted(_lib.PROTOTYPES) == names


def test_no_cpu_fallback_without_device():
    if kmer_id_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    parent, cnt = synth.load_taxonomy("bact10")
    with pytest.raises(kmer_id_amd.KidError) as e:
        kmer_id_amd.KmerDB(np.array([1], np.uint64), np.array([2], np.uint32), parent, log2_slots=10)
    assert e.value.status == -6  # KID_ERR_NO_DEVICE


def test_synth_generators_are_deterministic_and_canonical():
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, 1e-4))
    k1, t1 = synth.db_keys(cum)
    k2, t2 = synth.db_keys(cum, j0=100, n=50)
    assert np.array_equal(k1[100:150], k2) and np.array_equal(t1[100:150], t2)
    assert np.all(np.diff(t1.astype(np.int64)) >= 0)  # target order, like the builder's file
    # canonical: key <= reverse complement
    def rc(v):
        r = 0
        for i in range(30):
            r = (r << 2) | (3 - ((v >> (2 * i)) & 3))
        return r
    for v in k1[:200].tolist():
        assert v <= rc(v)
    a = synth.reads(cum, parent, 64, 150)
    b = synth.reads(cum, parent, 32, 150, r0=32)
    assert np.array_equal(a[32 * 150:], b)
    assert set(np.unique(a).tolist()) <= set(b"ACGTacgtN")
