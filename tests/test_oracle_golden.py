"""The oracle (oracle/kmer_oracle.c, our plain-C restatement) against the golden
vectors the COMPILED REFERENCE produced (oracle/make_golden.py).  CPU only."""
import filecmp
import gzip
import hashlib
import json
import os
import shutil

import numpy as np
import pytest

from helpers import K, ob, synth, unpack_strings


@pytest.fixture(scope="module")
def kat_db(kat, tmp_path_factory):
    """oracle DB loaded through its own probes-file parser from the KAT probes text"""
    d = tmp_path_factory.mktemp("katdb")
    p = os.path.join(d, "probes.txt.gz")
    with open(p, "wb") as fh:
        fh.write(bytes(kat["probes_gz"]))
    parent, _ = synth.load_taxonomy("bact10")
    db = ob.OracleDB(int(kat["ntar"]), K, int(kat["log2_slots"]), parent=parent)
    n = db.load_probes_gz(p)
    assert n > 2500
    return db


def test_fmix64(kat):
    lib = ob.load()
    got = np.array([lib.ko_fmix64(int(x)) for x in kat["fmix_in"]], np.uint64)
    assert np.array_equal(got, kat["fmix_out"])


def test_gethash_first_wins_and_quirks(kat, kat_db):
    got = kat_db.get(kat["lookup_in"]).astype(np.int64)
    assert np.array_equal(got, kat["lookup_out"])


def test_msca_pairs(kat, kat_db):
    got = np.array([kat_db.msca(x, y) for x, y in zip(kat["msca_x"].tolist(), kat["msca_y"].tolist())])
    assert np.array_equal(got, kat["msca_out"])


def test_msca_all_pairs_checksum(kat, kat_db):
    assert int(kat["msca_all_ntar"]) == kat_db.ntar
    assert kat_db.lib.ko_msca_checksum(kat_db.h) == int(kat["msca_all_sum"])


def test_process_qual(kat):
    seqs = unpack_strings(kat["qual_seq_data"], kat["qual_seq_off"])
    quals = unpack_strings(kat["qual_qual_data"], kat["qual_qual_off"])
    exp = kat["qual_out"]
    for i, (s, q) in enumerate(zip(seqs, quals)):
        called, st, sp = ob.process_qual(q, len(s), K)
        assert called == exp[i, 0], i
        if called:
            assert (st, sp) == (exp[i, 1], exp[i, 2]), i


def test_process_qual_then_read(kat, kat_db):
    seqs = unpack_strings(kat["qual_seq_data"], kat["qual_seq_off"])
    quals = unpack_strings(kat["qual_qual_data"], kat["qual_qual_off"])
    exp = kat["qual_out"]
    s = ob.OracleSample(kat_db)
    lib = kat_db.lib
    for i, (sq, q) in enumerate(zip(seqs, quals)):
        called, st, sp = ob.process_qual(q, len(sq), K)
        if called == 1:
            f = lib.ko_process_read(s.h, sq, st, sp, None)
            assert f == exp[i, 3], i


def test_process_read_finals_and_counts(kat, kat_db):
    reads = unpack_strings(kat["reads_data"], kat["reads_off"])
    s = ob.OracleSample(kat_db)
    lib = kat_db.lib
    got = np.array([lib.ko_process_read(s.h, r, 0, len(r) - 1, None) for r in reads])
    assert np.array_equal(got, kat["reads_final"])
    g, u = s.counts()
    rc = kat["reads_counts"]
    eg = np.zeros_like(g); eu = np.zeros_like(u)
    eg[rc[:, 0]] = rc[:, 1]; eu[rc[:, 0]] = rc[:, 2]
    assert np.array_equal(g, eg)
    assert np.array_equal(u, eu)


def _setup_bact10_dir(tmp, scale):
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, scale))
    keys, targets = synth.db_keys(cum, K)
    db = ob.OracleDB(parent.size, K, 22, parent=parent)
    p = os.path.join(tmp, "probes10.txt.gz")
    synth.write_probes_gz(p, keys, targets, K)
    assert db.load_probes_gz(p) == keys.size
    return db, parent, cum


def test_e2e_small_files(gold_dir, tmp_path):
    """whole-program parity: FASTQ.gz in -> _result.txt/_reads.txt byte-identical to nk10_ref_small"""
    src = os.path.join(gold_dir, "e2e_small")
    params = json.load(open(os.path.join(src, "params.json")))
    db, _, _ = _setup_bact10_dir(str(tmp_path), params["scale"])
    fq = os.path.join(tmp_path, "fq")
    os.makedirs(fq)
    for f in os.listdir(src):
        if f.endswith(".fastq.gz"):
            shutil.copy(os.path.join(src, f), fq)
    s = ob.OracleSample(db)
    for prefix in ("S1", "S2"):
        assert s.run_sample(fq + "/", prefix) == 0
        for suffix in ("_result.txt", "_reads.txt"):
            assert filecmp.cmp(os.path.join(fq, prefix + suffix), os.path.join(src, prefix + suffix), shallow=False), prefix + suffix


def test_e2e_seeded(gold_dir, tmp_path):
    params = json.load(open(os.path.join(gold_dir, "e2e_seeded.json")))
    db, parent, cum = _setup_bact10_dir(str(tmp_path), params["scale"])
    assert int(cum[-1]) == params["n_keys"]
    fq = os.path.join(tmp_path, "fq")
    os.makedirs(fq)
    n, L = params["n_pairs"], params["read_len"]
    synth.write_fastq_gz(os.path.join(fq, "big_R1_tr.fastq.gz"), synth.reads(cum, parent, n, L, K, r0=0), synth.qualities(n, L, r0=0), L, mate=1)
    synth.write_fastq_gz(os.path.join(fq, "big_R2_tr.fastq.gz"), synth.reads(cum, parent, n, L, K, r0=n), synth.qualities(n, L, r0=n), L, mate=2)
    s = ob.OracleSample(db)
    assert s.run_sample(fq + "/", "big") == 0
    res = open(os.path.join(fq, "big_result.txt"), "rb").read()
    assert res == gzip.open(os.path.join(gold_dir, "e2e_seeded_result.txt.gz")).read()
    assert hashlib.sha256(res).hexdigest() == params["result_sha256"]
    assert hashlib.sha256(open(os.path.join(fq, "big_reads.txt"), "rb").read()).hexdigest() == params["reads_sha256"]


def test_e2e_unmodified_reference_2pow30(gold_dir):
    """oracle/time_reference.py: the reference program as shipped (2^30 cells, 24 GiB) on 200 000 seeded pairs; the
    oracle (here on all host cores: same counters as sequentially, oracle_selftest.c) must reproduce its _result.txt"""
    meta = json.load(open(os.path.join(gold_dir, "e2e_ref_full.json")))
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, meta["scale"]))
    keys, targets = synth.db_keys(cum, K)
    assert keys.size == meta["n_keys"]
    n, L = meta["n_pairs"], meta["read_len"]
    bases = np.concatenate([synth.reads(cum, parent, n, L, K, r0=0), synth.reads(cum, parent, n, L, K, r0=n)])
    db = ob.OracleDB(parent.size, K, 22, parent=parent)
    db.add(keys, targets)
    _, g, u, st = db.classify_mt(bases, synth.fixed_offsets(2 * n, L), max(1, len(os.sched_getaffinity(0))))
    res = "".join("%d,%d,%d\n" % (i, g[i], u[i]) for i in range(parent.size)).encode()
    assert st["lookups"] == meta["lookups"]
    assert hashlib.sha256(res).hexdigest() == meta["result_sha256"]
    assert res == gzip.open(os.path.join(gold_dir, "e2e_ref_full_result.txt.gz")).read()
