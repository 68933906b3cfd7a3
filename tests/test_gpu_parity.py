"""Parity of the HIP path (through the C ABI) with the oracle and with the golden
vectors of the compiled reference.  Bit-exact: everything here is integer work."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

import kmer_id_amd
from kmer_id_amd import KID_FLAG_HOST_BUILD, KID_FLAG_REF_GEOMETRY, KID_FLAG_U_IS_T, KmerDB, synth
from helpers import K, concat_reads, ob, oracle_db, parse_probes_text, small_db, unpack_strings

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kat_entries(kat):
    text = gzip.decompress(bytes(kat["probes_gz"]))
    keys, targets = parse_probes_text(text)
    parent, _ = synth.load_taxonomy("bact10")
    return parent, keys, targets


GEOMETRIES = {"minimizer_localised": 0, "gpu_reference_geometry": KID_FLAG_REF_GEOMETRY, "host_sequential": KID_FLAG_HOST_BUILD}


@pytest.fixture(scope="module", params=list(GEOMETRIES))
def kat_db(request, kat, kat_entries):
    parent, keys, targets = kat_entries
    flags = GEOMETRIES[request.param]
    db = KmerDB(keys, targets, parent, k=K, log2_slots=int(kat["log2_slots"]), flags=flags)
    assert db.info.host_built == (1 if flags == KID_FLAG_HOST_BUILD else 0)
    assert db.info.geometry == (1 if flags == 0 else 0)
    yield db
    db.close()


def test_gethash_golden(kat, kat_db):
    got = kat_db.lookup(kat["lookup_in"]).astype(np.int64)
    assert np.array_equal(got, kat["lookup_out"])


def test_host_built_table_has_reference_geometry(kat, kat_entries):
    """same probe counts as the oracle's table => same cell for every entry"""
    parent, keys, targets = kat_entries
    db = KmerDB(keys, targets, parent, k=K, log2_slots=12, flags=KID_FLAG_HOST_BUILD)
    odb = oracle_db(parent, keys, targets, 12)
    t, p = db.lookup(kat["lookup_in"], with_probes=True)
    et, ep = odb.get(kat["lookup_in"], with_probes=True)
    assert np.array_equal(t, et) and np.array_equal(p, ep)
    assert db.info.n_occupied == int((targets != 0).sum())  # the (z0,0) entry left its cell empty; (z0,5) took it


def test_msca_golden_pairs(kat, kat_db):
    got = kat_db.msca(kat["msca_x"], kat["msca_y"])
    assert np.array_equal(got, kat["msca_out"])


def _mix(k):
    k = k.astype(np.uint64)
    with np.errstate(over="ignore"):
        k ^= k >> np.uint64(30); k *= np.uint64(0xbf58476d1ce4e5b9)
        k ^= k >> np.uint64(27); k *= np.uint64(0x94d049bb133111eb)
        k ^= k >> np.uint64(31)
    return k


def test_msca_all_pairs_checksum(kat, kat_db):
    ntar = int(kat["msca_all_ntar"])
    ys = np.arange(1, ntar, dtype=np.int32)
    total = np.uint64(0)
    rows = 256
    with np.errstate(over="ignore"):
        for x0 in range(1, ntar, rows):
            xs = np.arange(x0, min(ntar, x0 + rows), dtype=np.int32)
            X = np.repeat(xs, ys.size)
            Y = np.tile(ys, xs.size)
            m = kat_db.msca(X, Y).astype(np.uint64)
            w = _mix(X.astype(np.uint64) * np.uint64(ntar) + Y.astype(np.uint64))
            total += (w * m).sum(dtype=np.uint64)
    assert int(total) == int(kat["msca_all_sum"])


def test_trim_golden(kat, kat_db):
    quals = unpack_strings(kat["qual_qual_data"], kat["qual_qual_off"])
    seqs = unpack_strings(kat["qual_seq_data"], kat["qual_seq_off"])
    # the reference trims over seq.length(); quals may be longer: cut them to the read length
    q = [qq[:len(s)] for qq, s in zip(quals, seqs)]
    data, off = concat_reads(q)
    start, stop, keep = kat_db.trim(data, off)
    exp = kat["qual_out"]
    assert np.array_equal(keep.astype(np.int64), exp[:, 0])
    called = exp[:, 0] == 1
    assert np.array_equal(start[called], exp[called, 1])
    assert np.array_equal(stop[called], exp[called, 2])


def test_trim_then_classify_golden(kat, kat_db):
    quals = unpack_strings(kat["qual_qual_data"], kat["qual_qual_off"])
    seqs = unpack_strings(kat["qual_seq_data"], kat["qual_seq_off"])
    qd, off = concat_reads([qq[:len(s)] for qq, s in zip(quals, seqs)])
    sd, off2 = concat_reads(seqs)
    assert np.array_equal(off, off2)
    start, stop, keep = kat_db.trim(qd, off)
    s = kat_db.sample()
    final = s.classify(sd, off, start, stop)
    exp = kat["qual_out"]
    called = exp[:, 0] == 1
    assert np.array_equal(final[called].astype(np.int64), exp[called, 3])
    s.close()


def test_process_read_golden(kat, kat_db):
    reads = unpack_strings(kat["reads_data"], kat["reads_off"])
    data, off = concat_reads(reads)
    s = kat_db.sample()
    final = s.classify(data, off)
    assert np.array_equal(final.astype(np.int64), kat["reads_final"])
    g, u = s.end()
    rc = kat["reads_counts"]
    eg = np.zeros_like(g); eu = np.zeros_like(u)
    eg[rc[:, 0]] = rc[:, 1]; eu[rc[:, 0]] = rc[:, 2]
    assert np.array_equal(g, eg)
    assert np.array_equal(u, eu)
    s.close()


# ------------------------------------------------------------------ seeded runs against the oracle
@pytest.fixture(scope="module", params=["minimizer_localised", "gpu_reference_geometry"])
def seeded(request):
    parent, cum, keys, targets = small_db(1e-3)
    odb = oracle_db(parent, keys, targets, 20)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=20, flags=GEOMETRIES[request.param])
    yield parent, cum, keys, targets, odb, db
    db.close()


@pytest.mark.parametrize("read_len", [150, 250, 31, 30, 29])
def test_seeded_reads_vs_oracle(seeded, read_len):
    parent, cum, keys, targets, odb, db = seeded
    n = 20000
    bases = synth.reads(cum, parent, n, read_len, K)
    off = synth.fixed_offsets(n, read_len)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    eg, eu = os_.counts()
    s = db.sample()
    got = s.classify(bases, off)
    assert np.array_equal(got, exp)
    g, u = s.end()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    st, est = s.stats(), os_.stats()
    assert st["reads"] == n and st["lookups"] == est["lookups"] and st["hits"] == est["hits"]
    if read_len >= 150:
        assert st["hits"] > 0 and (exp > 1).sum() > n // 4
    s.close()


def test_probe_counts_match_on_reference_geometry(seeded):
    parent, cum, keys, targets, odb, _ = seeded
    db = KmerDB(keys, targets, parent, k=K, log2_slots=20, flags=KID_FLAG_HOST_BUILD)
    n = 5000
    bases = synth.reads(cum, parent, n, 150, K, r0=777)
    off = synth.fixed_offsets(n, 150)
    os_ = ob.OracleSample(odb)
    os_.classify(bases, off)
    s = db.sample()
    s.classify(bases, off, want_final=False)
    assert s.stats()["probes"] == os_.stats()["probes"]
    s.close(); db.close()


def test_batching_does_not_change_results(seeded):
    parent, cum, keys, targets, odb, db = seeded
    n, L = 9000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=123)
    off = synth.fixed_offsets(n, L)
    s1 = db.sample()
    f1 = s1.classify(bases, off)
    g1, u1 = s1.end()
    s2 = db.sample()
    parts = []
    for a, b in ((0, 1), (1, 4000), (4000, 4001), (4001, n)):
        parts.append(s2.classify(bases, off[a:b + 1]))  # absolute offsets into the same buffer
    s2.classify(bases, off[0:1])  # empty batch
    g2, u2 = s2.end()
    assert np.array_equal(np.concatenate(parts), f1)
    assert np.array_equal(g1, g2) and np.array_equal(u1, u2)
    # reset really resets
    s2.reset()
    g3, u3 = s2.end()
    assert g3.sum() == 0 and u3.sum() == 0
    s1.close(); s2.close()


def test_device_resident_fixed_layout(seeded):
    parent, cum, keys, targets, odb, db = seeded
    lib = kmer_id_amd.load()
    n, L = 7000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=55)
    d_b, d_o = C.c_void_p(), C.c_void_p()
    _lib = kmer_id_amd._lib
    _lib.check(lib.kid_dev_alloc(0, bases.size + 32, C.byref(d_b)))
    _lib.check(lib.kid_dev_alloc(0, n * 4, C.byref(d_o)))
    _lib.check(lib.kid_dev_upload(0, d_b, bases.ctypes.data_as(C.c_void_p), bases.size))
    s = db.sample()
    s.classify_fixed_device(d_b.value, L, n, d_out=d_o.value)
    _lib.check(lib.kid_dev_sync(0))
    got = np.empty(n, np.uint32)
    _lib.check(lib.kid_dev_download(0, got.ctypes.data_as(C.c_void_p), d_o, n * 4))
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, synth.fixed_offsets(n, L))
    assert np.array_equal(got, exp)
    g, u = s.end()
    eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    lib.kid_dev_free(0, d_b); lib.kid_dev_free(0, d_o)
    s.close()


def test_ragged_long_and_adversarial_reads(seeded):
    parent, cum, keys, targets, odb, db = seeded
    rng = np.random.default_rng(5)
    kseq = [synth.key_to_seq(int(x)).encode() for x in keys[rng.integers(0, keys.size, 64)]]
    reads = [b"", b"A", b"ACGT" * 7, b"N" * 200, b"acgtn" * 50]
    for L in (959, 960, 961, 989, 990, 991, 1919, 1920, 1949, 1950, 1951, 5000, 16383):
        s = bytearray(rng.choice(list(b"ACGT"), L).tolist())
        # hits straddling the 960-k-mer segment boundaries and the read ends
        for p in (0, 930, 931, 945, 959, 960, 961, 975, 1890, 1919, 1920, L - 30):
            if 0 <= p <= L - 30:
                s[p:p + 30] = kseq[(p + L) % 64]
        if L > 1000:
            s[500] = ord("N")
        reads.append(bytes(s))
    for i in range(300):
        L = int(rng.integers(1, 400))
        s = bytearray(rng.choice(list(b"ACGTacgtNRY-"), L, p=[.2, .2, .2, .2, .04, .04, .04, .04, .01, .01, .01, .01]).tolist())
        if L >= 30 and i % 2:
            p = int(rng.integers(0, L - 29))
            s[p:p + 30] = kseq[i % 64]
        reads.append(bytes(s))
    data, off = concat_reads(reads)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(data, off)
    s = db.sample()
    got = s.classify(data, off)
    assert np.array_equal(got, exp)
    g, u = s.end()
    eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    assert s.stats()["lookups"] == os_.stats()["lookups"]
    s.close()


def test_start_stop_outside_read_is_an_error(seeded):
    parent, cum, keys, targets, odb, db = seeded
    bases = synth.reads(cum, parent, 4, 150, K)
    off = synth.fixed_offsets(4, 150)
    s = db.sample()
    with pytest.raises(kmer_id_amd.KidError):
        s.classify(bases, off, np.array([0, 0, 0, 0], np.int32), np.array([149, 150, 149, 149], np.int32))
    s.close()


# ------------------------------------------------------------------ variants of the sibling programs
def test_probe_cap_m3(seeded):
    """kmer_read_m3.cpp:232: lookups give up after 16 probes; geometry must be the sequential one"""
    parent, cum, keys, targets, _, _ = seeded
    n = 3600  # 2^12 cells at load 0.88: many keys sit deeper than 16 probes
    odb = oracle_db(parent, keys[:n], targets[:n], 12, max_probes=16)
    odb0 = oracle_db(parent, keys[:n], targets[:n], 12, max_probes=0)
    db = KmerDB(keys[:n], targets[:n], parent, k=K, log2_slots=12, max_probes=16)
    assert db.info.host_built == 1
    q = np.concatenate([keys[:n], keys[n:n + 2000]])
    t, p = db.lookup(q, with_probes=True)
    et, ep = odb.get(q, with_probes=True)
    assert np.array_equal(t, et) and np.array_equal(p, ep)
    assert (et[:n] == 0).sum() > 0 and p.max() == 16
    assert not np.array_equal(et, odb0.get(q))
    bases = synth.reads(synth.cumulative(np.bincount(targets[:n], minlength=parent.size)), parent, 3000, 150, K)
    off = synth.fixed_offsets(3000, 150)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    s = db.sample()
    assert np.array_equal(s.classify(bases, off), exp)
    g, u = s.end(); eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    s.close(); db.close()


def test_u_is_t_flag_vf6(seeded):
    parent, cum, keys, targets, _, _ = seeded
    n = 2000
    bases = synth.reads(cum, parent, n, 150, K, r0=999).copy()
    tpos = np.flatnonzero(bases == ord("T"))
    bases[tpos[::3]] = ord("U")
    tpos = np.flatnonzero(bases == ord("t"))
    bases[tpos[::2]] = ord("u")
    off = synth.fixed_offsets(n, 150)
    for flags in (0, KID_FLAG_U_IS_T):
        odb = oracle_db(parent, keys, targets, 20, flags=flags)
        db = KmerDB(keys, targets, parent, k=K, log2_slots=20, flags=flags)
        os_ = ob.OracleSample(odb)
        exp = os_.classify(bases, off)
        s = db.sample()
        assert np.array_equal(s.classify(bases, off), exp)
        g, u = s.end(); eg, eu = os_.counts()
        assert np.array_equal(g, eg) and np.array_equal(u, eu)
        if flags:
            assert (exp > 0).sum() > n // 4
        s.close(); db.close()


def test_mito_taxonomy_large_ntar_path():
    """17227 targets: gcount leaves the LDS histogram for the run-length + global-atomic path"""
    parent, cnt = synth.load_taxonomy("mito")
    cum = synth.cumulative(synth.scaled_counts(cnt, 2e-3))
    keys, targets = synth.db_keys(cum, K, seed=0x317)
    odb = oracle_db(parent, keys, targets, 18)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=18)
    assert db.info.tree_depth == 6
    n = 10000
    bases = synth.reads(cum, parent, n, 150, K, db_seed=0x317)
    off = synth.fixed_offsets(n, 150)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    s = db.sample()
    assert np.array_equal(s.classify(bases, off), exp)
    g, u = s.end(); eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    s.close(); db.close()


def test_deep_tree_uses_climb_fallback():
    """a 40-level chain with side branches does not fit the 8-entry ancestor rows"""
    ntar = 400
    parent = np.ones(ntar, np.int32)
    for i in range(2, 42):
        parent[i] = i - 1 if i > 2 else 1      # chain 2 <- 3 <- ... <- 41
    for i in range(42, ntar):
        parent[i] = 2 + (i * 7) % 40           # side branches hanging off the chain
    cnt = np.zeros(ntar, np.int64); cnt[2:] = 5
    cum = synth.cumulative(cnt)
    keys, targets = synth.db_keys(cum, K, seed=9)
    odb = oracle_db(parent, keys, targets, 14)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=14)
    assert db.info.tree_depth > 8
    rng = np.random.default_rng(1)
    x = rng.integers(1, ntar, 20000).astype(np.int32); y = rng.integers(1, ntar, 20000).astype(np.int32)
    exp = np.array([odb.msca(a, b) for a, b in zip(x.tolist(), y.tolist())], np.int32)
    assert np.array_equal(db.msca(x, y), exp)
    n = 4000
    bases = synth.reads(cum, parent, n, 150, K, db_seed=9)
    off = synth.fixed_offsets(n, 150)
    os_ = ob.OracleSample(odb)
    e = os_.classify(bases, off)
    s = db.sample()
    assert np.array_equal(s.classify(bases, off), e)
    g, u = s.end(); eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    s.close(); db.close()


def test_depth8_tree_row_encoding():
    """fungal-style taxonomy: 8 ranks below root (kmer_read_vf6 DBs)"""
    ntar = 3000
    rng = np.random.default_rng(3)
    parent = np.ones(ntar, np.int32)
    depth = np.zeros(ntar, np.int64)
    for i in range(2, ntar):
        p = int(rng.integers(1, i)) if i > 2 else 1
        while depth[p] >= 8:
            p = int(parent[p])
        parent[i] = p
        depth[i] = depth[p] + 1 if p != 1 else 1
    assert depth.max() == 8
    cnt = np.zeros(ntar, np.int64); cnt[2:] = 2
    cum = synth.cumulative(cnt)
    keys, targets = synth.db_keys(cum, K, seed=11)
    odb = oracle_db(parent, keys, targets, 15)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=15)
    assert db.info.tree_depth == 8
    deep = np.flatnonzero(depth >= 7).astype(np.int32)
    x = np.concatenate([rng.integers(1, ntar, 30000).astype(np.int32), np.repeat(deep, 8)[:20000]])
    y = np.concatenate([rng.integers(1, ntar, 30000).astype(np.int32), np.tile(deep, 8)[:20000]])
    n = min(x.size, y.size); x, y = x[:n], y[:n]
    exp = np.array([odb.msca(a, b) for a, b in zip(x.tolist(), y.tolist())], np.int32)
    assert np.array_equal(db.msca(x, y), exp)
    nr = 6000
    bases = synth.reads(cum, parent, nr, 250, K, db_seed=11)
    off = synth.fixed_offsets(nr, 250)
    os_ = ob.OracleSample(odb)
    e = os_.classify(bases, off)
    s = db.sample()
    assert np.array_equal(s.classify(bases, off), e)
    g, u = s.end(); eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    s.close(); db.close()


def test_duplicate_keys_first_wins_on_gpu_build():
    parent, cum, keys, targets = small_db(2e-4)
    rng = np.random.default_rng(8)
    # every key three times with different targets, shuffled; ordinal order decides
    k3 = np.concatenate([keys, keys, keys])
    t3 = np.concatenate([targets, (targets + 3) % 5000 + 2, (targets + 11) % 5000 + 2]).astype(np.uint32)
    perm = rng.permutation(k3.size)
    k3, t3 = k3[perm], t3[perm]
    t3[::17] = 0  # target-0 entries are invisible
    odb = oracle_db(parent, k3, t3, 18)
    exp = odb.get(keys)
    for flags in (0, KID_FLAG_REF_GEOMETRY):
        db = KmerDB(k3, t3, parent, k=K, log2_slots=18, flags=flags)
        assert db.info.host_built == 0
        assert np.array_equal(db.lookup(keys), exp)
        db.close()


def test_crowded_table_and_heavy_minimizers():
    """load 0.9 and thousands of k-mers sharing one minimizer (tandem-repeat style keys): long
    linear-probe runs in the minimizer-localised table, results unchanged"""
    parent, cum, keys, targets = small_db(2e-4)
    rng = np.random.default_rng(21)
    # 3000 keys that all contain the same 15-mer at varying offsets -> few distinct minimizers
    core = int(rng.integers(0, 1 << 30))
    fam = []
    for i in range(3000):
        off = int(rng.integers(0, 16))
        left = int(rng.integers(0, 1 << (2 * off))) if off else 0
        right_bits = 2 * (15 - off)
        right = int(rng.integers(0, 1 << right_bits)) if right_bits else 0
        fam.append((left << (30 + right_bits)) | (core << right_bits) | right)
    fam = np.array(fam, np.uint64)
    kk = np.concatenate([keys[:11000], fam])
    tt = np.concatenate([targets[:11000], (np.arange(3000) % 5000 + 2).astype(np.uint32)])
    assert kk.size > 0.85 * (1 << 14)
    odb = oracle_db(parent, kk, tt, 14)
    q = np.concatenate([kk, fam ^ np.uint64(1), keys[11000:14000]])
    exp = odb.get(q)
    # (flags, log2_slots): the bucketed lines need <= 80 % load, so 2^14 falls back to the reference
    # placement; at 2^15 the family of 3000 keys chains through hundreds of full lines
    for flags, l2, geo in ((0, 14, 0), (KID_FLAG_REF_GEOMETRY, 14, 0), (0, 15, 1)):
        db = KmerDB(kk, tt, parent, k=K, log2_slots=l2, flags=flags)
        assert db.info.geometry == geo
        assert np.array_equal(db.lookup(q), exp)
        # reads made of the family k-mers back to back
        reads = []
        for i in range(400):
            s = b"".join(synth.key_to_seq(int(fam[(i * 7 + j) % 3000])).encode() for j in range(5))
            reads.append(s)
        data, off = concat_reads(reads)
        os_ = ob.OracleSample(odb)
        e = os_.classify(data, off)
        smp = db.sample()
        assert np.array_equal(smp.classify(data, off), e)
        g, u = smp.end(); eg, eu = os_.counts()
        assert np.array_equal(g, eg) and np.array_equal(u, eu)
        smp.close(); db.close()


def test_table_full_and_bad_inputs():
    parent, cum, keys, targets = small_db(2e-4)
    with pytest.raises(kmer_id_amd.KidError) as e:
        KmerDB(keys[:1000], targets[:1000], parent, log2_slots=10)  # 1000 > 1024 - 32
    assert e.value.status == -4
    bad = parent.copy(); bad[10] = 11; bad[11] = 10
    with pytest.raises(kmer_id_amd.KidError) as e:
        KmerDB(keys[:10], targets[:10], bad, log2_slots=10)
    assert e.value.status == -5
    with pytest.raises(kmer_id_amd.KidError) as e:
        KmerDB(keys[:10], np.full(10, 6000, np.uint32), parent, log2_slots=10)
    assert e.value.status == -7


def test_seen_bitmap_export_or_and_ranges(seeded):
    """the multi-GPU merge primitives on one GPU: two shards of a sample, merged by hand, must equal
    the unsharded sample (ucount is not additive, the OR of the seen bitmaps is)"""
    parent, cum, keys, targets, odb, db = seeded
    n, L = 12000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=4242)
    off = synth.fixed_offsets(n, L)
    whole = db.sample(); whole.classify(bases, off, want_final=False)
    g, u = whole.end()
    a = db.sample(); a.classify(bases, off[:n // 2 + 1], want_final=False)
    b = db.sample(); b.classify(bases, off[n // 2:], want_final=False)
    ga, ua = a.end(); gb, ub = b.end()
    assert np.array_equal(ga + gb, g)
    assert (ua + ub).sum() > u.sum()          # double counting if one simply adds
    nbytes = a.seen_bytes()
    assert nbytes == b.seen_bytes() and nbytes % 16 == 0
    half = (nbytes // 2) & ~15
    # rank 0 owns [0, half), rank 1 owns [half, nbytes): OR the peer's slice in, count the own slice
    a.seen_or(0, b.seen_export(0, half))
    b.seen_or(half, a.seen_export(half, nbytes - half))
    u0 = a.ucount_range(0, half * 8)
    u1 = b.ucount_range(half * 8, nbytes * 8)
    assert np.array_equal(u0 + u1, u)
    for s_ in (whole, a, b):
        s_.close()


def test_e2e_unmodified_reference_2pow30_on_gpu(gold_dir):
    """the _result.txt of the reference program as shipped (2^30 cells; oracle/time_reference.py) on 200 000 seeded pairs"""
    import gzip
    import hashlib
    import json
    meta = json.load(open(os.path.join(gold_dir, "e2e_ref_full.json")))
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, meta["scale"]))
    keys, targets = synth.db_keys(cum, K)
    n, L = meta["n_pairs"], meta["read_len"]
    db = KmerDB(keys, targets, parent, k=K, log2_slots=24)
    s = db.sample()
    for r0 in (0, n):  # R1 then R2, two batches
        s.classify(synth.reads(cum, parent, n, L, K, r0=r0), synth.fixed_offsets(n, L), want_final=False)
    g, u = s.end()
    assert s.stats()["lookups"] == meta["lookups"]
    res = "".join("%d,%d,%d\n" % (i, g[i], u[i]) for i in range(parent.size)).encode()
    assert hashlib.sha256(res).hexdigest() == meta["result_sha256"]
    assert res == gzip.open(os.path.join(gold_dir, "e2e_ref_full_result.txt.gz")).read()
    s.close(); db.close()


def test_fmix64_golden_on_gpu(kat):
    """Hashtable::integerHash as the device computes it, against the reference's own values"""
    assert np.array_equal(kmer_id_amd.hash_keys(kat["fmix_in"]), kat["fmix_out"])


@pytest.mark.parametrize("flags_b", [0, KID_FLAG_REF_GEOMETRY, KID_FLAG_HOST_BUILD])
def test_seen_bitmaps_merge_across_independent_tables(flags_b):
    """What N ranks do: every rank builds ITS OWN replica of the table (cell placement depends on the race order of
    the GPU builder, or is another geometry altogether), classifies its shard and ORs seen-bitmaps with the others.
    Bits are entry ordinals, so the merge must be exact whatever the placements -- duplicates included: the DB holds
    keys inserted twice with different targets (first insert wins) and a target-0 entry."""
    parent, cum, keys, targets = small_db(1e-3)
    rng = np.random.default_rng(5)
    dup = rng.choice(keys.size, 3000, replace=False)
    keys = np.concatenate([keys, keys[dup], keys[dup[:500]]])
    targets = np.concatenate([targets, (targets[dup] % 5000 + 7).astype(np.uint32), np.full(500, 2, np.uint32)])
    targets[dup[:20]] = 0  # first insert leaves the cell empty: the later copy with a target becomes visible
    odb = oracle_db(parent, keys, targets, 20)
    db_a = KmerDB(keys, targets, parent, k=K, log2_slots=20)
    db_b = KmerDB(keys, targets, parent, k=K, log2_slots=20, flags=flags_b)
    n, L = 16000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=777)
    # make sure the duplicated keys are hit: implant some of them
    for j, e in enumerate(dup[:2000].tolist()):
        v = int(keys[e])
        bases[j * L + 40:j * L + 70] = np.frombuffer("".join("ACGT"[(v >> (2 * (29 - i))) & 3] for i in range(30)).encode(), np.uint8)
    off = synth.fixed_offsets(n, L)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    eg, eu = os_.counts()
    a = db_a.sample(); fa = a.classify(bases, off[:n // 2 + 1])
    b = db_b.sample(); fb = b.classify(bases, off[n // 2:])
    assert np.array_equal(np.concatenate([fa, fb]), exp)
    nbytes = a.seen_bytes()
    assert nbytes == b.seen_bytes()
    half = (nbytes // 2) & ~15
    a.seen_or(0, b.seen_export(0, half))
    b.seen_or(half, a.seen_export(half, nbytes - half))
    u = a.ucount_range(0, half * 8) + b.ucount_range(half * 8, nbytes * 8)
    assert np.array_equal(a.gcount() + b.gcount(), eg)
    assert np.array_equal(u, eu)
    for x in (a, b, db_a, db_b):
        x.close()


def test_replicas_and_merged_end(seeded):
    """kid_db_replicate + kid_sample_end_merged: three replicas of the table (on this one GPU), the reads dealt out in
    ragged batches over their samples, one merged result = the single-table result"""
    parent, cum, keys, targets, odb, db = seeded
    n, L = 15000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=31337)
    off = synth.fixed_offsets(n, L)
    whole = db.sample(); exp = whole.classify(bases, off)
    g, u = whole.end()
    reps = [db, db.replicate(0), db.replicate(0)]
    assert reps[1].info.n_entries == db.info.n_entries and reps[1].info.geometry == db.info.geometry
    samples = [r.sample() for r in reps]
    cuts = [0, 1, 1000, 1001, 5000, 9999, 12000, n]
    got = np.empty(n, np.uint32)
    for j in range(len(cuts) - 1):
        a, b = cuts[j], cuts[j + 1]
        got[a:b] = samples[j % 3].classify(bases, off[a:b + 1])
    assert np.array_equal(got, exp)
    gm, um = kmer_id_amd.end_merged(samples)
    assert np.array_equal(gm, g) and np.array_equal(um, u)
    for s_ in samples + [whole]:
        s_.close()
    for r in reps[1:]:
        r.close()


@pytest.mark.parametrize("entry", ["host_buffers", "device_resident", "device_resident_then_a_larger_batch"])
def test_very_long_records_take_the_two_pass_path(seeded, entry):
    """FASTA contigs classified whole (kmer_read_vf6.cpp:803-861): records of more than 65536 k-mers go through
    kid_long_hits_kernel + kid_long_fold_kernel (every k-mer looked up by a lane of its own, one workgroup folds a
    record's hits in position order) instead of being walked by one wave -- sorted out on the device (the prepare kernel
    flags them, kid_long_plan_kernel places them), so also for batches whose offsets the host never saw.  Same
    per-record results and counters as the oracle's sequential fold: hits of several lineages in an order that matters,
    N and lower-case stretches, a start/stop range, a record just above and one just below the threshold, short reads in
    between.  "then_a_larger_batch": a second batch four times as large through the same sample (the hit array, sized by
    the first, has to grow; lists and plans of the scratch sets are reused)."""
    parent, cum, keys, targets, odb, db = seeded
    rng = np.random.default_rng(99)
    lut = np.frombuffer(b"ACGT", np.uint8)

    def record(n_bases, every):
        b = lut[rng.integers(0, 4, n_bases)].copy()
        for p in range(500, n_bases - 40, every):   # DB k-mers of random targets, either strand: the fold order matters
            v = int(keys[rng.integers(0, keys.size)])
            if rng.random() < 0.5:
                v = int(synth_revcomp(v))
            b[p:p + 30] = np.frombuffer("".join("ACGT"[(v >> (2 * (29 - i))) & 3] for i in range(30)).encode(), np.uint8)
        for p in rng.integers(0, n_bases - 50, 20):
            b[p] = ord("N")
        q = int(rng.integers(1000, n_bases - 2000))
        b[q:q + 300] |= 0x20                         # a lower-case stretch
        return b

    def synth_revcomp(v):
        r = 0
        for i in range(30):
            r = (r << 2) | (3 - ((v >> (2 * i)) & 3))
        return r

    recs = [record(200_000, 3000), synth.reads(cum, parent, 40, 150, K, r0=1), record(65536 + 29 + 1, 900),
            record(65536 + 29, 900), record(300_000, 40), synth.reads(cum, parent, 25, 150, K, r0=77), record(90_000, 100_000)]
    seqs = []
    for r_ in recs:
        if r_.size in (40 * 150, 25 * 150):
            seqs += [r_[i * 150:(i + 1) * 150] for i in range(r_.size // 150)]
        else:
            seqs.append(r_)
    bases = np.concatenate(seqs)
    off = np.zeros(len(seqs) + 1, np.uint64)
    off[1:] = np.cumsum([x.size for x in seqs])
    start = np.zeros(len(seqs), np.int32)
    stop = np.array([x.size - 1 for x in seqs], np.int32)
    start[0], stop[0] = 137, 199_000                 # a range inside the first long record
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off, start, stop)
    eg, eu = os_.counts()
    assert len(set(exp[[0, 41, 42, 43, len(seqs) - 1]].tolist())) > 1
    s = db.sample()
    if entry == "host_buffers":
        got = s.classify(bases, off, start, stop)
    else:
        import torch

        def on_device(bs, of, st_, sp_):
            pad = np.zeros(bs.size + 64, np.uint8); pad[:bs.size] = bs
            d_b = torch.from_numpy(pad).cuda()
            d_o = torch.from_numpy(of.view(np.int64)).cuda()
            d_s, d_e = torch.from_numpy(st_).cuda(), torch.from_numpy(sp_).cuda()
            d_out = torch.full((of.size - 1,), -1, dtype=torch.int32, device="cuda")
            s.classify_device(d_b.data_ptr(), bs.size, d_o.data_ptr(), of.size - 1, d_start=d_s.data_ptr(), d_stop=d_e.data_ptr(),
                              d_out=d_out.data_ptr())
            torch.cuda.synchronize()
            return d_out.cpu().numpy().view(np.uint32)
        got = on_device(bases, off, start, stop)
        if entry == "device_resident_then_a_larger_batch":
            nb = bases.size
            b4 = np.concatenate([bases] * 4)
            o4 = np.concatenate([off[:-1] + np.uint64(i * nb) for i in range(4)] + [np.array([4 * nb], np.uint64)])
            got4 = on_device(b4, o4, np.tile(start, 4), np.tile(stop, 4))
            assert np.array_equal(got4, np.tile(exp, 4))
            os_.classify(b4, o4, np.tile(start, 4), np.tile(stop, 4))
            eg, eu = os_.counts()
    assert np.array_equal(got, exp)
    g, u = s.end()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    st, est = s.stats(), os_.stats()
    assert st["lookups"] == est["lookups"] and st["hits"] == est["hits"]
    assert st["reads"] == len(seqs) * (5 if entry == "device_resident_then_a_larger_batch" else 1)
    s.close()


def test_async_host_batches_and_pinned_buffers(seeded):
    """kid_classify_batch_async / kid_classify_fixed_async / kid_classify_wait: more batches than staging slots in
    flight, ragged sizes, pinned (kid_host_alloc) and pageable buffers, waits out of order and twice, an empty batch;
    the sample must end with the counters of the same reads classified synchronously"""
    from kmer_id_amd import PinnedBuffer
    parent, cum, keys, targets, odb, db = seeded
    n, L = 9000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=555)
    off = synth.fixed_offsets(n, L)
    ref = db.sample(); exp = ref.classify(bases, off); g, u = ref.end()
    s = db.sample()
    cuts = [0, 1, 700, 701, 2500, 2500, 4000, 6500, 8999, n]   # (one empty batch)
    outs, tickets, keep = [], [], []
    pin = PinnedBuffer(n * L)
    pin.array[:] = bases
    for j in range(len(cuts) - 1):
        a, b = cuts[j], cuts[j + 1]
        out = np.full(b - a, 0xFFFFFFFF, np.uint32)
        src = pin.array if j % 2 == 0 else bases            # pinned and pageable callers
        o = np.ascontiguousarray(off[a:b + 1])
        keep.append((o, out))
        tickets.append(s.classify_async(src, o, out=out))
        outs.append((a, b, out))
    assert tickets[4] == 0                                     # the empty batch
    for t in reversed(tickets):                                # out of order; the slots of the early ones were reused long ago
        s.wait(t)
    s.wait(tickets[0])                                         # waiting again is harmless
    got = np.concatenate([o for _, _, o in outs])
    assert np.array_equal(got, exp)
    # fixed-length form straight from pinned memory, mixed with a synchronous call on the same sample
    out_pin = PinnedBuffer(4 * 3000)
    t = s.classify_fixed_async(pin.ptr, L, 3000, out_pin.ptr)
    sync_part = s.classify(bases[3000 * L:], off[3000:] - off[3000])
    s.wait(t)
    assert np.array_equal(out_pin.array.view(np.uint32)[:3000], exp[:3000]) and np.array_equal(sync_part, exp[3000:])
    g2, u2 = s.end()
    assert np.array_equal(g2, 2 * g) and np.array_equal(u2, u)   # every read twice: reads double, distinct k-mers do not
    with pytest.raises(kmer_id_amd.KidError):
        s.wait(10 ** 9)                                         # a ticket that was never issued
    pin.close(); out_pin.close(); s.close(); ref.close()


def test_inputs_ready_option_and_streams(seeded):
    """KID_OPT_INPUTS_READY: pack + prepare of a device batch on the library's own stream; consecutive batches of one
    sample on different caller streams are ordered by the library.  Same counters as the plain path."""
    import torch
    parent, cum, keys, targets, odb, db = seeded
    n, L = 8000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=4711)
    off = synth.fixed_offsets(n, L)
    ref = db.sample(); exp = ref.classify(bases, off); g, u = ref.end()
    dev = torch.device("cuda", 0)
    d = torch.from_numpy(np.concatenate([bases, np.zeros(64, np.uint8)])).to(dev)
    outs = [torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4)]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    for opt in (0, 1):
        s = db.sample()
        s.set_option(kmer_id_amd.KID_OPT_INPUTS_READY, opt)
        for i in range(4):
            s.classify_fixed_device(d.data_ptr(), L, n, d_out=outs[i].data_ptr(), stream=streams[i % 2].cuda_stream)
        g4, u4 = s.end()
        assert np.array_equal(g4, 4 * g) and np.array_equal(u4, u)
        for o in outs:
            assert np.array_equal(o.cpu().numpy().view(np.uint32), exp)
        s.close()
    with pytest.raises(kmer_id_amd.KidError):
        ref.set_option(99, 1)
    ref.close()


def test_merge_sample_over_rccl_single_rank(seeded):
    """kmer_id_amd.dist.merge_sample with the collectives forced on (world size 1, backend nccl = RCCL)"""
    import os
    import torch
    import torch.distributed as dist
    from kmer_id_amd.dist import merge_sample
    parent, cum, keys, targets, odb, db = seeded
    n, L = 6000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=99)
    s = db.sample(); s.classify(bases, synth.fixed_offsets(n, L), want_final=False)
    g, u = s.end()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    created = False
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        gm, um = merge_sample(s, torch.device("cuda", 0), force_collectives=True)
    finally:
        if created:
            dist.destroy_process_group()
    assert np.array_equal(gm, g) and np.array_equal(um, u)
    s.close()


def test_config5_fungal_vf6_250bp():
    """BASELINE configs[4]: fungal taxonomy (derived from the reference's fung1_list_vf6.txt by
    tools/taxonomy_from_list.py, 7078 nodes, 7 ranks deep), synthetic probes, 250 bp reads, U = T"""
    parent, cnt = synth.load_taxonomy("fungal")
    cum = synth.cumulative(synth.scaled_counts(cnt, 0.02))
    keys, targets = synth.db_keys(cum, K, seed=0xF6)
    odb = oracle_db(parent, keys, targets, 19, flags=ob.KO_FLAG_U_IS_T)
    n, L = 12000, 250
    bases = synth.reads(cum, parent, n, L, K, db_seed=0xF6, read_seed=0x250).copy()
    t = np.flatnonzero(bases == ord("T"))
    bases[t[::5]] = ord("U")
    off = synth.fixed_offsets(n, L)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    eg, eu = os_.counts()
    assert (exp > 1).sum() > n // 4 and len(np.unique(exp)) > 500
    for flags in (KID_FLAG_U_IS_T, KID_FLAG_U_IS_T | KID_FLAG_REF_GEOMETRY):
        db = KmerDB(keys, targets, parent, k=K, log2_slots=19, flags=flags)
        assert db.info.tree_depth == 7 and db.info.ntar == 7078
        s = db.sample()
        assert np.array_equal(s.classify(bases, off), exp)
        g, u = s.end()
        assert np.array_equal(g, eg) and np.array_equal(u, eu)
        assert s.stats()["lookups"] == os_.stats()["lookups"]
        s.close(); db.close()


@pytest.mark.parametrize("k", [15, 21, 24, 27, 29, 31])
def test_other_kmer_lengths(k):
    """k is a run-time parameter of the library (KSIZE = 30 in all shipped reference programs): the
    minimizer placement is used for k >= 24 (m = k - 14, window 15; k = 31: m = 16, window 16), the
    reference placement below that; the k = 30 kernel specialisation must not leak into other k"""
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, 5e-4))
    keys, targets = synth.db_keys(cum, k, seed=0x100 + k)
    odb = oracle_db(parent, keys, targets, 19, k=k)
    db = KmerDB(keys, targets, parent, k=k, log2_slots=19)
    assert db.info.geometry == (1 if k >= 24 else 0)
    q = np.concatenate([keys[::7], keys[:3000] ^ np.uint64(1)])
    assert np.array_equal(db.lookup(q), odb.get(q))
    n, L = 8000, 151
    bases = synth.reads(cum, parent, n, L, k, db_seed=0x100 + k, read_seed=k)
    off = synth.fixed_offsets(n, L)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    s = db.sample()
    assert np.array_equal(s.classify(bases, off), exp)
    g, u = s.end(); eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    assert s.stats()["lookups"] == os_.stats()["lookups"] and (exp > 1).sum() > n // 5
    s.close(); db.close()


# ------------------------------------------------------------------ the pair kernel's lookup queue under load
def _genome_db(parent, n_genomes, genome_len, rng):
    """k-mers of random genomes; the targets of one genome walk up and down one lineage (so the msca
    fold has work) with a few k-mers of a foreign lineage in between (so it also meets real LCAs)."""
    depth = np.zeros(parent.size, np.int64)
    for t in range(2, parent.size):
        d, x = 0, t
        while x > 1 and d < 64:
            x = int(parent[x]); d += 1
        depth[t] = d
    leaves = np.flatnonzero(depth >= 3)
    code = np.zeros(256, np.int64)
    for i, ch in enumerate(b"ACGT"):
        code[ch] = i
    genomes, keys, targets = [], [], []
    for g in range(n_genomes):
        seq = rng.choice(np.frombuffer(b"ACGT", np.uint8), genome_len)
        genomes.append(seq)
        c = code[seq]
        nwin = genome_len - K + 1
        key = np.zeros(nwin, np.uint64)
        for j in range(K):
            key = (key << np.uint64(2)) | c[j:j + nwin].astype(np.uint64)
        t0 = int(rng.choice(leaves))
        lineage = [t0, int(parent[t0]), int(parent[int(parent[t0])])]
        tg = np.array(lineage, np.uint32)[rng.integers(0, 3, nwin)]
        foreign = rng.random(nwin) < 0.02
        tg[foreign] = rng.choice(leaves, int(foreign.sum())).astype(np.uint32)
        keys.append(key); targets.append(tg)
    return genomes, np.concatenate(keys), np.concatenate(targets)


def test_many_hits_per_read_switch_the_hit_log_off_and_reset_switches_it_on():
    """kmer_seen (newkmer_10nx.cpp:596-603) on a sample whose reads hit 100 times each: after the first pass over the
    hit log has reported more than 8 entries per read, the library stops logging and the resolvers set the bits with
    atomics (kid_seenlog_pace).  The counters must not notice: batches before the switch (logged), at it, behind it,
    then the same sample reset and used for sparse reads again."""
    parent, _ = synth.load_taxonomy("bact10")
    rng = np.random.default_rng(4242)
    genomes, keys, targets = _genome_db(parent, 120, 2500, rng)
    odb = oracle_db(parent, keys, targets, 21)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=21)
    L, n = 150, 40_000

    def dense_batch():
        gi = rng.integers(0, len(genomes), n)
        pos = rng.integers(0, 2500 - L + 1, n)
        return np.concatenate([genomes[g][p:p + L] for g, p in zip(gi, pos)])
    off = synth.fixed_offsets(n, L)
    s = db.sample()
    os_ = ob.OracleSample(odb)
    for step in range(5):
        b = dense_batch()
        assert np.array_equal(s.classify(b, off), os_.classify(b, off))
        if step == 1:   # reading counters makes the library apply the log: the pass reports ~120 entries per read
            eg, eu = os_.counts()
            assert np.array_equal(s.gcount(), eg)
            nb = s.seen_bytes()
            assert np.array_equal(s.ucount_range(0, nb * 8), eu)
    g, u = s.end()
    eg, eu = os_.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu) and int(u.sum()) > 1000
    # the same sample again, sparse reads: the log is back (nothing to see from outside but the counters)
    s.reset()
    os2 = ob.OracleSample(odb)
    sparse = rng.choice(np.frombuffer(b"ACGT", np.uint8), n * L)
    view = sparse.reshape(n, L)
    few = rng.integers(0, n, 2000)
    for r in few:
        g_ = genomes[int(rng.integers(0, len(genomes)))]
        p_ = int(rng.integers(0, 2500 - 40))
        view[r, 10:50] = g_[p_:p_ + 40]
    for _ in range(2):
        assert np.array_equal(s.classify(sparse, off), os2.classify(sparse, off))
    g, u = s.end()
    eg, eu = os2.counts()
    assert np.array_equal(g, eg) and np.array_equal(u, eu) and int(g[2:].sum()) > 0
    s.close(); db.close()


@pytest.mark.parametrize("read_len", [150, 157, 100, 250, 285, 286])
def test_dense_hits_through_the_lookup_queue(read_len):
    """Reads cut from genomes whose every k-mer is in the DB: up to 128 queued lookups per read, runs
    of one read split over resolver chunks, more than 64 reads per wave (tags wrap), both strands,
    mutated copies with fewer hits, masked bases and reads without any hit in between.  157 bp = 128
    k-mers (the largest single group), 250 / 285 bp = two groups per read (a read stays open in the
    resolver between them), 286 bp = 257 k-mers (general loops)."""
    parent, _ = synth.load_taxonomy("bact10")
    rng = np.random.default_rng(77 + read_len)
    genomes, keys, targets = _genome_db(parent, 120, 2500, rng)
    odb = oracle_db(parent, keys, targets, 21)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=21)
    n = 560_000 if read_len == 150 else 60_000
    comp = np.zeros(256, np.uint8)
    for a, b_ in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b_
    bases = np.empty(n * read_len, np.uint8)
    kind = rng.integers(0, 4, n)
    gi = rng.integers(0, len(genomes), n)
    pos = rng.integers(0, 2500 - read_len + 1, n)
    strand = rng.integers(0, 2, n)
    for r in range(n):
        if kind[r] == 3:
            continue
        s = genomes[gi[r]][pos[r]:pos[r] + read_len]
        bases[r * read_len:(r + 1) * read_len] = comp[s[::-1]] if strand[r] else s
    rnd = np.flatnonzero(kind == 3)
    view = bases.reshape(n, read_len)
    view[rnd] = rng.choice(np.frombuffer(b"ACGT", np.uint8), (rnd.size, read_len))
    mut = np.flatnonzero(kind == 1)   # a substitution every ~25 bases: scattered hits
    m = rng.random((mut.size, read_len)) < 0.04
    sub = view[mut]
    sub[m] = rng.choice(np.frombuffer(b"ACGT", np.uint8), int(m.sum()))
    view[mut] = sub
    masked = np.flatnonzero(kind == 2)[::3]  # an N somewhere: two runs of hits
    view[masked, rng.integers(0, read_len, masked.size)] = ord("N")
    off = synth.fixed_offsets(n, read_len)
    os_ = ob.OracleSample(odb)
    exp = os_.classify(bases, off)
    eg, eu = os_.counts()
    s = db.sample()
    got = s.classify(bases, off)
    assert np.array_equal(got, exp)
    g, u = s.end()
    assert np.array_equal(g, eg) and np.array_equal(u, eu)
    st, est = s.stats(), os_.stats()
    assert st["lookups"] == est["lookups"] and st["hits"] == est["hits"]
    assert st["hits"] > 20 * n  # dense
    s.close()
    db.close()


def test_kernel_time_on_two_clocks(seeded):
    """kid_sample_kernel_time (HIP events) and kid_sample_kernel_time_device (the device's realtime counter,
    first workgroup start to last workgroup end) bracket the same launches."""
    parent, cum, keys, targets, odb, db = seeded
    n, read_len = 400_000, 150
    bases = synth.reads(cum, parent, n, read_len, K)
    import torch
    d = torch.from_numpy(bases).cuda()
    s = db.sample()
    s.classify_fixed_device(d.data_ptr(), read_len, n)  # warm-up (scratch allocation)
    s.set_timing(True)
    s.kernel_time(); s.kernel_time_device()
    for _ in range(5):
        s.classify_fixed_device(d.data_ptr(), read_len, n)
    ev_ms, ev_n = s.kernel_time()
    dev_ms, dev_n = s.kernel_time_device()
    assert ev_n == 5 and dev_n == 5
    assert 0 < dev_ms <= ev_ms * 1.05 and dev_ms >= ev_ms * 0.5
    s.close()
