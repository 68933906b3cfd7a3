"""kid_classify_fastq_async: a block of FASTQ text whose lines the host has found -- process_qual (newkmer_10nx.cpp:714-760),
its ">= 30" test (:757) and process_read (:452-617) all on the GPU -- against the oracle's process_qual + classification
and against the reference's own golden answers (kat_10nx.npz: 3000 process_qual cases with the final target of every
read the reference handed to process_read)."""
import numpy as np
import pytest

import kmer_id_amd
from kmer_id_amd import KmerDB
from helpers import K, ob, oracle_db, parse_probes_text, small_db, synth, unpack_strings

pytestmark = pytest.mark.gpu


def index_fastq(text):
    """The host's share of process_fqgz (:762-816) restated in python for small inputs: split at '\\n', drop one trailing
    '\\r', skip empty lines without advancing the 4-line phase, drop the unterminated tail.
    -> uint32[n, 4] (seq_off, seq_len, qual_off, qual_len), list of header lines"""
    recs, accs = [], []
    pos, phase, cur = 0, 0, [0, 0, 0, 0]
    acc = b""
    while True:
        nl = text.find(b"\n", pos)
        if nl < 0:
            break
        l = nl - pos
        if l > 0 and text[pos + l - 1:pos + l] == b"\r":
            l -= 1
        if l > 0:
            if phase == 0:
                acc = text[pos:pos + l]
            elif phase == 1:
                cur[0], cur[1] = pos, l
            elif phase == 3:
                cur[2], cur[3] = pos, l
                recs.append(list(cur)); accs.append(acc)
            phase = (phase + 1) % 4
        pos = nl + 1
    return np.array(recs, np.uint32).reshape(-1, 4), accs


def fastq_text(seqs, quals, eols=(b"\n",), blank_every=0, final_newline=True):
    out = []
    for i, (s, q) in enumerate(zip(seqs, quals)):
        e = eols[i % len(eols)]
        out += [b"@r%d" % i, e, bytes(s), e, b"+", e, bytes(q), e]
        if blank_every and i % blank_every == 0:
            out += [e, b"\n"]       # blank lines (with and without '\r') between records
    text = b"".join(out)
    return text if final_newline else text[:-1]


@pytest.fixture(scope="module")
def dbs():
    parent, cum, keys, targets = small_db(1e-3)
    odb = oracle_db(parent, keys, targets, 20)
    db = KmerDB(keys, targets, parent, k=K, log2_slots=20)
    yield parent, cum, odb, db
    db.close()


def oracle_fastq(odb, seqs, quals):
    """what the reference does with these records: process_qual, then process_read on the kept ones"""
    called, start, stop = [], [], []
    for s, q in zip(seqs, quals):
        c, a, b = ob.process_qual(q, len(s), K)
        called.append(c); start.append(a); stop.append(b)
    called = np.array(called); start = np.array(start, np.int32); stop = np.array(stop, np.int32)
    kept = [i for i in range(len(seqs)) if called[i] == 1]
    data = np.frombuffer(b"".join(bytes(seqs[i]) for i in kept), np.uint8)
    off = np.zeros(len(kept) + 1, np.uint64)
    off[1:] = np.cumsum([len(seqs[i]) for i in kept])
    os_ = ob.OracleSample(odb)
    fin = os_.classify(data, off, start[kept], stop[kept])
    g, u = os_.counts()
    os_.close()
    final = np.zeros(len(seqs), np.uint32)
    final[kept] = fin
    return called, start, stop, final, g, u


@pytest.mark.parametrize("layout", ["lf", "crlf_blank_lines_no_final_newline"])
def test_fastq_block_vs_oracle(dbs, layout):
    parent, cum, odb, db = dbs
    n, L = 6000, 150
    bases = synth.reads(cum, parent, n, L, K, r0=4242).reshape(n, L)
    quals = synth.qualities(n, L, r0=4242)
    seqs = [bases[i].tobytes() for i in range(n)]
    qs = [quals[i].tobytes() for i in range(n)]
    # ragged records too: short reads, a read of exactly k and k + 1 bases, a quality line longer than its sequence
    seqs += [seqs[0][:29], seqs[1][:30], seqs[2][:31], seqs[3][:77], seqs[4]]
    qs += [b"I" * 29, b"I" * 30, b"I" * 31, b"I" * 77, b"I" * 200]
    if layout == "lf":
        text = fastq_text(seqs, qs)
    else:
        text = fastq_text(seqs, qs, eols=(b"\n", b"\r\n"), blank_every=7, final_newline=False)
        seqs, qs = seqs[:-1], qs[:-1]          # the unterminated last line is dropped: its record is never complete
    recs, accs = index_fastq(text)
    assert recs.shape[0] == len(seqs)
    called, start, stop, final, g, u = oracle_fastq(odb, seqs, qs)
    s = db.sample()
    gf, gs, ge = s.classify_fastq(text, recs)
    gg, gu = s.end()
    kept = called == 1
    assert np.array_equal((ge - gs >= K), kept)
    assert np.array_equal(gs[kept], start[kept]) and np.array_equal(ge[kept], stop[kept])
    assert np.array_equal(gf, final)                      # (0 for the records process_qual drops)
    assert np.array_equal(gg, g) and np.array_equal(gu, u)  # dropped records are counted nowhere
    assert int(gg.sum()) == int(kept.sum()) and 0 < int(kept.sum()) < len(seqs)
    s.close()


def test_fastq_block_golden_process_qual(kat):
    """the reference's own process_qual + process_read answers (kat_10nx.npz) through the FASTQ block entry point"""
    text_db = __import__("gzip").decompress(bytes(kat["probes_gz"]))
    keys, targets = parse_probes_text(text_db)
    parent, _ = synth.load_taxonomy("bact10")
    db = KmerDB(keys, targets, parent, k=K, log2_slots=int(kat["log2_slots"]))
    quals = unpack_strings(kat["qual_qual_data"], kat["qual_qual_off"])
    seqs = unpack_strings(kat["qual_seq_data"], kat["qual_seq_off"])
    ok = [i for i in range(len(seqs)) if len(quals[i]) >= len(seqs[i]) and len(seqs[i]) > 0]  # (the others make the reference throw)
    text = fastq_text([seqs[i] for i in ok], [quals[i] for i in ok])
    recs, _ = index_fastq(text)
    s = db.sample()
    final, start, stop = s.classify_fastq(text, recs)
    exp = kat["qual_out"][ok]
    called = exp[:, 0] == 1
    assert np.array_equal((stop - start >= K), called)
    assert np.array_equal(start[called], exp[called, 1]) and np.array_equal(stop[called], exp[called, 2])
    assert np.array_equal(final[called].astype(np.int64), exp[called, 3])
    assert not final[~called].any()
    g, u = s.end()
    assert int(g.sum()) == int(called.sum())
    s.close(); db.close()


def test_fastq_block_quality_shorter_than_sequence(dbs):
    """std::string::at throws in the reference (:727): reported as KID_ERR_FORMAT when the sample is closed"""
    parent, cum, odb, db = dbs
    text = b"@a\nACGTACGTACGTACGTACGTACGTACGTACGTACGT\n+\nIIII\n"
    recs, _ = index_fastq(text)
    s = db.sample()
    s.classify_fastq(text, recs)
    with pytest.raises(kmer_id_amd.KidError) as e:
        s.end()
    assert e.value.status == -9
    s.close()


def test_fastq_block_bad_index_is_rejected(dbs):
    parent, cum, odb, db = dbs
    text = b"@a\nACGT\n+\nIIII\n"
    s = db.sample()
    with pytest.raises(kmer_id_amd.KidError):
        s.classify_fastq(text, np.array([[3, 400, 10, 4]], np.uint32))
    s.close()
