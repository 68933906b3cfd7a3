"""The host code's gzip reader (kmer_id_amd/host/kid_inflate.cpp) against zlib's gzread, which is what the reference
reads every input with (newkmer_10nx.cpp:673, :762-816; zlib is a system library, not part of the reference).  Same text
for every kind of deflate block and gzip member layout; and for files that are cut off or damaged, the same thing a
caller of gzread observes: a read error with zlib's message after the text in front of the damage (exit 255 in the
reference, :776), or -- for a file that ends inside a stream -- all the text, end of file, and a failing gzclose (:815).

tools/kid_gzcat.cpp prints a file's text with either reader; exit 0 = fine, 3 = read error, 4 = close failed.
Every file also goes through the parallel reader (kmer_id_amd/host/kid_pargz.cpp: block headers found from the outside,
pieces inflated into symbols, windows resolved afterwards) with small pieces, so that each file is cut many times."""
import gzip
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from kmer_id_amd import _build


@pytest.fixture(scope="module")
def gzcat():
    _build.build_tools()
    path = os.path.join(_build.BIN_DIR, "kid_gzcat")
    assert os.path.exists(path)
    return path


def run(gzcat, path, *flags):
    r = subprocess.run([gzcat] + list(flags) + [path], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return r.returncode, r.stdout, r.stderr.decode("latin-1").strip()


def same_as_zlib(gzcat, path, rooms=(1 << 20,), cut_off=False):
    rc_z, out_z, err_z = run(gzcat, path, "--zlib")
    for room in rooms:
        rc, out, err = run(gzcat, path, "--room", str(room))
        if cut_off and rc_z == 0 and rc == 4 and len(out_z) and len(out_z) % 0x4000 == 0:
            # A quirk of zlib 1.2.11 that is NOT reproduced: when a gzread call's buffer fills up at the moment the last
            # bytes of a cut-off file sit in inflate's bit accumulator, the next gzread sees "no input left, end of
            # file" and stops without an error -- the symbols in the accumulator are lost and gzclose succeeds.  Ours
            # hands out every whole symbol and fails the close, as zlib does at any other cutting point.
            assert out[:len(out_z)] == out_z and len(out) - len(out_z) <= 32 * 258
            continue
        assert rc == rc_z, (path, room, rc, rc_z, err, err_z)
        if rc == 3:
            # zlib drops what the failing gzread call had inflated; ours hands out everything in front of the damage
            assert out[:len(out_z)] == out_z and err == err_z, (path, room, err, err_z, len(out), len(out_z))
        else:
            assert out == out_z and err == err_z, (path, room, err, err_z, len(out), len(out_z))
    # pieces of the file inflated side by side (kid_pargz.cpp): the sequential reader's text and failures, byte for byte
    rc, out, err = run(gzcat, path, "--room", "70000")
    # (pieces of 4 KiB are smaller than a block: most of them do not fit and the stretch is read again in order --
    # with patience for it, and with the default of 4 misfits in a row after which the rest is read sequentially)
    for threads, chunk, patience in ((3, 4096, 1 << 20), (3, 4096, 4), (2, 30000, 4), (4, 1 << 20, 4)):
        rc_p, out_p, err_p = run(gzcat, path, "--threads", str(threads), "--chunk", str(chunk), "--patience", str(patience), "--room", "70000")
        err_p = "\n".join(l for l in err_p.splitlines() if not l.startswith("parallel: "))
        assert (rc_p, err_p) == (rc, err) and out_p == out, (path, threads, chunk, patience, rc_p, rc, err_p, err, len(out_p), len(out))
    return rc_z, out_z, err_z


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15, flags=0, extra=b"", name=b"", comment=b"", hcrc=False):
    c = zlib.compressobj(level, zlib.DEFLATED, -wbits, 8, strategy)
    body = c.compress(data) + c.flush()
    flg = (4 if extra else 0) | (8 if name else 0) | (16 if comment else 0) | (2 if hcrc else 0) | flags
    head = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0\0\x03"
    if extra:
        head += struct.pack("<H", len(extra)) + extra
    if name:
        head += name + b"\0"
    if comment:
        head += comment + b"\0"
    if hcrc:
        head += struct.pack("<H", zlib.crc32(head) & 0xffff)
    return head + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)


def texts():
    rng = np.random.default_rng(11)
    fastq = b"".join(b"@r%09d/1\n" % i + bytes(rng.choice(list(b"ACGT"), 150).astype(np.uint8)) + b"\n+\n" +
                     bytes(rng.integers(35, 74, 150).astype(np.uint8)) + b"\n" for i in range(3000))
    probes = b"".join(bytes(rng.choice(list(b"ACGT"), 30).astype(np.uint8)) + b",%d,%d,%d,1,1\n" % (i % 5982, i, i * 7) for i in range(20000))
    return {
        "fastq": fastq,
        "probes": probes,
        "zeros": bytes(300000),                                             # matches at distance 1, length 258
        "noise": bytes(rng.integers(0, 256, 200000).astype(np.uint8)),      # stored blocks
        "short": b"ACGT\n",                                                 # one fixed-code block
        "period3": b"abc" * 50000, "period7": b"0123456" * 30000, "period9": b"012345678" * 30000,  # overlapping copies
        "far": (bytes(rng.integers(0, 256, 32768).astype(np.uint8)) * 4),   # matches at the far end of the window
        "empty": b"",
        # every kind of block in one stream, several times over
        "mixed": b"".join(fastq[i * 150000:(i + 1) * 150000] + bytes(rng.integers(0, 256, 40000).astype(np.uint8)) +
                          probes[i * 150000:(i + 1) * 150000] + bytes(70000) + b"xyz" * 10 for i in range(4)),
    }


def test_same_text_for_every_block_type_and_room(gzcat, tmp_path):
    for name, data in texts().items():
        for tag, kw in (("l1", dict(level=1)), ("l6", dict(level=6)), ("l9", dict(level=9)), ("l0", dict(level=0)),
                        ("fixed", dict(strategy=zlib.Z_FIXED)), ("huff", dict(strategy=zlib.Z_HUFFMAN_ONLY)),
                        ("rle", dict(strategy=zlib.Z_RLE)), ("w9", dict(wbits=9))):
            p = str(tmp_path / ("%s_%s.gz" % (name, tag)))
            open(p, "wb").write(member(data, **kw))
            rc, out, _ = same_as_zlib(gzcat, p, rooms=(4096, 5001, 70000, 1 << 20))
            assert rc == 0 and out == data


def test_member_layouts(gzcat, tmp_path):
    t = texts()
    a, b = t["fastq"][:50000], t["probes"][:70000]
    cases = {
        "two": member(a) + member(b),
        "with_empty_members": member(b"") + member(a) + member(b"") + member(b) + member(b""),
        "header_fields": member(a, extra=b"\1\2\3\4\5", name=b"reads.fastq", comment=b"a comment", hcrc=True),
        "name_only": member(a, name=b"x" * 5000),
        "garbage_behind": member(a) + b"this is not a gzip header" * 10,
        "one_byte_behind": member(a) + b"\x1f",
        "zeros_behind": member(a) + bytes(1000),
        "plain_text": a,                        # gzread passes a file without the magic through
        "one_byte": b"\x1f",
        "empty_file": b"",
        "python_gzip": gzip.compress(b, 9),
    }
    # bgzip-style: a member every few KiB of text, each with an extra field, an empty member at the end
    small = [member(t["probes"][i:i + 9000], extra=b"BC\x02\x00\x00\x00") for i in range(0, 400000, 9000)]
    cases["many_small_members"] = b"".join(small) + member(b"", extra=b"BC\x02\x00\x1b\x00")
    for name, blob in cases.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        rc, _, _ = same_as_zlib(gzcat, p, rooms=(4096, 1 << 20))
        assert rc == 0, name
    bad = {
        "bad_method": b"\x1f\x8b\x07" + member(a)[3:],
        "reserved_flag": member(a, flags=0x20),
        "bad_header_crc": member(a, name=b"n", hcrc=True)[:12] + b"\xff\xff" + member(a, name=b"n", hcrc=True)[14:],
        "bad_crc": member(a)[:-8] + b"\0\0\0\0" + member(a)[-4:],
        "bad_length": member(a)[:-4] + b"\1\0\0\0",
        "second_member_damaged": member(a) + member(b)[:200] + b"\xff" * 50 + member(b)[250:],
        # wrong sums in the middle of a file of many small members: the text in front of that member's end, then the error
        "small_member_bad_crc": b"".join(small[:20]) + small[20][:-8] + b"\1\2\3\4" + small[20][-4:] + b"".join(small[21:]),
        "small_member_bad_length": b"".join(small[:7]) + small[7][:-4] + b"\1\0\0\0" + b"".join(small[8:]),
    }
    for name, blob in bad.items():
        p = str(tmp_path / (name + ".gz"))
        open(p, "wb").write(blob)
        rc, _, err = same_as_zlib(gzcat, p, rooms=(4096, 1 << 20))
        assert rc == 3 and err, name


def test_cut_off_files_end_quietly_and_fail_at_close(gzcat, tmp_path):
    t = texts()
    small = member(t["fastq"][:3000], name=b"n", hcrc=True)
    p = str(tmp_path / "cut.gz")
    for n in range(1, len(small)):               # every prefix of a small file, the header included
        open(p, "wb").write(small[:n])
        rc, _, _ = same_as_zlib(gzcat, p, rooms=(4096,), cut_off=True)
        assert rc in (0, 4)                      # (0: the one-byte file is "plain text")
    rng = np.random.default_rng(5)
    for name in ("fastq", "noise", "zeros", "period7", "mixed"):
        for kw in (dict(level=1), dict(level=6), dict(strategy=zlib.Z_FIXED)):
            blob = member(t[name], **kw) + member(t["probes"][:5000])
            for n in sorted(set(rng.integers(1, len(blob), 12).tolist() + [len(blob) - 1, len(blob) - 8, len(blob) - 9])):
                open(p, "wb").write(blob[:n])
                rc, _, _ = same_as_zlib(gzcat, p, rooms=(4096, 1 << 20), cut_off=True)
                assert rc in (0, 4)


def test_damaged_streams_fail_like_zlib(gzcat, tmp_path):
    t = texts()
    rng = np.random.default_rng(7)
    p = str(tmp_path / "flip.gz")
    seen = {}
    for name, kw in (("fastq", dict(level=6)), ("fastq", dict(level=1)), ("probes", dict(level=6)), ("probes", dict(strategy=zlib.Z_FIXED)),
                     ("noise", dict(level=6)), ("period7", dict(level=6)), ("mixed", dict(level=6)), ("mixed", dict(level=1))):
        blob = bytearray(member(t[name], **kw))
        for _ in range(40):
            at = int(rng.integers(10, len(blob)))
            bit = 1 << int(rng.integers(0, 8))
            blob[at] ^= bit
            open(p, "wb").write(bytes(blob))
            rc, _, err = same_as_zlib(gzcat, p, rooms=(4096, 1 << 20), cut_off=True)
            blob[at] ^= bit
            assert rc in (0, 3, 4)               # (4, or zlib's quiet 0: the damage makes the stream run past the end of the file)
            seen[err.split(": ", 1)[-1]] = seen.get(err.split(": ", 1)[-1], 0) + 1
    # the flips reach the different checks, not only the final CRC
    assert len(seen) >= 5 and "incorrect data check" in seen, seen
