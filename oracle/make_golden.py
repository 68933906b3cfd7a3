#!/usr/bin/env python3
"""Generate tests/golden/ from the COMPILED REFERENCE (oracle/_ref, built by
oracle/Makefile from /root/reference).  Runs only in the build container.

What is committed is data: inputs we made up and the numbers / result files the
reference program printed for them.  No reference source text is stored.

  tests/golden/kat_10nx.npz        unit known-answer vectors out of oracle/_ref/kat_10nx
                                   (integerHash, getHash, msca, process_qual, process_read)
  tests/golden/e2e_small/          hand-made + synthetic FASTQ.gz pairs with the files
                                   nk10_ref_small wrote for them (_result.txt, _reads.txt)
  tests/golden/e2e_seeded.json     parameters of a larger seeded run + sha256 / gz of its outputs
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kmer_id_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")
GOLD = os.path.join(ROOT, "tests", "golden")
K = 30


def revcomp_seq(s):
    return s[::-1].translate(str.maketrans("ACGTacgt", "TGCAtgca"))


def write_tree(path, parent, crlf=False):
    eol = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as fh:
        for y, x in enumerate(parent.tolist()):
            if y >= 2 and x != 1:
                fh.write("%d\t%d%s" % (x, y, eol))


def pack_strings(strs):
    data = np.frombuffer("".join(strs).encode("latin-1"), np.uint8).copy()
    off = np.zeros(len(strs) + 1, np.uint64)
    off[1:] = np.cumsum([len(s) for s in strs])
    return data, off


# ------------------------------------------------------------------ unit KATs
def make_kat():
    rng = np.random.default_rng(20260401)
    parent, cnt = synth.load_taxonomy("bact10")
    ntar = parent.size
    wd = tempfile.mkdtemp(prefix="kat_")
    write_tree(os.path.join(wd, "tree.txt"), parent)

    # DB for a 2^12-cell table at load ~0.63: long probe chains, duplicates, quirks
    cum = synth.cumulative(synth.scaled_counts(cnt, 2.2e-5))
    keys, targets = synth.db_keys(cum, K)
    keys, targets = keys[:2500], targets[:2500]
    lines = []
    exp_keys, exp_targets = [], []

    def add_line(seq, t):
        lines.append("%s,%d,0,%d,F,1" % (seq, t, len(lines)))

    for j, (key, t) in enumerate(zip(keys.tolist(), targets.tolist())):
        add_line(synth.key_to_seq(key), t)
    dup = rng.choice(2500, 40, replace=False)
    for j in dup.tolist():  # duplicates with another target: the first insert must win
        add_line(synth.key_to_seq(int(keys[j])), int((targets[j] + 7) % (ntar - 2) + 2))
    z0 = synth.key_to_seq(int(synth.db_keys(cum, K, seed=77, n=4)[0][0]))
    z1 = synth.key_to_seq(int(synth.db_keys(cum, K, seed=77, n=4)[0][1]))
    add_line(z0, 0)            # target 0: writes the key but leaves the cell empty ...
    add_line(z0, 5)            # ... so this one takes the same cell
    add_line(z1, 0)            # target 0 alone: never found
    long_seq = synth.key_to_seq(int(synth.db_keys(cum, K, seed=78, n=2)[0][0])) + "ACGTTGCA"  # 38 bases -> 9 windows
    add_line(long_seq, 11)
    add_line("ACGTNACGTACGTACGTACGTACGTACGTACGTACGTAC", 12)   # N resets, then one full window... (35 after N)
    add_line("acgtacgtacgtacgtacgtacgtacgtac", 13)            # lower case never inserts (process_kmer is upper-case only)
    add_line("A" * 30, 14)                                    # key 0 is a legal key
    add_line("T" * 30, 15)                                    # forward key of poly-T is NOT canonical: stored as is
    lines.append("garbage line without fields")
    lines.append("ACGTACGTACGTACGTACGTACGTACGTAC,notanumber,0,0,F,1")
    lines.append("")
    text = "\n".join(lines[:1000]) + "\n" + "\r\n".join(lines[1000:]) + "\r\n" + "ACGTACGTACGTACGTACGTACGTACGTAA,9,0,0,F,1"  # unterminated tail is dropped
    with gzip.open(os.path.join(wd, "probes.txt.gz"), "wb") as fh:
        fh.write(text.encode())

    fmix_in = np.concatenate([rng.integers(0, 2**64, 1000, dtype=np.uint64),
                              np.array([0, 1, 2**60 - 1, 2**64 - 1, 2**63], np.uint64)])
    np.savetxt(os.path.join(wd, "fmix_in.txt"), fmix_in, fmt="%d")

    def k2i(seq):
        v = 0
        for ch in seq:
            v = (v << 2) | "ACGT".index(ch)
        return v

    extra = [k2i(z0), k2i(z1), 0, k2i("T" * 30), k2i("ACGTACGTACGTACGTACGTACGTACGTAA")]
    extra += [k2i(long_seq[i:i + 30]) for i in range(9)]
    extra += [k2i("ACGTACGTACGTACGTACGTACGTACGTACGTAC"[i:i + 30]) for i in range(5)]
    lookup_in = np.concatenate([keys, rng.integers(0, 2**60, 3000, dtype=np.uint64), np.array(extra, np.uint64)])
    np.savetxt(os.path.join(wd, "lookup_in.txt"), lookup_in, fmt="%d")

    mx = rng.integers(1, ntar, 6000)
    my = rng.integers(1, ntar, 6000)
    mx[:200] = 1
    my[200:400] = 1
    my[400:600] = mx[400:600]
    np.savetxt(os.path.join(wd, "msca_in.txt"), np.stack([mx, my], 1), fmt="%d")

    # process_qual inputs: sequences carry DB k-mers so that final_targ is informative
    qual_seqs, qual_quals = [], []
    nq = 3000
    lens = rng.integers(1, 220, nq)
    lens[:50] = np.arange(1, 51)
    for i in range(nq):
        L = int(lens[i])
        seq = "".join(rng.choice(list("ACGT"), L))
        if L >= 30 and rng.random() < 0.6:
            p = int(rng.integers(0, L - 29))
            seq = seq[:p] + synth.key_to_seq(int(keys[rng.integers(0, 2500)])) + seq[p + 30:]
        mode = int(rng.integers(0, 6))
        if mode == 0:
            q = np.full(L, ord("I"))
        elif mode == 1:
            q = rng.integers(33, 75, L)
        elif mode == 2:
            q = np.where(np.arange(L) > L * rng.random(), rng.integers(33, 50, L), ord("I"))
        elif mode == 3:
            q = np.where(np.arange(L) < L * rng.random() * 0.5, rng.integers(33, 50, L), ord("F"))
        elif mode == 4:
            q = rng.integers(40, 60, L)
        else:
            q = rng.integers(33, 127, L)
        qs = "".join(chr(int(c)) for c in q)
        if rng.random() < 0.1:
            qs += "IIII"  # qual longer than seq is legal
        qual_seqs.append(seq)
        qual_quals.append(qs)
    with open(os.path.join(wd, "qual_in.txt"), "w") as fh:
        for s, q in zip(qual_seqs, qual_quals):
            fh.write(s + "\t" + q + "\n")

    # whole-read classification: order-dependent folds, N, lower case, short reads
    by_target = {}
    for key, t in zip(keys.tolist(), targets.tolist()):
        by_target.setdefault(t, []).append(key)
    tlist = sorted(by_target)
    reads = []
    for i in range(4000):
        L = int(rng.choice([29, 30, 31, 45, 100, 150, 150, 150, 250, 400, 1100, 2300]))
        seq = list("".join(rng.choice(list("ACGT"), L)))
        nimp = int(rng.integers(0, 7)) if L >= 30 else 0
        tb = int(rng.choice(tlist))
        lineage = []
        z = tb
        while z > 1:
            if z in by_target:
                lineage.append(z)
            z = int(parent[z])
        for j in range(nimp):
            if lineage and rng.random() < 0.7:  # mostly one lineage, at different ranks, in random order
                t = int(rng.choice(lineage))
            else:
                t = int(rng.choice(tlist))
            kseq = synth.key_to_seq(int(rng.choice(by_target[t])))
            if rng.random() < 0.5:
                kseq = revcomp_seq(kseq)
            p = int(rng.integers(0, L - 29))
            seq[p:p + 30] = list(kseq)
        s = "".join(seq)
        r = rng.random()
        if r < 0.05:
            s = s.lower()
        elif r < 0.10:
            s = "".join(c.lower() if rng.random() < 0.3 else c for c in s)
        if rng.random() < 0.15:
            p = int(rng.integers(0, L))
            s = s[:p] + str(rng.choice(list("NnRYUu-.*"))) + s[p + 1:]
        reads.append(s)
    reads.append("A" * 200)   # key 0 everywhere (target 14)
    reads.append("T" * 200)   # canonical of poly-T is poly-A
    reads.append(long_seq)
    with open(os.path.join(wd, "reads_in.txt"), "w") as fh:
        fh.write("\n".join(reads) + "\n")

    subprocess.check_call([os.path.join(REF, "kat_10nx"), wd], stdout=subprocess.DEVNULL)

    fmix_out = np.loadtxt(os.path.join(wd, "fmix_out.txt"), dtype=np.uint64)
    lookup_out = np.loadtxt(os.path.join(wd, "lookup_out.txt"), dtype=np.int64)
    msca_out = np.loadtxt(os.path.join(wd, "msca_out.txt"), dtype=np.int64)
    msca_all = open(os.path.join(wd, "msca_all.txt")).read().split()
    qual_out = np.loadtxt(os.path.join(wd, "qual_out.txt"), dtype=np.int64)
    reads_out = np.loadtxt(os.path.join(wd, "reads_out.txt"), dtype=np.int64)
    rc = np.loadtxt(os.path.join(wd, "reads_counts.txt"), dtype=np.int64).reshape(-1, 3)
    assert len(fmix_out) == len(fmix_in) and len(lookup_out) == len(lookup_in)
    assert len(qual_out) == nq and len(reads_out) == len(reads)

    probes_data = np.frombuffer(gzip.compress(text.encode(), 9), np.uint8)
    qs_data, qs_off = pack_strings(qual_seqs)
    qq_data, qq_off = pack_strings(qual_quals)
    rd_data, rd_off = pack_strings(reads)
    np.savez_compressed(
        os.path.join(GOLD, "kat_10nx.npz"),
        log2_slots=np.int64(12), k=np.int64(K), ntar=np.int64(ntar),
        probes_gz=probes_data,
        fmix_in=fmix_in, fmix_out=fmix_out,
        lookup_in=lookup_in, lookup_out=lookup_out,
        msca_x=mx, msca_y=my, msca_out=msca_out,
        msca_all_ntar=np.int64(int(msca_all[0])), msca_all_sum=np.uint64(int(msca_all[1])),
        qual_seq_data=qs_data, qual_seq_off=qs_off, qual_qual_data=qq_data, qual_qual_off=qq_off, qual_out=qual_out,
        reads_data=rd_data, reads_off=rd_off, reads_final=reads_out, reads_counts=rc,
    )
    shutil.rmtree(wd)
    print("kat_10nx.npz: %d fmix, %d lookups (%d hits), %d msca, %d quals (%d kept), %d reads (%d classified)" % (
        len(fmix_in), len(lookup_in), int((lookup_out > 0).sum()), len(mx), nq, int(qual_out[:, 0].sum()),
        len(reads), int((reads_out > 0).sum())))


# ------------------------------------------------------------------ end-to-end runs of the real program
def run_ref(binary, cwd, fastq_dir):
    out = subprocess.run([os.path.join(REF, binary), fastq_dir], cwd=cwd, stdout=subprocess.PIPE, check=True)
    return out.stdout.decode()


def setup_db_dir(cwd, parent, keys, targets, crlf=True):
    os.makedirs(os.path.join(cwd, "bact10"))
    write_tree(os.path.join(cwd, "bact10", "btree_10.txt"), parent, crlf=crlf)
    with open(os.path.join(cwd, "bact10", "bData10.txt"), "w") as fh:
        fh.write("4\tCP000828\r\n3\tAFEJ01\r\n")
    synth.write_probes_gz(os.path.join(cwd, "bact10", "probes10.txt.gz"), keys, targets, K)


E2E_SCALE = 2e-4


def make_e2e_small():
    rng = np.random.default_rng(7)
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, E2E_SCALE))
    keys, targets = synth.db_keys(cum, K)
    cwd = tempfile.mkdtemp(prefix="e2e_")
    setup_db_dir(cwd, parent, keys, targets)
    fq = os.path.join(cwd, "fq") + "/"
    os.makedirs(fq)
    outdir = os.path.join(GOLD, "e2e_small")
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir)

    # S1: 400 synthetic pairs, mixed qualities, LF
    n, L = 400, 150
    b1 = synth.reads(cum, parent, n, L, K, r0=0)
    b2 = synth.reads(cum, parent, n, L, K, r0=n)
    synth.write_fastq_gz(fq + "S1_R1_tr.fastq.gz", b1, synth.qualities(n, L, r0=0), L, mate=1)
    synth.write_fastq_gz(fq + "S1_R2_tr.fastq.gz", b2, synth.qualities(n, L, r0=n), L, mate=2)
    # S2: reader quirks: CRLF, blank lines (do not advance the 4-line phase), no final newline,
    # ragged lengths, a species with more than 12 reads (only the first 12 are saved)
    n2 = 120
    lens = rng.integers(20, 260, n2)
    hot = [k_ for k_, t in zip(keys.tolist(), targets.tolist()) if t == int(targets[len(targets) // 2])][:3]
    recs1, recs2 = [], []
    for i in range(n2):
        Lr = int(lens[i])
        s = "".join(rng.choice(list("ACGT"), Lr))
        if Lr >= 60 and i % 3 != 2:
            s = s[:10] + synth.key_to_seq(int(hot[i % len(hot)])) + s[40:]
        q = "".join(chr(int(c)) for c in np.where(np.arange(Lr) >= Lr - int(rng.integers(0, 12)), 35, 70))
        (recs1 if i % 2 == 0 else recs2).append(("@q%d some comment" % i, s, "+", q))
    def dump(path, recs, eol, blank_every, final_newline):
        parts = []
        for j, r in enumerate(recs):
            for f in r:
                parts.append(f)
            if blank_every and j % blank_every == 1:
                parts.append("")
        text = eol.join(parts) + (eol if final_newline else "")
        with gzip.open(path, "wb") as fh:
            fh.write(text.encode())
    dump(fq + "S2_R1_tr.fastq.gz", recs1, "\r\n", 5, True)
    dump(fq + "S2_R2_tr.fastq.gz", recs2, "\n", 7, False)  # the unterminated last quality line is dropped
    # a file that only contains the R1 marker inside a longer name still defines a sample prefix
    stdout = run_ref("nk10_ref_small", cwd, fq)
    for f in sorted(os.listdir(fq)):
        shutil.copy(os.path.join(fq, f), os.path.join(outdir, f))
    with open(os.path.join(outdir, "stdout.txt"), "w") as fh:
        fh.write(stdout.replace(fq, "<DIR>"))
    with open(os.path.join(outdir, "params.json"), "w") as fh:
        json.dump({"db": "bact10", "scale": E2E_SCALE, "k": K, "db_seed": synth.DB_SEED, "n_keys": int(keys.size)}, fh)
    shutil.rmtree(cwd)
    print("e2e_small:", sorted(os.listdir(outdir)))


def make_e2e_seeded():
    """A larger run whose inputs are re-generated from seeds in the test; only the
    reference's outputs are stored (gz)."""
    parent, cnt = synth.load_taxonomy("bact10")
    scale = 1e-3
    cum = synth.cumulative(synth.scaled_counts(cnt, scale))
    keys, targets = synth.db_keys(cum, K)
    cwd = tempfile.mkdtemp(prefix="e2e_")
    setup_db_dir(cwd, parent, keys, targets, crlf=False)
    fq = os.path.join(cwd, "fq") + "/"
    os.makedirs(fq)
    n, L = 20000, 150
    synth.write_fastq_gz(fq + "big_R1_tr.fastq.gz", synth.reads(cum, parent, n, L, K, r0=0), synth.qualities(n, L, r0=0), L, mate=1)
    synth.write_fastq_gz(fq + "big_R2_tr.fastq.gz", synth.reads(cum, parent, n, L, K, r0=n), synth.qualities(n, L, r0=n), L, mate=2)
    run_ref("nk10_ref_small", cwd, fq)
    res = open(fq + "big_result.txt", "rb").read()
    rds = open(fq + "big_reads.txt", "rb").read()
    with gzip.open(os.path.join(GOLD, "e2e_seeded_result.txt.gz"), "wb", 9) as fh:
        fh.write(res)
    # _reads.txt of this run is 2.6 MB: only its sha256 is kept
    with open(os.path.join(GOLD, "e2e_seeded.json"), "w") as fh:
        json.dump({"db": "bact10", "scale": scale, "k": K, "n_pairs": n, "read_len": L, "db_seed": synth.DB_SEED,
                   "read_seed": synth.READ_SEED, "qual_seed": 0x9A1, "n_keys": int(keys.size),
                   "result_sha256": hashlib.sha256(res).hexdigest(), "reads_sha256": hashlib.sha256(rds).hexdigest()}, fh)
    shutil.rmtree(cwd)
    print("e2e_seeded: result %d bytes, reads %d bytes" % (len(res), len(rds)))


def _mixed_records(rng, keys, n, with_u=True):
    """(acc, seq, qual) records with DB k-mers, lower case, N and (optionally) U bases"""
    recs = []
    for i in range(n):
        L = int(rng.integers(25, 320))
        s = list("".join(rng.choice(list("ACGT"), L)))
        for _ in range(int(rng.integers(0, 4))):
            if L >= 30:
                p = int(rng.integers(0, L - 29))
                ks = synth.key_to_seq(int(keys[rng.integers(0, keys.size)]))
                if rng.random() < 0.5:
                    ks = revcomp_seq(ks)
                s[p:p + 30] = list(ks)
        s = "".join(s)
        r = rng.random()
        if with_u and r < 0.3:
            s = s.replace("T", "U", int(rng.integers(1, 6)))
        elif r < 0.4:
            s = s.lower()
        elif r < 0.5:
            p = int(rng.integers(0, L)); s = s[:p] + "N" + s[p + 1:]
        q = "".join(chr(int(c)) for c in np.where(np.arange(L) >= L - int(rng.integers(0, 15)), 36, 72))
        recs.append(("r%d" % i, s, q))
    return recs


def _write_inputs(d, rng, keys, with_u):
    """four files, one per reader: .fastq.gz, .fasta.gz (multi-line), .fasta and .fastq (plain, with the
    token/blank-line quirks of the getline + >> readers)"""
    recs = _mixed_records(rng, keys, 400, with_u)
    with gzip.open(os.path.join(d, "a.fastq.gz"), "wb") as fh:
        fh.write("".join("@%s\n%s\n+\n%s\n" % r for r in recs[:120]).encode())
    with gzip.open(os.path.join(d, "b.fasta.gz"), "wb") as fh:
        out = []
        for acc, s, _ in recs[120:220]:
            out.append(">%s some description" % acc)
            for i in range(0, len(s), 70):
                out.append(s[i:i + 70])
        long = "".join(r[1] for r in recs[220:240]).replace("N", "A")
        out.append(">contig_long")
        out += [long[i:i + 60] for i in range(0, len(long), 60)]
        fh.write(("\n".join(out) + "\n").encode())
    with open(os.path.join(d, "c.fasta"), "w", newline="") as fh:
        out = []
        for j, (acc, s, _) in enumerate(recs[240:320]):
            out.append(">%s extra tokens ignored" % acc)
            out.append(s[:len(s) // 2] + "   trailing junk")
            if j % 7 == 3:
                out.append("")            # blank line: the previous token is appended again
            out.append(s[len(s) // 2:])
            if j % 11 == 5:
                out.append("   ")
        fh.write("\r\n".join(out) + "\r\n")
    with open(os.path.join(d, "d.fastq"), "w", newline="") as fh:
        out = []
        for j, (acc, s, q) in enumerate(recs[320:400]):
            # (a blank line here would re-use the previous token, shift the 4-line phase and make the
            #  reference die in qual.at(): covered by the exit-code test instead)
            out += ["@%s comment" % acc, s + "\tjunk", "+", q + " junk"]
        fh.write("\n".join(out) + "\n")
    return ["a.fastq.gz", "b.fasta.gz", "c.fasta", "d.fastq"]


def make_e2e_vf6():
    rng = np.random.default_rng(11)
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, E2E_SCALE))
    keys, targets = synth.db_keys(cum, K)
    cwd = tempfile.mkdtemp(prefix="vf6_")
    os.makedirs(os.path.join(cwd, "DB")); os.makedirs(os.path.join(cwd, "J"))
    write_tree(os.path.join(cwd, "DB", "DB_tree.txt"), parent, crlf=True)
    with open(os.path.join(cwd, "DB", "DB_data.txt"), "w", newline="") as fh:
        for t in range(2, parent.size, 37):
            fh.write("%d\tACC%06d\r\n" % (t, t))
        fh.write("%d\tLAST\r\n" % (parent.size - 1))
    synth.write_probes_gz(os.path.join(cwd, "DB", "DB_probes.txt.gz"), keys, targets, K)
    ind = os.path.join(cwd, "in"); os.makedirs(ind)
    files = _write_inputs(ind, rng, keys, with_u=True)
    with open(os.path.join(cwd, "J", "J.txt"), "w") as fh:
        fh.write("jobA 2\nin/%s\nin/%s\n\njobB 2\nin/%s\nin/%s\n" % tuple(files))
    hot = int(targets[len(targets) // 3])
    outdir = os.path.join(GOLD, "e2e_vf6")
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(os.path.join(outdir, "in"))
    for f in files:
        shutil.copy(os.path.join(ind, f), os.path.join(outdir, "in", f))
    shutil.copy(os.path.join(cwd, "J", "J.txt"), os.path.join(outdir, "J.txt"))
    shutil.copy(os.path.join(cwd, "DB", "DB_data.txt"), os.path.join(outdir, "DB_data.txt"))
    for tag, extra in (("plain", []), ("target", ["-target", str(hot)])):
        out = subprocess.run([os.path.join(REF, "vf6_ref_small"), "-name", "DB", "-jname", "J"] + extra, cwd=cwd,
                             stdout=subprocess.PIPE, check=True).stdout.decode()
        os.makedirs(os.path.join(outdir, tag))
        with open(os.path.join(outdir, tag, "stdout.txt"), "w") as fh:
            fh.write(out)
        for f in sorted(os.listdir(os.path.join(cwd, "J"))):
            if f != "J.txt":
                shutil.move(os.path.join(cwd, "J", f), os.path.join(outdir, tag, f))
    with open(os.path.join(outdir, "params.json"), "w") as fh:
        json.dump({"db": "bact10", "scale": E2E_SCALE, "k": K, "target": hot, "log2_slots": 22}, fh)
    shutil.rmtree(cwd)
    print("e2e_vf6:", sorted(os.listdir(os.path.join(outdir, "plain"))), sorted(os.listdir(os.path.join(outdir, "target"))))


def make_e2e_m3():
    """kmer_read_m3 with its 16-probe cap on a table that is ~87 % full (m3_ref_tiny: 2^16 cells)"""
    rng = np.random.default_rng(12)
    parent, cnt = synth.load_taxonomy("mito")
    c = synth.scaled_counts(cnt, 2.7e-3)
    cum = synth.cumulative(c)
    keys, targets = synth.db_keys(cum, K, seed=0x317)
    assert 0.8 < keys.size / 65536 < 0.95, keys.size
    cwd = tempfile.mkdtemp(prefix="m3_")
    wd = os.path.join(cwd, "wd") + "/"
    os.makedirs(wd)
    write_tree(wd + "mitochondria_tree.txt", parent)
    with open(wd + "mitochondria_data.txt", "w") as fh:
        for t in range(2, parent.size, 53):
            fh.write("%d\tNC_%06d\n" % (t, t))
        fh.write("%d\tNC_LAST\n" % (parent.size - 1))
    synth.write_probes_gz(wd + "mitochondria_probes.txt.gz", keys, targets, K)
    files = _write_inputs(wd, rng, keys, with_u=False)
    outdir = os.path.join(GOLD, "e2e_m3")
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir)
    for f in files:
        shutil.copy(wd + f, os.path.join(outdir, f))
    shutil.copy(wd + "mitochondria_data.txt", os.path.join(outdir, "mitochondria_data.txt"))
    runs = {"fqgz_fagz": ("a.fastq.gz", "b.fasta.gz"), "fa_fq": ("c.fasta", "d.fastq"), "single": ("a.fastq.gz", "none")}
    for tag, (f1, f2) in runs.items():
        out = subprocess.run([os.path.join(REF, "m3_ref_tiny"), "-wdir", wd, "-f1", wd + f1, "-f2", (wd + f2) if f2 != "none" else "none"],
                             cwd=cwd, stdout=subprocess.PIPE, check=True).stdout.decode()
        with open(os.path.join(outdir, tag + "_stdout.txt"), "w") as fh:
            fh.write(out.replace(wd, "<WD>"))
        shutil.move(wd + "result.txt", os.path.join(outdir, tag + "_result.txt"))
    with open(os.path.join(outdir, "params.json"), "w") as fh:
        json.dump({"db": "mito", "scale": 2.7e-3, "k": K, "db_seed": 0x317, "log2_slots": 16, "n_keys": int(keys.size), "runs": runs}, fh)
    shutil.rmtree(cwd)
    print("e2e_m3:", sorted(os.listdir(outdir)))


def make_e2e_fasta():
    """newkmer_10nx compiled with FASTQ = 0: <prefix>_R1.fasta per sample through process_fa"""
    rng = np.random.default_rng(13)
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, E2E_SCALE))
    keys, targets = synth.db_keys(cum, K)
    cwd = tempfile.mkdtemp(prefix="fa_")
    setup_db_dir(cwd, parent, keys, targets)
    fq = os.path.join(cwd, "fa") + "/"
    os.makedirs(fq)
    tmpd = tempfile.mkdtemp()
    _write_inputs(tmpd, rng, keys, with_u=False)
    shutil.copy(os.path.join(tmpd, "c.fasta"), fq + "X_R1.fasta")
    recs = _mixed_records(rng, keys, 150, with_u=False)
    with open(fq + "Y_R1.fasta", "w") as fh:
        fh.write("".join(">%s\n%s\n" % (a, s) for a, s, _ in recs))
    out = run_ref("nk10_ref_fasta_small", cwd, fq)
    outdir = os.path.join(GOLD, "e2e_fasta")
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir)
    for f in sorted(os.listdir(fq)):
        shutil.copy(fq + f, os.path.join(outdir, f))
    with open(os.path.join(outdir, "stdout.txt"), "w") as fh:
        fh.write(out.replace(fq, "<DIR>"))
    shutil.rmtree(cwd); shutil.rmtree(tmpd)
    print("e2e_fasta:", sorted(os.listdir(outdir)))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    what = sys.argv[1:] or ["kat", "small", "seeded", "vf6", "m3", "fasta"]
    if "fasta" in what:
        make_e2e_fasta()
    if "vf6" in what:
        make_e2e_vf6()
    if "m3" in what:
        make_e2e_m3()
    if "kat" in what:
        make_kat()
    if "small" in what:
        make_e2e_small()
    if "seeded" in what:
        make_e2e_seeded()
