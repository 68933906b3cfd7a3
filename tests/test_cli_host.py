"""Host stages of the nk10 program (database text loaders, FASTQ reader, trimming) on the
CPU via `nk10 --dry-run`, against the oracle and the reference's golden run; and, on a GPU,
the whole program against the files the compiled reference wrote."""
import filecmp
import gzip
import hashlib
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import K, ob, synth
from kmer_id_amd import _build


def write_tree(path, parent, eol="\r\n"):
    with open(path, "w", newline="") as fh:
        for y, x in enumerate(parent.tolist()):
            if y >= 2 and x != 1:
                fh.write("%d\t%d%s" % (x, y, eol))


def make_db_dir(cwd, scale, extra_probe_text=b""):
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, scale))
    keys, targets = synth.db_keys(cum, K)
    os.makedirs(os.path.join(cwd, "bact10"), exist_ok=True)
    write_tree(os.path.join(cwd, "bact10", "btree_10.txt"), parent)
    with open(os.path.join(cwd, "bact10", "bData10.txt"), "w") as fh:
        fh.write("4\tCP000828\r\n")
    p = os.path.join(cwd, "bact10", "probes10.txt.gz")
    synth.write_probes_gz(p, keys, targets, K)
    if extra_probe_text:
        raw = gzip.open(p).read() + extra_probe_text
        with gzip.open(p, "wb") as fh:
            fh.write(raw)
    return parent, cum, keys, targets


@pytest.fixture(scope="module")
def nk10():
    return _build.build_cli()


def parse_dry(path):
    sec, parent, probes, files = None, {}, [], {}
    hdr = {}
    for line in open(path, encoding="latin-1"):
        line = line.rstrip("\n")
        if line.startswith("PARENT "):
            sec = "parent"; hdr["ntar"] = int(line.split()[1]); continue
        if line.startswith("PROBES "):
            sec = "probes"; hdr["nkeys"] = int(line.split()[1]); hdr["lines"] = int(line.split()[2]); continue
        if line.startswith("FILE "):
            sec = line[5:]; files[sec] = []; continue
        if sec == "parent":
            a, b = line.split(); parent[int(a)] = int(b)
        elif sec == "probes":
            a, b = line.split(); probes.append((int(a), int(b)))
        else:
            acc, st, sp, seq = line.split("\t")
            files[sec].append((acc, int(st), int(sp), len(seq)))
    return hdr, parent, probes, files


def test_dry_run_host_stages(nk10, gold_dir, tmp_path, kat):
    src = os.path.join(gold_dir, "e2e_small")
    cwd = str(tmp_path)
    # the KAT probes text carries every parser quirk (CRLF, blank, garbage, long, lower case, tail)
    quirks = gzip.decompress(bytes(kat["probes_gz"]))
    parent, cum, keys, targets = make_db_dir(cwd, 2e-4, extra_probe_text=quirks)
    fq = os.path.join(cwd, "fq")
    os.makedirs(fq)
    for f in os.listdir(src):
        if f.endswith(".fastq.gz"):
            shutil.copy(os.path.join(src, f), fq)
    dump = os.path.join(cwd, "dry.txt")
    subprocess.run([nk10, fq + "/", "--dry-run", dump], cwd=cwd, check=True, stdout=subprocess.PIPE)
    hdr, par, probes, files = parse_dry(dump)
    # taxonomy
    exp_par = {i: int(p) for i, p in enumerate(parent.tolist()) if p != 1}
    assert par == exp_par and hdr["ntar"] == 5982
    # probes: the oracle's own parser must accept the same number of lines and answer the same lookups
    odb = ob.OracleDB(parent.size, K, 20, parent=parent)
    assert odb.load_probes_gz(os.path.join(cwd, "bact10", "probes10.txt.gz")) == hdr["lines"]
    assert odb.lib.ko_db_size(odb.h) == hdr["nkeys"] == len(probes)
    pk = np.array([p[0] for p in probes], np.uint64); pt = np.array([p[1] for p in probes], np.uint32)
    assert np.array_equal(pk[:keys.size], keys) and np.array_equal(pt[:keys.size], targets)
    odb2 = ob.OracleDB(parent.size, K, 20, parent=parent)
    odb2.add(pk, pt)
    q = np.concatenate([pk, kat["lookup_in"]])
    assert np.array_equal(odb.get(q), odb2.get(q))
    # reads: the same records with the same (start, stop) as the oracle's reader + process_qual
    for name, recs in files.items():
        raw = gzip.open(os.path.join(fq, name)).read()
        lines = raw.split(b"\n")[:-1]
        lines = [l[:-1] if l.endswith(b"\r") else l for l in lines]
        lines = [l for l in lines if l]
        exp = []
        for i in range(0, len(lines) - 3, 4):
            acc, seq, qual = lines[i], lines[i + 1], lines[i + 3]
            called, st, sp = ob.process_qual(qual, len(seq), K)
            assert called >= 0
            if called:
                exp.append((acc.decode("latin-1"), st, sp, len(seq)))
        assert recs == exp, name
    assert sum(len(v) for v in files.values()) > 700


def test_host_stages_with_threads_on_one_stream(nk10, tmp_path):
    """The probes file and the FASTQ files as single gzip members large enough to be inflated in pieces
    (kmer_id_amd/host/kid_pargz.cpp): what the host stages make of them must not depend on the number of threads."""
    tool = _build.build_tools()
    cwd = str(tmp_path)
    parent, cnt = synth.load_taxonomy("bact10")
    os.makedirs(os.path.join(cwd, "bact10"))
    with open(os.path.join(cwd, "counts.txt"), "w") as fh:
        fh.write("".join("%d,%d\n" % (t, c) for t, c in enumerate(cnt.tolist())))
    tree = os.path.join(cwd, "bact10", "btree_10.txt")
    with open(tree, "w") as fh:
        fh.write("".join("%d\t%d\n" % (x, y) for y, x in enumerate(parent.tolist()) if y >= 2 and x != 1))
    open(os.path.join(cwd, "bact10", "bData10.txt"), "w").write("4\tX\n")
    counts = os.path.join(cwd, "counts.txt")
    subprocess.check_call([tool, "probes", "--counts", counts, "--scale", "0.01", "--level", "6", "--out",
                           os.path.join(cwd, "bact10", "probes10.txt.gz")], stderr=subprocess.DEVNULL)
    fq = os.path.join(cwd, "fq") + "/"
    os.makedirs(fq)
    subprocess.check_call([tool, "fastq", "--counts", counts, "--tree", tree, "--scale", "0.01", "--out-dir", fq, "--samples", "1",
                           "--pairs", "60000", "--level", "6"], stderr=subprocess.DEVNULL)
    assert os.path.getsize(os.path.join(cwd, "bact10", "probes10.txt.gz")) > (8 << 20)   # several 1 MiB pieces each
    assert os.path.getsize(fq + "S0_R1_tr.fastq.gz") > (3 << 20)
    dumps = []
    for threads in (1, 8):
        dump = os.path.join(cwd, "dry%d.txt" % threads)
        subprocess.run([nk10, fq, "--dry-run", dump, "--threads", str(threads)], cwd=cwd, check=True, stdout=subprocess.PIPE)
        dumps.append(open(dump, "rb").read())
    assert len(dumps[0]) > (20 << 20) and dumps[0] == dumps[1]
    # the same file cut off: everything in front is read, then "failed gzclose" (:815), exit 255
    r1 = fq + "S0_R1_tr.fastq.gz"
    whole = open(r1, "rb").read()

    def dry(threads):
        return subprocess.run([nk10, fq, "--dry-run", os.path.join(cwd, "bad.txt"), "--threads", str(threads)], cwd=cwd,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    open(r1, "wb").write(whole[:len(whole) * 3 // 5])
    for threads in (1, 8):
        r = dry(threads)
        assert r.returncode == 255 and b"failed gzclose" in r.stderr, (threads, r.returncode, r.stderr[-300:])
    # ... and damaged in the middle: what the damage inflates to is read like any text until the stream's own checks
    # trip (exit 255, :776) -- or until a garbage record has a quality line shorter than its sequence (exit 134, as in
    # the reference).  Whichever it is, it is the same with one thread and with eight.
    open(r1, "wb").write(whole[:len(whole) // 2] + bytes(64) + whole[len(whole) // 2 + 64:])
    a, b = dry(1), dry(8)
    assert a.returncode in (134, 255) and (a.returncode, a.stderr) == (b.returncode, b.stderr), (a.returncode, b.returncode, a.stderr[-200:], b.stderr[-200:])


def test_fatal_inputs_exit_codes(nk10, tmp_path):
    cwd = str(tmp_path)
    make_db_dir(cwd, 2e-5)
    fq = os.path.join(cwd, "fq"); os.makedirs(fq)
    dump = os.path.join(cwd, "dry.txt")
    # a 16384-byte line is fatal with exit code 255 (newkmer_10nx.cpp:773), 16383 bytes is fine
    def fq_with_line(n):
        with gzip.open(os.path.join(fq, "L_R1_tr.fastq.gz"), "wb") as fh:
            fh.write(b"@a\n" + b"A" * n + b"\n+\n" + b"I" * n + b"\n")
        with gzip.open(os.path.join(fq, "L_R2_tr.fastq.gz"), "wb") as fh:
            fh.write(b"@b\nACGT\n+\nIIII\n")
        return subprocess.run([nk10, fq + "/", "--dry-run", dump], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert fq_with_line(16383).returncode == 0
    r = fq_with_line(16384)
    assert r.returncode == 255 and b"Buffer to small" in r.stderr
    # quality shorter than sequence: the reference dies in std::string::at (abort -> 134)
    with gzip.open(os.path.join(fq, "L_R1_tr.fastq.gz"), "wb") as fh:
        fh.write(b"@a\nACGTACGT\n+\nIIII\n")
    r = subprocess.run([nk10, fq + "/", "--dry-run", dump], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 134
    # missing R2 file: gzopen fails -> exit 255
    os.remove(os.path.join(fq, "L_R2_tr.fastq.gz"))
    with gzip.open(os.path.join(fq, "L_R1_tr.fastq.gz"), "wb") as fh:
        fh.write(b"@a\nACGTACGT\n+\nIIIIIIII\n")
    r = subprocess.run([nk10, fq + "/", "--dry-run", dump], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 255
    # no arguments: usage
    assert subprocess.run([nk10], stdout=subprocess.PIPE, stderr=subprocess.PIPE).returncode == 2


# ------------------------------------------------------------------ whole program on the GPU
def _run_e2e_small(nk10, gold_dir, cwd, extra, runs=1):
    src = os.path.join(gold_dir, "e2e_small")
    params = json.load(open(os.path.join(src, "params.json")))
    make_db_dir(cwd, params["scale"])
    fq = os.path.join(cwd, "fq"); os.makedirs(fq)
    for f in os.listdir(src):
        if f.endswith(".fastq.gz"):
            shutil.copy(os.path.join(src, f), fq)
    for _ in range(runs):
        for prefix in ("S1", "S2"):
            for suffix in ("_result.txt", "_reads.txt"):
                if os.path.exists(os.path.join(fq, prefix + suffix)):
                    os.remove(os.path.join(fq, prefix + suffix))
        r = subprocess.run([nk10, fq + "/", "--log2-slots", "22"] + extra, cwd=cwd, stdout=subprocess.PIPE, check=True)
        for prefix in ("S1", "S2"):
            for suffix in ("_result.txt", "_reads.txt"):
                assert filecmp.cmp(os.path.join(fq, prefix + suffix), os.path.join(src, prefix + suffix), shallow=False), prefix + suffix
        # stdout: same lines as the reference (sample order is readdir order on both sides)
        got = r.stdout.decode().replace(fq + "/", "<DIR>").splitlines()
        exp = open(os.path.join(src, "stdout.txt")).read().splitlines()
        assert sorted(got) == sorted(exp)
        assert got[:3] == exp[:3]


@pytest.mark.gpu
def test_nk10_end_to_end_small(nk10, gold_dir, tmp_path):
    _run_e2e_small(nk10, gold_dir, str(tmp_path), ["--batch-reads", "97"])


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0,0", "0,0,0"])
def test_nk10_several_devices(nk10, gold_dir, tmp_path, devices):
    """--devices: one replica of the table + one sample per device, batches dealt round-robin, counters merged when a
    sample is closed, _reads.txt in file order.  On a one-GPU box the same GPU is named several times: every replica
    and every sample is its own object, which is all the merge cares about."""
    _run_e2e_small(nk10, gold_dir, str(tmp_path), ["--batch-reads", "53", "--devices", devices])


@pytest.mark.gpu
def test_nk10_db_cache_on_gpu(nk10, gold_dir, tmp_path):
    """the binary DB cache through the HIP path: the first run parses the text files and writes the cache,
    the second one loads it; both must produce the reference's files"""
    cache = os.path.join(str(tmp_path), "db.kidx")
    _run_e2e_small(nk10, gold_dir, str(tmp_path), ["--batch-reads", "4096", "--db-cache", cache], runs=2)
    assert os.path.getsize(cache) > 0


@pytest.mark.gpu
def test_nk10_end_to_end_seeded(nk10, gold_dir, tmp_path):
    params = json.load(open(os.path.join(gold_dir, "e2e_seeded.json")))
    cwd = str(tmp_path)
    parent, cum, keys, targets = make_db_dir(cwd, params["scale"])
    fq = os.path.join(cwd, "fq"); os.makedirs(fq)
    n, L = params["n_pairs"], params["read_len"]
    synth.write_fastq_gz(os.path.join(fq, "big_R1_tr.fastq.gz"), synth.reads(cum, parent, n, L, K, r0=0), synth.qualities(n, L, r0=0), L, mate=1)
    synth.write_fastq_gz(os.path.join(fq, "big_R2_tr.fastq.gz"), synth.reads(cum, parent, n, L, K, r0=n), synth.qualities(n, L, r0=n), L, mate=2)
    subprocess.run([nk10, fq + "/", "--log2-slots", "22", "--batch-reads", "5000"], cwd=cwd, stdout=subprocess.PIPE, check=True)
    res = open(os.path.join(fq, "big_result.txt"), "rb").read()
    assert res == gzip.open(os.path.join(gold_dir, "e2e_seeded_result.txt.gz")).read()
    assert hashlib.sha256(open(os.path.join(fq, "big_reads.txt"), "rb").read()).hexdigest() == params["reads_sha256"]


@pytest.mark.gpu
def test_nk10_unreadable_directory(nk10, tmp_path):
    cwd = str(tmp_path)
    make_db_dir(cwd, 2e-5)
    r = subprocess.run([nk10, os.path.join(cwd, "nope") + "/", "--log2-slots", "16"], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 1 and b"hosed" in r.stdout


def test_db_cache_round_trip(nk10, tmp_path):
    """--db-cache: the second run loads the binary cache and hands over exactly the same database;
    touching the probes file makes the cache stale"""
    cwd = str(tmp_path)
    parent, cum, keys, targets = make_db_dir(cwd, 2e-4)
    fq = os.path.join(cwd, "fq"); os.makedirs(fq)
    with gzip.open(os.path.join(fq, "S_R1_tr.fastq.gz"), "wb") as fh:
        fh.write(b"@a\nACGT\n+\nIIII\n")
    with gzip.open(os.path.join(fq, "S_R2_tr.fastq.gz"), "wb") as fh:
        fh.write(b"@b\nACGT\n+\nIIII\n")
    cache = os.path.join(cwd, "db.kidx")
    def run(tag):
        dump = os.path.join(cwd, tag + ".txt")
        r = subprocess.run([nk10, fq + "/", "--dry-run", dump, "--db-cache", cache], cwd=cwd, check=True, stdout=subprocess.PIPE)
        return open(dump, "rb").read(), r.stdout
    first, out1 = run("first")
    assert os.path.getsize(cache) == 8 + 8 + 8 + 8 + 32 + 4 * 5982 + 12 * keys.size
    mtime = os.path.getmtime(cache)
    second, out2 = run("second")
    assert first == second and out1 == out2
    assert os.path.getmtime(cache) == mtime                    # served from the cache, not rewritten
    # a different probes file -> stale -> re-parsed and rewritten
    synth.write_probes_gz(os.path.join(cwd, "bact10", "probes10.txt.gz"), keys[:1000], targets[:1000], K)
    third, _ = run("third")
    assert third != first and b"PROBES 1000 1000" in third
    assert os.path.getsize(cache) == 64 + 4 * 5982 + 12 * 1000


@pytest.mark.gpu
def test_nk10_fasta_mode(nk10, gold_dir, tmp_path):
    """the reference's FASTQ = 0 build (plain FASTA through process_fa, one file per sample)"""
    src = os.path.join(gold_dir, "e2e_fasta")
    cwd = str(tmp_path)
    make_db_dir(cwd, 2e-4)
    fa = os.path.join(cwd, "fa"); os.makedirs(fa)
    for f in os.listdir(src):
        if f.endswith(".fasta"):
            shutil.copy(os.path.join(src, f), fa)
    r = subprocess.run([nk10, fa + "/", "--fasta", "--r1", "_R1.fasta", "--log2-slots", "22", "--batch-reads", "41"], cwd=cwd,
                       stdout=subprocess.PIPE, check=True)
    for prefix in ("X", "Y"):
        for suffix in ("_result.txt", "_reads.txt"):
            assert filecmp.cmp(os.path.join(fa, prefix + suffix), os.path.join(src, prefix + suffix), shallow=False), prefix + suffix
    assert sorted(r.stdout.decode().replace(fa + "/", "<DIR>").splitlines()) == sorted(open(os.path.join(src, "stdout.txt")).read().splitlines())
