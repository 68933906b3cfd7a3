"""The sibling front-ends: kmer_read_vf6 (job lists, FASTA/FASTQ readers, U = T, dynamic
number of targets, -target) and kmer_read_m3 (16-probe cap, single result.txt).  CPU: the host
stages (`--dry-run`) + the oracle must reproduce the files the compiled reference wrote; GPU: the
programs themselves must."""
import filecmp
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from helpers import K, ob, synth
from kmer_id_amd import _build
from test_cli_host import write_tree

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def bins():
    _build.build_cli()
    return {n: _build.cli_path(n) for n in ("kmer_read_vf6", "kmer_read_m3")}


def parse_dump(path):
    """-> parent dict, probes (keys, targets), [(label, [(acc, start, stop, seq)])]"""
    ntar, parent, probes, files, sec = 0, {}, [], [], None
    with open(path, "rb") as fh:
        for raw in fh:
            line = raw.rstrip(b"\n")
            if line.startswith(b"PARENT "):
                sec = "parent"; ntar = int(line.split()[1]); continue
            if line.startswith(b"PROBES "):
                sec = "probes"; continue
            if line.startswith(b"FILE "):
                sec = "file"; files.append((line[5:].decode(), [])); continue
            if sec == "parent":
                a, b = line.split(); parent[int(a)] = int(b)
            elif sec == "probes":
                a, b = line.split(); probes.append((int(a), int(b)))
            else:
                acc, st, sp, seq = line.split(b"\t")
                files[-1][1].append((acc, int(st), int(sp), seq))
    par = np.ones(ntar, np.int32)
    for i, p in parent.items():
        par[i] = p
    return par, np.array([p[0] for p in probes], np.uint64), np.array([p[1] for p in probes], np.uint32), files


def oracle_results(par, keys, targets, file_groups, log2_slots, max_probes=0, flags=0, save_target=0, first12=True):
    """classify each group of files as one sample with the oracle; -> [(result_text, reads_text, target_reads_text, n)]"""
    odb = ob.OracleDB(par.size, K, log2_slots, max_probes, flags, parent=par)
    odb.add(keys, targets)
    out = []
    for group in file_groups:
        s = ob.OracleSample(odb)
        reads_txt, treads_txt, n = [], [], 0
        seen = np.zeros(par.size, np.int64)
        for recs in group:
            for acc, st, sp, seq in recs:
                f = odb.lib.ko_process_read(s.h, seq, st, sp, None)
                if f > 1 and seen[f] < 12 and first12:
                    reads_txt.append(b">%d:%s\n%s\n" % (f, acc, seq[st:sp + 1]))
                if f > 1 and f == save_target:
                    treads_txt.append(b">%d:%s\n%s\n" % (f, acc, seq[st:sp + 1]))
                seen[f] += 1
                n += 1
        g, u = s.counts()
        res = "".join("%d,%d,%d\n" % (i, g[i], u[i]) for i in range(par.size)).encode()
        out.append((res, b"".join(reads_txt), b"".join(treads_txt), n))
    return out


# ------------------------------------------------------------------ vf6
def setup_vf6(cwd):
    src = os.path.join(GOLD, "e2e_vf6")
    params = json.load(open(os.path.join(src, "params.json")))
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, params["scale"]))
    keys, targets = synth.db_keys(cum, K)
    os.makedirs(os.path.join(cwd, "DB")); os.makedirs(os.path.join(cwd, "J"))
    write_tree(os.path.join(cwd, "DB", "DB_tree.txt"), parent)
    shutil.copy(os.path.join(src, "DB_data.txt"), os.path.join(cwd, "DB", "DB_data.txt"))
    synth.write_probes_gz(os.path.join(cwd, "DB", "DB_probes.txt.gz"), keys, targets, K)
    shutil.copytree(os.path.join(src, "in"), os.path.join(cwd, "in"))
    shutil.copy(os.path.join(src, "J.txt"), os.path.join(cwd, "J", "J.txt"))
    return src, params


@pytest.mark.parametrize("mode", ["plain", "target"])
def test_vf6_host_stages_and_oracle_reproduce_the_reference(bins, tmp_path, mode):
    cwd = str(tmp_path)
    src, params = setup_vf6(cwd)
    dump = os.path.join(cwd, "dry.txt")
    subprocess.run([bins["kmer_read_vf6"], "-name", "DB", "-jname", "J", "--dry-run", dump], cwd=cwd, check=True,
                   stdout=subprocess.PIPE)
    par, keys, targets, files = parse_dump(dump)
    assert par.size == 5982 and [f[0] for f in files] == ["jobA in/a.fastq.gz", "jobA in/b.fasta.gz", "jobB in/c.fasta", "jobB in/d.fastq"]
    tgt = params["target"] if mode == "target" else 0
    res = oracle_results(par, keys, targets, [[files[0][1], files[1][1]], [files[2][1], files[3][1]]], 20,
                         flags=ob.KO_FLAG_U_IS_T, save_target=tgt, first12=(tgt == 0))
    for job, (result, reads, treads, n) in zip(("jobA", "jobB"), res):
        assert result == open(os.path.join(src, mode, job + "_result.txt"), "rb").read(), job
        assert reads == open(os.path.join(src, mode, job + "_reads.txt"), "rb").read(), job
        if tgt:
            assert treads == open(os.path.join(src, mode, job + "_target_reads.txt"), "rb").read(), job
    exp_out = open(os.path.join(src, mode, "stdout.txt")).read().splitlines()
    assert "%d reads loaded" % res[0][3] in exp_out and "%d reads loaded" % res[1][3] in exp_out
    assert any(len(r[3]) > 2000 for r in files[1][1])      # the multi-line contig went through whole
    assert any(b"U" in r[3] for r in files[0][1])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,devices", [("plain", None), ("target", None), ("plain", "0,0"), ("target", "0,0,0")])
def test_vf6_end_to_end(bins, tmp_path, mode, devices):
    cwd = str(tmp_path)
    src, params = setup_vf6(cwd)
    extra = ["-target", str(params["target"])] if mode == "target" else []
    if devices:  # several replicas of the table (on this one GPU), batches dealt round-robin, counters merged
        extra += ["--devices", devices]
    r = subprocess.run([bins["kmer_read_vf6"], "-name", "DB", "-jname", "J"] + extra + ["--log2-slots", "22", "--batch-reads", "37"],
                       cwd=cwd, check=True, stdout=subprocess.PIPE)
    assert r.stdout.decode() == open(os.path.join(src, mode, "stdout.txt")).read()
    produced = sorted(f for f in os.listdir(os.path.join(cwd, "J")) if f != "J.txt")
    expected = sorted(f for f in os.listdir(os.path.join(src, mode)) if f != "stdout.txt")
    assert produced == expected
    for f in expected:
        assert filecmp.cmp(os.path.join(cwd, "J", f), os.path.join(src, mode, f), shallow=False), f


# ------------------------------------------------------------------ m3
def m3_reference_result(path, ntar):
    """kmer_read_m3.cpp:981 never initialises num_targ before taking the maximum over the strain list:
    the compiled reference started from stack garbage (21928 in the golden run) and wrote that many
    all-zero extra lines.  The defined part of its output is the first ntar lines."""
    lines = open(path, "rb").read().split(b"\n")
    assert lines[-1] == b""
    lines = lines[:-1]
    for i, l in enumerate(lines[ntar:], start=ntar):
        assert l == b"%d,0,0" % i
    return b"\n".join(lines[:ntar]) + b"\n"


def setup_m3(cwd):
    src = os.path.join(GOLD, "e2e_m3")
    params = json.load(open(os.path.join(src, "params.json")))
    parent, cnt = synth.load_taxonomy("mito")
    cum = synth.cumulative(synth.scaled_counts(cnt, params["scale"]))
    keys, targets = synth.db_keys(cum, K, seed=params["db_seed"])
    assert keys.size == params["n_keys"]
    wd = os.path.join(cwd, "wd") + "/"
    os.makedirs(wd)
    write_tree(wd + "mitochondria_tree.txt", parent, eol="\n")
    shutil.copy(os.path.join(src, "mitochondria_data.txt"), wd)
    synth.write_probes_gz(wd + "mitochondria_probes.txt.gz", keys, targets, K)
    for f in ("a.fastq.gz", "b.fasta.gz", "c.fasta", "d.fastq"):
        shutil.copy(os.path.join(src, f), wd)
    return src, params, wd


def test_m3_host_stages_and_oracle_reproduce_the_reference(bins, tmp_path):
    cwd = str(tmp_path)
    src, params, wd = setup_m3(cwd)
    for tag, (f1, f2) in params["runs"].items():
        dump = os.path.join(cwd, "dry_%s.txt" % tag)
        subprocess.run([bins["kmer_read_m3"], "-wdir", wd, "-f1", wd + f1, "-f2", (wd + f2) if f2 != "none" else "none",
                        "--dry-run", dump], cwd=cwd, check=True, stdout=subprocess.PIPE)
        par, keys, targets, files = parse_dump(dump)
        assert par.size == 17227 and len(files) == (1 if f2 == "none" else 2)
        (result, _, _, n), = oracle_results(par, keys, targets, [[f[1] for f in files]], params["log2_slots"], max_probes=16)
        assert result == m3_reference_result(os.path.join(src, tag + "_result.txt"), par.size), tag
        assert "%d reads loaded" % n in open(os.path.join(src, tag + "_stdout.txt")).read()
    # the probe cap matters on this table: some DB keys sit deeper than 16 probes and are never found
    capped = ob.OracleDB(par.size, K, params["log2_slots"], 16, 0, parent=par); capped.add(keys, targets)
    free = ob.OracleDB(par.size, K, params["log2_slots"], 0, 0, parent=par); free.add(keys, targets)
    assert (capped.get(keys) == 0).sum() > 100 and (free.get(keys) == 0).sum() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0"])
def test_m3_end_to_end(bins, tmp_path, devices):
    cwd = str(tmp_path)
    src, params, wd = setup_m3(cwd)
    for tag, (f1, f2) in params["runs"].items():
        r = subprocess.run([bins["kmer_read_m3"], "-wdir", wd, "-f1", wd + f1, "-f2", (wd + f2) if f2 != "none" else "none",
                            "--log2-slots", str(params["log2_slots"]), "--batch-reads", "53"] + (["--devices", devices] if devices else []),
                           cwd=cwd, check=True, stdout=subprocess.PIPE)
        got = r.stdout.decode().replace(wd, "<WD>").splitlines()
        exp = open(os.path.join(src, tag + "_stdout.txt")).read().splitlines()
        # line 6 is "<length of the -f1 path> : <its last character>": the path differs, the character does not
        assert [l for i, l in enumerate(got) if i != 6] == [l for i, l in enumerate(exp) if i != 6], tag
        assert got[6].split(" : ")[1] == exp[6].split(" : ")[1]
        assert open(wd + "result.txt", "rb").read() == m3_reference_result(os.path.join(src, tag + "_result.txt"), 17227), tag
