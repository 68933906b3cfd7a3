#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only: needs oracle/_ref/nk10_ref, i.e. /root/reference).

Runs the UNMODIFIED reference program (oracle/_ref/nk10_ref = newkmer_10nx.cpp as shipped: 2^30 cells,
24 GiB) on a seeded bact10-synth workload, times its phases from the arrival of its own progress
lines, and stores

  tests/golden/e2e_ref_full_result.txt.gz   the _result.txt it wrote (the inputs are re-generated from seeds)
  tests/golden/e2e_ref_full.json            seeds, sizes, sha256 of _result.txt and _reads.txt, and the timings
                                            (reference 1 thread; oracle 1 thread / all cores on the same reads)

The same inputs also go through the 2^22-cell build (nk10_ref_small) that the other goldens use, and the
two outputs must be byte-identical -- the equivalence SURVEY.md 8(c) rests on, now checked by a committed script.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from kmer_id_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
REFSRC = "/root/reference"
K = 30
SCALE = 0.01       # 1 085 855... probes: the survey's "1 M-probe synthetic DB"
N_PAIRS = 200_000  # "a 200 k-pair bact10-synth run"
L = 150


def run_timed(binary, cwd, fq):
    """-> (stdout lines with arrival times, wall seconds)"""
    t0 = time.time()
    p = subprocess.Popen([binary, fq], cwd=cwd, stdout=subprocess.PIPE, bufsize=0)
    lines = []
    buf = b""
    while True:
        ch = p.stdout.read(1)
        if not ch:
            break
        if ch == b"\n":
            lines.append((time.time() - t0, buf.decode(errors="replace")))
            buf = b""
        else:
            buf += ch
    rc = p.wait()
    if rc != 0:
        raise SystemExit("%s exited with %d" % (binary, rc))
    return lines, time.time() - t0


def main():
    full = ob.ref_binary("nk10_ref")
    small = ob.ref_binary("nk10_ref_small")
    if not full or not small:
        raise SystemExit("oracle/_ref/nk10_ref is not built (make -C oracle ref; needs /root/reference)")
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, SCALE))
    keys, targets = synth.db_keys(cum, K)
    cwd = tempfile.mkdtemp(prefix="reffull_", dir="/tmp")
    os.makedirs(os.path.join(cwd, "bact10"))
    # the reference's own tree and strain list, as shipped (CRLF)
    shutil.copy(os.path.join(REFSRC, "b10", "btree_10.txt"), os.path.join(cwd, "bact10", "btree_10.txt"))
    shutil.copy(os.path.join(REFSRC, "b10", "bData10.txt"), os.path.join(cwd, "bact10", "bData10.txt"))
    synth.write_probes_gz(os.path.join(cwd, "bact10", "probes10.txt.gz"), keys, targets, K)
    results = {}
    for tag, binary in (("small", small), ("full", full)):
        fq = os.path.join(cwd, "fq_" + tag) + "/"
        os.makedirs(fq)
        r1 = synth.reads(cum, parent, N_PAIRS, L, K, r0=0)
        r2 = synth.reads(cum, parent, N_PAIRS, L, K, r0=N_PAIRS)
        q = np.full((N_PAIRS, L), ord("I"), np.uint8)  # no trimming: 121 lookups per read (roofline runs)
        synth.write_fastq_gz(fq + "ref_R1_tr.fastq.gz", r1, q, L, mate=1)
        synth.write_fastq_gz(fq + "ref_R2_tr.fastq.gz", r2, q, L, mate=2)
        lines, wall = run_timed(binary, cwd, fq)
        results[tag] = {"lines": lines, "wall": wall, "result": open(fq + "ref_result.txt", "rb").read(),
                        "reads": open(fq + "ref_reads.txt", "rb").read()}
        print(tag, "wall %.1f s" % wall, [(round(t, 1), s) for t, s in lines], flush=True)
    assert results["full"]["result"] == results["small"]["result"], "2^30 and 2^22 builds disagree on _result.txt"
    assert results["full"]["reads"] == results["small"]["reads"], "2^30 and 2^22 builds disagree on _reads.txt"

    def phases(lines):
        t = {s: tt for tt, s in lines}
        loaded = [tt for tt, s in lines if s.endswith("reads loaded")]
        kl = [tt for tt, s in lines if s.endswith("kmers loaded")]
        return {"db_loaded_s": kl[0], "r1_s": loaded[0] - kl[0], "r2_s": loaded[1] - loaded[0], "classify_s": loaded[1] - kl[0]}

    ph = phases(results["full"]["lines"])
    ph_small = phases(results["small"]["lines"])
    # the oracle on the same reads, same DB, in a 2^30-cell table of its own (24-byte cells like the reference's)
    odb = ob.OracleDB(parent.size, K, 30, parent=parent)
    odb.add(keys, targets)
    bases = np.concatenate([synth.reads(cum, parent, N_PAIRS, L, K, r0=0), synth.reads(cum, parent, N_PAIRS, L, K, r0=N_PAIRS)])
    off = synth.fixed_offsets(2 * N_PAIRS, L)
    os_ = ob.OracleSample(odb)
    sec1 = os_.classify_timed(bases, off)
    g, u = os_.counts()
    st = os_.stats()
    threads = len(os.sched_getaffinity(0))
    secn, gm, um, _ = odb.classify_mt(bases, off, threads)
    assert np.array_equal(g, gm) and np.array_equal(u, um)
    exp = "".join("%d,%d,%d\n" % (i, g[i], u[i]) for i in range(parent.size)).encode()
    assert exp == results["full"]["result"], "oracle and reference disagree"
    odb.close()
    shutil.rmtree(cwd)
    res = results["full"]["result"]
    with gzip.open(os.path.join(GOLD, "e2e_ref_full_result.txt.gz"), "wb", 9) as fh:
        fh.write(res)
    reads = 2 * N_PAIRS
    meta = {"db": "bact10", "scale": SCALE, "k": K, "n_pairs": N_PAIRS, "read_len": L, "db_seed": synth.DB_SEED,
            "read_seed": synth.READ_SEED, "quality": "I", "n_keys": int(keys.size), "lookups": st["lookups"],
            "result_sha256": hashlib.sha256(res).hexdigest(),
            "reads_sha256": hashlib.sha256(results["full"]["reads"]).hexdigest(),
            "binary": "oracle/_ref/nk10_ref = newkmer_10nx.cpp unmodified (MAXHASH = 2^30), g++ -O3",
            "equal_to_2pow22_build": True,
            "timing": {"host": "build container, %d vCPU" % threads,
                       "reference_2pow30": dict(ph, wall_s=results["full"]["wall"],
                                                pairs_per_s=N_PAIRS / ph["classify_s"], reads_per_s=reads / ph["classify_s"],
                                                lookups_per_s=st["lookups"] / ph["classify_s"]),
                       "reference_2pow22": dict(ph_small, wall_s=results["small"]["wall"],
                                                pairs_per_s=N_PAIRS / ph_small["classify_s"]),
                       "oracle_1_thread": {"classify_s": sec1, "pairs_per_s": N_PAIRS / sec1, "lookups_per_s": st["lookups"] / sec1},
                       "oracle_all_cores": {"threads": threads, "classify_s": secn, "pairs_per_s": N_PAIRS / secn,
                                            "lookups_per_s": st["lookups"] / secn}}}
    with open(os.path.join(GOLD, "e2e_ref_full.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print(json.dumps(meta["timing"], indent=1))


if __name__ == "__main__":
    main()
