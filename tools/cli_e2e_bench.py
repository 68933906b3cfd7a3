#!/usr/bin/env python3
"""End-to-end rate of the nk10 program (FASTQ.gz directory in -> _result.txt out) on a GPU box:
synthetic bact10 DB at a small scale (the text DB load is not what is measured), S samples of P
pairs, reader-thread counts 1/2/4/8.  Prints reads/s per configuration (wall clock of the whole
process minus the DB load measured on an empty directory)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from kmer_id_amd import _build, synth  # noqa: E402

S, P, L, K = int(os.environ.get("S", 4)), int(os.environ.get("P", 250000)), 150, 30
nk10 = _build.build_cli()
cwd = tempfile.mkdtemp(prefix="e2e_")
parent, cnt = synth.load_taxonomy("bact10")
cum = synth.cumulative(synth.scaled_counts(cnt, 0.01))
keys, targets = synth.db_keys(cum, K)
os.makedirs(os.path.join(cwd, "bact10"))
with open(os.path.join(cwd, "bact10", "btree_10.txt"), "w") as fh:
    for y, x in enumerate(parent.tolist()):
        if y >= 2 and x != 1:
            fh.write("%d\t%d\n" % (x, y))
open(os.path.join(cwd, "bact10", "bData10.txt"), "w").write("4\tX\n")
t0 = time.time()
synth.write_probes_gz(os.path.join(cwd, "bact10", "probes10.txt.gz"), keys, targets, K)
fq = os.path.join(cwd, "fq") + "/"
empty = os.path.join(cwd, "empty") + "/"
os.makedirs(fq); os.makedirs(empty)
for s in range(S):
    for mate in (1, 2):
        r0 = (2 * s + mate - 1) * P
        synth.write_fastq_gz(fq + "S%d_R%d_tr.fastq.gz" % (s, mate), synth.reads(cum, parent, P, L, K, r0=r0), synth.qualities(P, L, r0=r0), L, mate=mate)
print("inputs generated in %.0f s: %d samples x %d pairs, %.0f MB gz" % (time.time() - t0, S, P, sum(os.path.getsize(fq + f) for f in os.listdir(fq)) / 1e6), flush=True)
cache = os.path.join(cwd, "db.kidx")
def run(d, threads):
    t = time.time()
    subprocess.run([nk10, d, "--log2-slots", "24", "--db-cache", cache, "--threads", str(threads)], cwd=cwd, check=True, stdout=subprocess.DEVNULL)
    return time.time() - t
run(empty, 1)              # writes the cache
base = min(run(empty, 1) for _ in range(2))
print("startup (DB from cache + table build + GPU init): %.2f s" % base)
for threads in (1, 2, 4, 8):
    w = run(fq, threads)
    print("threads %d: %.2f s wall, %.2f s net -> %.2f M reads/s (%.2f M pairs/s)" % (threads, w, w - base, 2 * S * P / (w - base) / 1e6, S * P / (w - base) / 1e6), flush=True)
