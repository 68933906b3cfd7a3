"""The rank path of bench.py --gpus N on one GPU: two fresh processes (started before anything in them touches the GPU),
one rank each, both on GPU 0, rendezvous over gloo on 127.0.0.1.  Every rank builds its OWN table from the same
entries (the builder's races place cells differently on every build; the seen-bitmap goes by entry ordinal, so the
bitmaps still merge), classifies its own shard of the reads and calls kmer_id_amd.dist.merge_sample; the merged
counters must equal what one table gives for all the reads -- also when the database lists keys twice (first insert
wins, newkmer_10nx.cpp:235-263).  SURVEY 8(e); ucount is not additive over shards (:596-603)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import json, os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
import kmer_id_amd
from kmer_id_amd import KmerDB, synth
from kmer_id_amd.dist import merge_sample
from helpers import K, small_db
parent, cum, keys, targets = small_db(1e-3)
# duplicate keys with other targets behind the originals: the first insert must win on every rank's table
dup = np.arange(0, keys.size, 7)
keys = np.concatenate([keys, keys[dup]])
targets = np.concatenate([targets, np.roll(targets[dup], 1)])
n_per, L = 30000, 150
db = KmerDB(keys, targets, parent, k=K, log2_slots=20, device=0)
s = db.sample()
bases = synth.reads(cum, parent, n_per, L, K, r0=rank * n_per)
s.classify(bases, synth.fixed_offsets(n_per, L), want_final=False)
own_g = s.gcount()
g, u = merge_sample(s, "cpu", force_collectives=True)
out = {"rank": rank, "gsum_own": int(own_g.sum()), "g": g.tolist(), "u": u.tolist()}
if rank == 0:   # one table, all the reads
    chk = db.sample()
    allb = synth.reads(cum, parent, world * n_per, L, K, r0=0)
    chk.classify(allb, synth.fixed_offsets(world * n_per, L), want_final=False)
    g1, u1 = chk.end()
    out["g1"] = g1.tolist(); out["u1"] = u1.tolist()
    chk.close()
dist.barrier()
json.dump(out, open(os.environ["KID_TEST_OUT"] + ".%%d" %% rank, "w"))
s.close(); db.close()
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_two_ranks_one_gpu_merge_equals_single_table(tmp_path):
    world = 2
    port = _free_port()
    out = str(tmp_path / "res")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   KID_TEST_OUT=out, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se.decode("latin-1")[-3000:]
    res = [json.load(open(out + ".%d" % r)) for r in range(world)]
    g1, u1 = np.array(res[0]["g1"]), np.array(res[0]["u1"])
    for r in res:
        assert r["gsum_own"] == 30000
        assert np.array_equal(np.array(r["g"]), g1)       # gcount adds up over the shards
        assert np.array_equal(np.array(r["u"]), u1)       # ucount = union of the shards' seen k-mers, counted once
    assert int(g1.sum()) == world * 30000 and int(u1.sum()) > 0
