"""ctypes binding of oracle/libkmer_oracle.so -- TEST INFRASTRUCTURE.

Import this only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (kmer_id_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libkmer_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")
KO_FLAG_U_IS_T = 1

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    src = os.path.join(HERE, "kmer_oracle.c")
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    lib = C.CDLL(LIB)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    sig = {
        "ko_fmix64": (u64, [u64]),
        "ko_db_new": (vp, [i32, i32, i32, i32, C.c_uint32]),
        "ko_db_free": (None, [vp]),
        "ko_db_size": (u64, [vp]),
        "ko_db_add_edge": (i32, [vp, i32, i32]),
        "ko_msca": (i32, [vp, i32, i32]),
        "ko_msca_checksum": (u64, [vp]),
        "ko_db_add_kmer": (i32, [vp, u64, C.c_uint32]),
        "ko_db_add_batch": (i32, [vp, vp, vp, u64]),
        "ko_db_get": (C.c_uint32, [vp, u64, vp]),
        "ko_classify_batch_timed": (C.c_double, [vp, vp, vp, vp, vp, u64]),
        "ko_db_get_batch": (None, [vp, vp, u64, vp, vp]),
        "ko_classify_batch_mt": (C.c_double, [vp, vp, vp, vp, vp, u64, i32, vp, vp, vp]),
        "ko_db_process_kmer": (i32, [vp, C.c_char_p, C.c_size_t, C.c_uint32]),
        "ko_db_probe_line": (i32, [vp, C.c_char_p, C.c_size_t]),
        "ko_db_load_probes_gz": (C.c_longlong, [vp, C.c_char_p]),
        "ko_db_load_tree": (i32, [vp, C.c_char_p]),
        "ko_sample_new": (vp, [vp]),
        "ko_sample_free": (None, [vp]),
        "ko_sample_reset": (None, [vp]),
        "ko_sample_gcount": (vp, [vp]),
        "ko_sample_ucount": (vp, [vp]),
        "ko_sample_tct": (C.c_int64, [vp]),
        "ko_sample_stats": (None, [vp, vp]),
        "ko_process_read": (i32, [vp, C.c_char_p, i32, i32, vp]),
        "ko_classify_batch": (None, [vp, vp, vp, vp, vp, u64, vp]),
        "ko_process_qual": (i32, [C.c_char_p, i32, i32, i32, vp, vp]),
        "ko_process_fqgz": (i32, [vp, C.c_char_p, vp]),
        "ko_run_sample": (i32, [vp, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleDB:
    def __init__(self, ntar, k=30, log2_slots=20, max_probes=0, flags=0, parent=None):
        self.lib = load()
        self.h = self.lib.ko_db_new(ntar, k, log2_slots, max_probes, flags)
        if not self.h:
            raise MemoryError("ko_db_new failed")
        self.ntar, self.k = ntar, k
        if parent is not None:
            self.set_parent(parent)

    def set_parent(self, parent):
        for y, x in enumerate(np.asarray(parent).tolist()):
            if x != 1:
                self.lib.ko_db_add_edge(self.h, int(x), y)

    def add(self, keys, targets):
        keys = np.ascontiguousarray(keys, np.uint64)
        targets = np.ascontiguousarray(targets, np.uint32)
        if self.lib.ko_db_add_batch(self.h, _p(keys), _p(targets), keys.size) != 0:
            raise RuntimeError("out of memory in table")

    def get(self, keys, with_probes=False):
        keys = np.ascontiguousarray(keys, np.uint64)
        t = np.empty(keys.size, np.uint32)
        p = np.empty(keys.size, np.uint32)
        self.lib.ko_db_get_batch(self.h, _p(keys), keys.size, _p(t), _p(p))
        return (t, p) if with_probes else t

    def msca(self, x, y):
        return self.lib.ko_msca(self.h, int(x), int(y))

    def classify_mt(self, bases, offsets, threads):
        """whole reads on `threads` host threads (shared table, per-thread counters, merged)
        -> (seconds, gcount, ucount, stats)"""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = offsets.size - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        start = np.zeros(n, np.int32)
        stop = (lens - 1).astype(np.int32)
        g = np.zeros(self.ntar, np.int64)
        u = np.zeros(self.ntar, np.int64)
        st = np.zeros(3, np.uint64)
        sec = self.lib.ko_classify_batch_mt(self.h, _p(bases), _p(offsets), _p(start), _p(stop), n, int(threads), _p(g), _p(u), _p(st))
        if sec < 0:
            raise MemoryError("ko_classify_batch_mt failed")
        return sec, g, u, {"lookups": int(st[0]), "probes": int(st[1]), "hits": int(st[2])}

    def load_probes_gz(self, path):
        return self.lib.ko_db_load_probes_gz(self.h, path.encode())

    def load_tree(self, path):
        return self.lib.ko_db_load_tree(self.h, path.encode())

    def close(self):
        if self.h:
            self.lib.ko_db_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleSample:
    def __init__(self, db):
        self.db = db
        self.lib = db.lib
        self.h = self.lib.ko_sample_new(db.h)

    def reset(self):
        self.lib.ko_sample_reset(self.h)

    def classify(self, bases, offsets, start=None, stop=None):
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = offsets.size - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        start = np.zeros(n, np.int32) if start is None else np.ascontiguousarray(start, np.int32)
        stop = (lens - 1).astype(np.int32) if stop is None else np.ascontiguousarray(stop, np.int32)
        out = np.empty(n, np.uint32)
        self.lib.ko_classify_batch(self.h, _p(bases), _p(offsets), _p(start), _p(stop), n, _p(out))
        return out

    def classify_timed(self, bases, offsets):
        """whole reads, single thread; -> seconds"""
        bases = np.ascontiguousarray(bases, np.uint8)
        offsets = np.ascontiguousarray(offsets, np.uint64)
        n = offsets.size - 1
        lens = (offsets[1:] - offsets[:-1]).astype(np.int64)
        start = np.zeros(n, np.int32)
        stop = (lens - 1).astype(np.int32)
        return self.lib.ko_classify_batch_timed(self.h, _p(bases), _p(offsets), _p(start), _p(stop), n)

    def counts(self):
        n = self.db.ntar
        g = np.ctypeslib.as_array(C.cast(self.lib.ko_sample_gcount(self.h), C.POINTER(C.c_int64)), (n,)).copy()
        u = np.ctypeslib.as_array(C.cast(self.lib.ko_sample_ucount(self.h), C.POINTER(C.c_int64)), (n,)).copy()
        return g, u

    def stats(self):
        out = np.zeros(3, np.uint64)
        self.lib.ko_sample_stats(self.h, _p(out))
        return {"lookups": int(out[0]), "probes": int(out[1]), "hits": int(out[2])}

    def run_sample(self, directory, prefix, e1="_R1_tr.fastq.gz", e2="_R2_tr.fastq.gz"):
        return self.lib.ko_run_sample(self.h, directory.encode(), prefix.encode(), e1.encode(), e2.encode())

    def close(self):
        if self.h:
            self.lib.ko_sample_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def process_qual(qual, seqlen, k=30):
    """-> (called, start, stop); called = -1 where the reference would throw."""
    lib = load()
    st, sp = C.c_int(0), C.c_int(0)
    q = bytes(qual)
    r = lib.ko_process_qual(q, seqlen, len(q), k, C.byref(st), C.byref(sp))
    return r, st.value, sp.value


def ref_binary(name):
    """Path of a compiled-reference binary under oracle/_ref (None if not built)."""
    p = os.path.join(REF_DIR, name)
    return p if os.path.exists(p) else None
