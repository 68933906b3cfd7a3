"""Host-side objects over the C ABI: KmerDB (hash table + taxonomy resident in HBM)
and Sample (per-sample gcount / seen-bitmap), mirroring the state the reference
keeps in globals (`ht`, `taxonomy`, `gcount`, `ucount`, `kmer_seen`,
newkmer_10nx.cpp:59-64,156,266).  numpy arrays in, numpy arrays out.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KidDbInfo, check


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _as(a, dtype):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def hash_keys(keys, device=0):
    """Hashtable::integerHash (fmix64) of every key, computed on the GPU."""
    lib = _lib.load()
    keys = _as(keys, np.uint64)
    out = np.empty(keys.size, np.uint64)
    check(lib.kid_hash_keys(device, _ptr(keys), keys.size, _ptr(out)))
    return out


class PinnedBuffer:
    """Page-locked host memory from the library (kid_host_alloc), viewed as a numpy array."""

    def __init__(self, nbytes, device=0):
        self._lib = _lib.load()
        p = C.c_void_p()
        check(self._lib.kid_host_alloc(device, nbytes, C.byref(p)))
        self.ptr = p.value
        self.nbytes = nbytes
        self.array = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (nbytes,))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._lib.kid_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KmerDB:
    """Replaces `new Hashtable()` + `new Tree1()` + the add_kmer/add_edge load loops."""

    def __init__(self, keys, targets, parent, k=30, log2_slots=30, max_probes=0, flags=0, device=0):
        lib = _lib.load()
        keys = _as(keys, np.uint64)
        targets = _as(targets, np.uint32)
        parent = _as(parent, np.int32)
        if keys.shape != targets.shape or keys.ndim != 1:
            raise ValueError("keys and targets must be 1-D arrays of equal length")
        h = C.c_void_p()
        check(lib.kid_db_build(_ptr(keys), _ptr(targets), keys.size, _ptr(parent), parent.size, k, log2_slots,
                               max_probes, flags, device, C.byref(h)))
        self._h = h
        self._lib = lib

    @classmethod
    def from_device(cls, d_keys, d_targets, n, parent, k=30, log2_slots=30, max_probes=0, flags=0, device=0):
        """keys/targets already in HBM (raw device pointers, e.g. torch .data_ptr())."""
        lib = _lib.load()
        parent = _as(parent, np.int32)
        self = cls.__new__(cls)
        h = C.c_void_p()
        check(lib.kid_db_build_device(C.c_void_p(d_keys), C.c_void_p(d_targets), n, _ptr(parent), parent.size, k,
                                      log2_slots, max_probes, flags, device, C.byref(h)))
        self._h = h
        self._lib = lib
        return self

    def replicate(self, device):
        """A replica of this database in the HBM of `device` (device-to-device copies)."""
        other = KmerDB.__new__(KmerDB)
        h = C.c_void_p()
        check(self._lib.kid_db_replicate(self._h, device, C.byref(h)))
        other._h = h
        other._lib = self._lib
        return other

    @property
    def info(self):
        out = KidDbInfo()
        check(self._lib.kid_db_get_info(self._h, C.byref(out)))
        return out

    def lookup(self, keys, with_probes=False):
        """Hashtable::getHash for a batch of keys (newkmer_10nx.cpp:204-233)."""
        keys = _as(keys, np.uint64)
        targets = np.empty(keys.size, np.uint32)
        probes = np.empty(keys.size, np.uint32) if with_probes else None
        check(self._lib.kid_db_lookup(self._h, _ptr(keys), keys.size, _ptr(targets), _ptr(probes)))
        return (targets, probes) if with_probes else targets

    def msca(self, x, y):
        """Tree1::msca for a batch of pairs (newkmer_10nx.cpp:118-144)."""
        x = _as(x, np.int32)
        y = _as(y, np.int32)
        out = np.empty(x.size, np.int32)
        check(self._lib.kid_db_msca(self._h, _ptr(x), _ptr(y), x.size, _ptr(out)))
        return out

    def trim(self, quals, offsets):
        """process_qual for a batch (newkmer_10nx.cpp:714-760) -> (start, stop, keep)."""
        quals = _as(quals, np.uint8)
        offsets = _as(offsets, np.uint64)
        n = offsets.size - 1
        start = np.empty(n, np.int32)
        stop = np.empty(n, np.int32)
        keep = np.empty(n, np.uint8)
        check(self._lib.kid_trim_batch(self._h, _ptr(quals), _ptr(offsets), n, _ptr(start), _ptr(stop), _ptr(keep)))
        return start, stop, keep

    def gather_ceiling(self, n_loads=1 << 28, inflight=4, iters=3):
        """Random gather rate over this DB's table: (ms per launch, loads per launch).  inflight 101 / 108: random
        128-byte lines asked for the way the classify kernel asks (64 per load / runs of 8 lanes), 4 loads in flight;
        1, 2, 4, 8: the round-1 cell probe (see include/kmer_id_amd.h)."""
        ms = C.c_float(0)
        loads = C.c_uint64(0)
        check(self._lib.kid_bench_gather(self._h, n_loads, inflight, iters, C.byref(ms), C.byref(loads)))
        return ms.value, loads.value

    def sample(self):
        return Sample(self)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kid_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def end_merged(samples):
    """One sample of the input dealt out over several Sample objects (one per GPU / replica) -> (gcount, ucount)."""
    lib = _lib.load()
    n = len(samples)
    arr = (C.c_void_p * n)(*[s._h for s in samples])
    ntar = samples[0].ntar
    g = np.empty(ntar, np.int64)
    u = np.empty(ntar, np.int64)
    check(lib.kid_sample_end_merged(arr, n, _ptr(g), _ptr(u)))
    return g, u


class Sample:
    """Per-sample counters; replaces the reset at newkmer_10nx.cpp:1017-1019."""

    def __init__(self, db):
        self.db = db
        self._lib = db._lib
        h = C.c_void_p()
        check(self._lib.kid_sample_begin(db._h, C.byref(h)))
        self._h = h
        self.ntar = db.info.ntar

    def reset(self):
        check(self._lib.kid_sample_reset(self._h))

    def set_option(self, option, value=1):
        check(self._lib.kid_sample_set_option(self._h, option, value))

    def classify(self, bases, offsets, start=None, stop=None, want_final=True):
        """process_read for a batch held in host memory; returns final_targ per read."""
        bases = _as(bases, np.uint8)
        offsets = _as(offsets, np.uint64)
        n = offsets.size - 1
        start = _as(start, np.int32)
        stop = _as(stop, np.int32)
        out = np.empty(n, np.uint32) if want_final else None
        check(self._lib.kid_classify_batch(self._h, _ptr(bases), _ptr(offsets), _ptr(start), _ptr(stop), n, _ptr(out)))
        return out

    @staticmethod
    def _borrowed(a, dtype, name):
        """an array the library reads / writes in place until wait(): it must already be what the C ABI expects"""
        if a is None:
            return None
        if not isinstance(a, np.ndarray) or a.dtype != np.dtype(dtype) or not a.flags["C_CONTIGUOUS"]:
            raise TypeError("%s must be a C-contiguous numpy array of %s (it is handed to the library as it is, not copied)" % (name, np.dtype(dtype)))
        return a

    def classify_async(self, bases, offsets, start=None, stop=None, out=None):
        """Queue a batch held in host memory (numpy arrays or PinnedBuffer views, which the caller keeps alive and
        untouched until wait(ticket)); -> ticket.  `out`: uint32[n] array that receives final_targ (or None)."""
        bases = self._borrowed(bases, np.uint8, "bases")
        offsets = self._borrowed(offsets, np.uint64, "offsets")
        start = self._borrowed(start, np.int32, "start")
        stop = self._borrowed(stop, np.int32, "stop")
        out = self._borrowed(out, np.uint32, "out")
        n = offsets.size - 1
        if out is not None and out.size < n:
            raise ValueError("out holds fewer than %d entries" % n)
        t = C.c_uint64(0)
        check(self._lib.kid_classify_batch_async(self._h, _ptr(bases), _ptr(offsets), _ptr(start), _ptr(stop), n, _ptr(out), C.byref(t)))
        return t.value

    def classify_fastq(self, text, recs):
        """A block of FASTQ text with its line index (uint32[n, 4]: seq_off, seq_len, qual_off, qual_len): process_qual,
        the >= k test and process_read on the GPU (kid_classify_fastq_async).  -> (final_targ, start, stop)"""
        text = _as(np.frombuffer(text, np.uint8) if isinstance(text, (bytes, bytearray)) else text, np.uint8)
        recs = _as(recs, np.uint32).reshape(-1, 4)
        n = recs.shape[0]
        final = np.empty(n, np.uint32)
        start = np.empty(n, np.int32)
        stop = np.empty(n, np.int32)
        t = C.c_uint64(0)
        check(self._lib.kid_classify_fastq_async(self._h, _ptr(text), text.size, _ptr(recs), n, _ptr(final), _ptr(start), _ptr(stop), C.byref(t)))
        self.wait(t.value)
        return final, start, stop

    def classify_fixed_async(self, bases_ptr, read_len, n_reads, out_ptr=0):
        """fixed-length whole reads back to back at the raw host address bases_ptr; -> ticket"""
        t = C.c_uint64(0)
        check(self._lib.kid_classify_fixed_async(self._h, C.c_void_p(bases_ptr), read_len, n_reads, C.c_void_p(out_ptr or None), C.byref(t)))
        return t.value

    def wait(self, ticket):
        check(self._lib.kid_classify_wait(self._h, ticket))

    def classify_device(self, d_bases, bases_nbytes, d_offsets, n_reads, d_start=0, d_stop=0, d_out=0, stream=0):
        """Asynchronous, device-resident inputs (raw pointers)."""
        check(self._lib.kid_classify_batch_device(self._h, C.c_void_p(d_bases), bases_nbytes, C.c_void_p(d_offsets),
                                                  C.c_void_p(d_start or None), C.c_void_p(d_stop or None), n_reads,
                                                  C.c_void_p(d_out or None), C.c_void_p(stream or None)))

    def classify_fixed_device(self, d_bases, read_len, n_reads, d_out=0, stream=0):
        check(self._lib.kid_classify_fixed_device(self._h, C.c_void_p(d_bases), read_len, n_reads,
                                                  C.c_void_p(d_out or None), C.c_void_p(stream or None)))

    def end(self):
        """-> (gcount[ntar], ucount[ntar]) as written to <prefix>_result.txt."""
        g = np.empty(self.ntar, np.int64)
        u = np.empty(self.ntar, np.int64)
        check(self._lib.kid_sample_end(self._h, _ptr(g), _ptr(u)))
        return g, u

    def gcount(self):
        g = np.empty(self.ntar, np.int64)
        check(self._lib.kid_sample_gcount(self._h, _ptr(g)))
        return g

    def ucount_range(self, slot_begin, slot_end):
        u = np.empty(self.ntar, np.int64)
        check(self._lib.kid_sample_ucount_range(self._h, slot_begin, slot_end, _ptr(u)))
        return u

    def stats(self):
        out = np.zeros(4, np.uint64)
        check(self._lib.kid_sample_stats(self._h, _ptr(out)))
        return {"reads": int(out[0]), "lookups": int(out[1]), "probes": int(out[2]), "hits": int(out[3])}

    def set_timing(self, enabled=True):
        check(self._lib.kid_sample_set_timing(self._h, 1 if enabled else 0))

    def kernel_time(self):
        """-> (total ms, launches) of kid_classify_kernel since the last call (HIP events on the launch stream)"""
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self._lib.kid_sample_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def kernel_time_device(self):
        """-> (total ms, batches) of kid_classify_kernel since the last call, on the device's own 100 MHz clock"""
        ms, n = C.c_double(0), C.c_uint64(0)
        check(self._lib.kid_sample_kernel_time_device(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def seen_bytes(self):
        n = C.c_uint64(0)
        check(self._lib.kid_sample_seen_bytes(self._h, C.byref(n)))
        return n.value

    def seen_export(self, byte_off, nbytes, dst_ptr=None, on_device=False):
        if dst_ptr is None:
            buf = np.empty(nbytes, np.uint8)
            check(self._lib.kid_sample_seen_export(self._h, byte_off, nbytes, _ptr(buf), 0))
            return buf
        check(self._lib.kid_sample_seen_export(self._h, byte_off, nbytes, C.c_void_p(dst_ptr), 1 if on_device else 0))
        return None  # dst_ptr may be a host pointer too (on_device=False)

    def seen_or(self, byte_off, src, nbytes=None, on_device=False):
        if isinstance(src, np.ndarray):
            src = np.ascontiguousarray(src, np.uint8)
            check(self._lib.kid_sample_seen_or(self._h, byte_off, src.size, _ptr(src), 0))
        else:
            check(self._lib.kid_sample_seen_or(self._h, byte_off, nbytes, C.c_void_p(src), 1 if on_device else 0))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.kid_sample_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
