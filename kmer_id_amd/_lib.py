"""ctypes binding of include/kmer_id_amd.h (the C ABI of libkmer_id_amd.so).

There is no fallback: if the library is missing or a HIP device is absent the
calls raise.  torch is imported first (when available) so that this process uses
ONE HIP runtime: torch bundles a libamdhip64.so with the same SONAME, and the
loader reuses whichever copy is already mapped.
"""
import ctypes as C
import os

from . import _build

c_u8p = C.POINTER(C.c_uint8)
c_u32p = C.POINTER(C.c_uint32)
c_i32p = C.POINTER(C.c_int32)
c_u64p = C.POINTER(C.c_uint64)
c_i64p = C.POINTER(C.c_int64)
c_void_pp = C.POINTER(C.c_void_p)


class KidDbInfo(C.Structure):
    _fields_ = [
        ("ntar", C.c_int32), ("k", C.c_int32), ("log2_slots", C.c_int32), ("max_probes", C.c_int32),
        ("flags", C.c_uint32), ("device", C.c_int32), ("tree_depth", C.c_int32), ("host_built", C.c_int32),
        ("geometry", C.c_int32), ("reserved_", C.c_int32),
        ("n_entries", C.c_uint64), ("n_occupied", C.c_uint64), ("table_bytes", C.c_uint64),
    ]


# name -> (restype, argtypes); every symbol include/kmer_id_amd.h declares
PROTOTYPES = {
    "kid_strerror": (C.c_char_p, [C.c_int]),
    "kid_last_error": (C.c_char_p, []),
    "kid_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "kid_db_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int, C.c_int, C.c_int,
                               C.c_uint32, C.c_int, c_void_pp]),
    "kid_db_build_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_int32, C.c_int, C.c_int,
                                      C.c_int, C.c_uint32, C.c_int, c_void_pp]),
    "kid_db_replicate": (C.c_int, [C.c_void_p, C.c_int, c_void_pp]),
    "kid_db_get_info": (C.c_int, [C.c_void_p, C.POINTER(KidDbInfo)]),
    "kid_db_destroy": (None, [C.c_void_p]),
    "kid_db_lookup": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
    "kid_hash_keys": (C.c_int, [C.c_int, C.c_void_p, C.c_uint64, C.c_void_p]),
    "kid_db_msca": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "kid_sample_begin": (C.c_int, [C.c_void_p, c_void_pp]),
    "kid_sample_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "kid_sample_reset": (C.c_int, [C.c_void_p]),
    "kid_sample_destroy": (None, [C.c_void_p]),
    "kid_classify_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "kid_classify_batch_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, c_u64p]),
    "kid_classify_fixed_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, c_u64p]),
    "kid_classify_fastq_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, c_u64p]),
    "kid_classify_wait": (C.c_int, [C.c_void_p, C.c_uint64]),
    "kid_host_alloc": (C.c_int, [C.c_int, C.c_uint64, c_void_pp]),
    "kid_host_free": (C.c_int, [C.c_void_p]),
    "kid_classify_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.c_void_p, C.c_void_p]),
    "kid_classify_fixed_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]),
    "kid_trim_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "kid_sample_end": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "kid_sample_end_merged": (C.c_int, [c_void_pp, C.c_int, C.c_void_p, C.c_void_p]),
    "kid_sample_stats": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kid_sample_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "kid_sample_kernel_time": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), c_u64p]),
    "kid_sample_kernel_time_device": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), c_u64p]),
    "kid_sample_seen_bytes": (C.c_int, [C.c_void_p, c_u64p]),
    "kid_sample_seen_export": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]),
    "kid_sample_seen_or": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]),
    "kid_sample_gcount": (C.c_int, [C.c_void_p, C.c_void_p]),
    "kid_sample_ucount_range": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]),
    "kid_synth_db_keys_host": (C.c_int, [C.c_uint64, C.c_int, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p,
                                         C.c_void_p]),
    "kid_synth_db_keys_device": (C.c_int, [C.c_uint64, C.c_int, C.c_void_p, C.c_int32, C.c_uint64, C.c_uint64, C.c_void_p,
                                           C.c_void_p, C.c_int]),
    "kid_synth_reads_host": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64,
                                       C.c_uint64, C.c_uint32, C.c_void_p]),
    "kid_synth_reads_device": (C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64,
                                         C.c_uint64, C.c_uint32, C.c_void_p, C.c_int]),
    "kid_bench_gather": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_float), c_u64p]),
    "kid_dev_alloc": (C.c_int, [C.c_int, C.c_uint64, c_void_pp]),
    "kid_dev_free": (C.c_int, [C.c_int, C.c_void_p]),
    "kid_dev_upload": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kid_dev_download": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]),
    "kid_dev_sync": (C.c_int, [C.c_int]),
}

KID_OPT_INPUTS_READY = 1
KID_OPT_LONG_RECORD_KMERS = 2
KID_FLAG_U_IS_T = 1
KID_FLAG_HOST_BUILD = 2
KID_FLAG_REF_GEOMETRY = 4

_lib = None


class KidError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("kmer_id_amd: %s (status %d)" % (detail, status))
        self.status = status


def lib_path():
    # KMER_ID_AMD_LIB: an alternative build of the same library (A/B experiments)
    return os.environ.get("KMER_ID_AMD_LIB") or _build.LIB


def load(build_if_missing=True):
    """Load libkmer_id_amd.so (building it with hipcc first if it is not there)."""
    global _lib
    if _lib is not None:
        return _lib
    try:  # one HIP runtime per process: let torch map its copy first
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the library itself
        pass
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise KidError(-6, "libkmer_id_amd.so has not been built (run __graft_entry__.build())")
        _build.build_library()
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError here = the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        lib = load()
        detail = lib.kid_last_error().decode("utf-8", "replace") or lib.kid_strerror(status).decode()
        raise KidError(status, detail)


def device_count():
    lib = load()
    n = C.c_int(0)
    lib.kid_device_count(C.byref(n))
    return n.value
