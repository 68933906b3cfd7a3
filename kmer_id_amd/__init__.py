"""kmer_id_amd -- MI355X-native k-mer read classifier (drop-in for the hot path of
mmammel8/kmer_id's newkmer_10nx.cpp).  The product is the C-ABI library
libkmer_id_amd.so (include/kmer_id_amd.h) and the nk10-compatible CLI; this
package is the thin Python host layer used by the tests, the benchmark and the
multi-GPU launcher.
"""
from ._lib import KidError, KID_FLAG_HOST_BUILD, KID_FLAG_REF_GEOMETRY, KID_FLAG_U_IS_T, KID_OPT_INPUTS_READY, device_count, load  # noqa: F401
from .api import KmerDB, PinnedBuffer, Sample, end_merged, hash_keys  # noqa: F401

__all__ = ["KmerDB", "Sample", "PinnedBuffer", "hash_keys", "end_merged", "KidError", "device_count", "load", "KID_FLAG_U_IS_T", "KID_FLAG_HOST_BUILD", "KID_FLAG_REF_GEOMETRY", "KID_OPT_INPUTS_READY"]
