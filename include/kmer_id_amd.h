/*
 * kmer_id_amd.h -- C ABI of libkmer_id_amd.so, the MI355X (gfx950) k-mer read
 * classifier.  This is the drop-in boundary for the hot path of
 * mmammel8/kmer_id: everything `process_read` (newkmer_10nx.cpp:452-617) does
 * with the global hash table `ht` (:158-266), the global taxonomy `taxonomy`
 * (:93-156) and the global counters `gcount/ucount/kmer_seen` (:61-64).
 *
 * The reference has no FFI of its own (it is one translation unit with global
 * state), so each entry point below names the reference code it replaces.
 * INTEGRATION.md shows the ~40-line patch that makes newkmer_10nx.cpp call
 * these instead of its own loop.
 *
 * Conventions: plain C types only; every function returns KID_OK (0) or a
 * negative kid_status; no exception crosses the boundary; the caller owns every
 * buffer it passes; the library owns the handles until *_destroy.  There is NO
 * CPU fallback: without a HIP device every compute entry point returns
 * KID_ERR_NO_DEVICE.
 */
#ifndef KMER_ID_AMD_H
#define KMER_ID_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum kid_status {
    KID_OK = 0,
    KID_ERR_ARG = -1,        /* bad argument (null pointer, size, offsets not monotone, start/stop outside read) */
    KID_ERR_NOMEM = -2,      /* host or device allocation failed */
    KID_ERR_HIP = -3,        /* a HIP runtime call failed; see kid_last_error() */
    KID_ERR_TABLE_FULL = -4, /* more than 2^log2_slots - 32 entries: the reference prints
                                "out of memory in table" and exit(1)s, newkmer_10nx.cpp:256-260 */
    KID_ERR_TREE = -5,       /* parent[] has an out-of-range entry or a cycle (the reference would loop forever in msca) */
    KID_ERR_NO_DEVICE = -6,  /* no usable HIP device */
    KID_ERR_TARGET = -7,     /* a target id >= ntar (the reference would index gcount[] out of bounds) */
    KID_ERR_IO = -8,         /* file could not be opened / gz error (reference: exit(255), newkmer_10nx.cpp:87-91) */
    KID_ERR_FORMAT = -9,     /* input line >= 16 KiB (exit 255, :773) or qual shorter than seq (std::out_of_range, :727) */
    KID_ERR_STATE = -10      /* call sequence error (e.g. classify after sample_end without reset) */
} kid_status;

/* option bits for kid_db_build*() */
#define KID_FLAG_U_IS_T 1u    /* U/u is read as T: kmer_read_vf6.cpp:496-500,521-525 */
#define KID_FLAG_HOST_BUILD 2u /* build the table on the host with the reference's sequential insert
                                  order (exact cell geometry).  Implied when max_probes > 0. */
#define KID_FLAG_REF_GEOMETRY 4u /* GPU build, but with the reference's cell placement (fmix64 +
                                  triangular probing) instead of the minimizer-localised one.  Lookup
                                  results are the same either way; only speed differs. */

typedef struct kid_db kid_db;         /* hash table + taxonomy, resident in one GPU's HBM */
typedef struct kid_sample kid_sample; /* per-sample counters: gcount, seen-bitmap (-> ucount) */

typedef struct kid_db_info {
    int32_t ntar;        /* number of taxonomy nodes (MAXTAR, newkmer_10nx.cpp:45) */
    int32_t k;           /* k-mer length (KSIZE, :43) */
    int32_t log2_slots;  /* log2 of the table size (MAXHASH, :49) */
    int32_t max_probes;  /* 0 = unbounded (10nx, vf6); 16 = kmer_read_m3.cpp:42,232 */
    uint32_t flags;
    int32_t device;
    int32_t tree_depth;  /* deepest node (root = 0) */
    int32_t host_built;  /* 1 if the sequential host builder produced the table */
    int32_t geometry;    /* 0 = reference placement, 1 = minimizer-localised placement */
    int32_t reserved_;
    uint64_t n_entries;  /* entries handed to the builder */
    uint64_t n_occupied; /* cells with value != 0 */
    uint64_t table_bytes;
} kid_db_info;

const char *kid_strerror(int status);
const char *kid_last_error(void); /* thread-local detail of the last failure */
int kid_device_count(int *count);

/* ---- database ------------------------------------------------------------
 * Replaces: Hashtable::Hashtable + HashClear + add_kmer (newkmer_10nx.cpp:173-180,
 * 199-202, 235-263) and Tree1::Tree1 + add_edge (:101-116).
 *   keys[i], targets[i]  the forward 2-bit keys and target ids in probes-file
 *                        order, exactly what process_kmer (:619-661) hands to
 *                        add_kmer; duplicates allowed, first one wins on lookup.
 *   parent[ntar]         Tree1::parent after all add_edge calls (default 1 = root).
 *   k                    KSIZE (1..31);  log2_slots  log2(MAXHASH) (6..32: slot indices are 32-bit)
 *   n                    at most 2^32-2 entries and at most 2^log2_slots - 32 (KID_ERR_TABLE_FULL beyond, like
 *                        the reference); the minimizer-localised placement is used while n <= 80 % of the cells
 *   max_probes           0 or the MAXREPROBE of kmer_read_m3.cpp
 *   device               HIP device ordinal
 * Table cells are 16 B {u64 key, u32 target, u32 insertion ordinal+1}.           */
int kid_db_build(const uint64_t *keys, const uint32_t *targets, uint64_t n,
                 const int32_t *parent, int32_t ntar, int k, int log2_slots,
                 int max_probes, uint32_t flags, int device, kid_db **out);
/* same, keys/targets already resident on `device` (used by the synthetic bench DB) */
int kid_db_build_device(const void *d_keys, const void *d_targets, uint64_t n,
                        const int32_t *parent, int32_t ntar, int k, int log2_slots,
                        int max_probes, uint32_t flags, int device, kid_db **out);
/* a replica of `src` in the HBM of `device` (device-to-device copies; the reference, pinned into every GPU) */
int kid_db_replicate(const kid_db *src, int device, kid_db **out);
int kid_db_get_info(const kid_db *db, kid_db_info *out);
void kid_db_destroy(kid_db *db);

/* Hashtable::getHash (newkmer_10nx.cpp:204-233) for a batch of keys, on the GPU.
 * probes (nullable) receives the number of cells each lookup read. */
int kid_db_lookup(kid_db *db, const uint64_t *keys, uint64_t n, uint32_t *targets, uint32_t *probes);
/* Hashtable::integerHash (newkmer_10nx.cpp:189-197, MurmurHash3 fmix64) for a batch of keys, computed by the
 * device code that places and finds the cells of the reference geometry. */
int kid_hash_keys(int device, const uint64_t *keys, uint64_t n, uint64_t *out);
/* Tree1::msca (newkmer_10nx.cpp:118-144) for a batch of (x,y), on the GPU. */
int kid_db_msca(kid_db *db, const int32_t *x, const int32_t *y, uint64_t n, int32_t *out);

/* ---- per-sample state ----------------------------------------------------
 * Replaces the per-sample reset in main (newkmer_10nx.cpp:1017-1019,1023).   */
int kid_sample_begin(kid_db *db, kid_sample **out);
/* options of a sample */
#define KID_OPT_INPUTS_READY 1 /* value 1: the read text handed to kid_classify_batch_device / kid_classify_fixed_device is
                                  final when the call is made and stays untouched until the batch is through (e.g. batches
                                  resident in HBM): the library may then pack a batch on a stream of its own while the batch
                                  before is still being classified, instead of strictly behind it in `stream` */
#define KID_OPT_LONG_RECORD_KMERS 2 /* value: records of more k-mers than this (default 65536; 0 = never) are classified by
                                      the long-record kernels -- every k-mer looked up by a lane of its own, one ordered
                                      fold per record -- instead of by one wavefront (FASTA contigs, kmer_read_vf6.cpp:803-861) */
int kid_sample_set_option(kid_sample *s, int option, int value);
int kid_sample_reset(kid_sample *s);
void kid_sample_destroy(kid_sample *s);

/* ---- classification (THE hot path) -----------------------------------------
 * Replaces process_read (newkmer_10nx.cpp:452-617) for n_reads reads at once.
 *   bases     ASCII read text, all reads concatenated (any bytes; only ACGTacgt,
 *             and Uu with KID_FLAG_U_IS_T, extend a k-mer: :478-525)
 *   offsets   n_reads+1 byte offsets into bases (read r = [offsets[r], offsets[r+1]))
 *   start/stop  inclusive range inside each read, as computed by process_qual
 *             (:714-760); pass NULL/NULL for whole reads (the FASTA callers, :851)
 *   out_final_targ  nullable; receives process_read's return value per read
 * Side effects on the sample: gcount[final_targ]++ per read (:613), and every
 * k-mer hit with target > 1 marks its DB entry as seen (:596-603).
 * Reads are independent; results do not depend on batch boundaries.  A batch
 * holds at most 2^31-1 reads (KID_ERR_ARG beyond; split the batch).           */
int kid_classify_batch(kid_sample *s, const uint8_t *bases, const uint64_t *offsets,
                       const int32_t *start, const int32_t *stop, uint64_t n_reads,
                       uint32_t *out_final_targ);
/* The same, asynchronous: returns once the batch has been queued -- its upload (a copy stream), its kernels (the
 * sample's stream) and the download of out_final_targ (a result stream) overlap with those of the batches before and
 * after it; up to three batches are in flight, a fourth call waits for the oldest.  The caller's buffers (inputs AND
 * out_final_targ) belong to the library until kid_classify_wait(ticket) returns.  Buffers from kid_host_alloc (pinned)
 * are transferred by DMA straight from / to the caller's memory; pageable ones work too but are staged by the HIP
 * runtime.  This replaces the reader loop of process_fqgz (newkmer_10nx.cpp:762-816) handing reads to process_read one
 * at a time.  kid_sample_end waits for everything in flight.                                                     */
int kid_classify_batch_async(kid_sample *s, const uint8_t *bases, const uint64_t *offsets,
                             const int32_t *start, const int32_t *stop, uint64_t n_reads,
                             uint32_t *out_final_targ, uint64_t *ticket);
/* fixed-length whole reads laid out back to back in host memory (no offsets array to upload) */
int kid_classify_fixed_async(kid_sample *s, const uint8_t *bases, uint32_t read_len, uint64_t n_reads,
                             uint32_t *out_final_targ, uint64_t *ticket);
/* A block of FASTQ text, asynchronous like kid_classify_batch_async: the caller has only FOUND the lines (the part of
 * process_fqgz, newkmer_10nx.cpp:762-816, that is inflate + splitting at '\n'); process_qual (:714-760), its
 * "stop - start >= 30" test (:757) and process_read run on the GPU for every record.
 *   text[text_nbytes]  the block as it came out of the file (< 4 GiB); recs[r] = byte offsets / lengths of the sequence
 *                      line and the quality line of record r (without '\n' / '\r')
 *   out_start/out_stop what process_qual computed; the reference hands the read to process_read iff stop - start >= k
 *                      -- records that fail the test are counted nowhere (gcount, ucount and "reads loaded" do not see
 *                      them) and get out_final_targ = 0
 * A quality line shorter than its sequence (std::out_of_range in the reference, :727) is reported as KID_ERR_FORMAT
 * by kid_sample_end / kid_sample_gcount.                                                                        */
typedef struct kid_fastq_rec {
    uint32_t seq_off, seq_len, qual_off, qual_len;
} kid_fastq_rec;
int kid_classify_fastq_async(kid_sample *s, const uint8_t *text, uint64_t text_nbytes, const kid_fastq_rec *recs,
                             uint64_t n_reads, uint32_t *out_final_targ, int32_t *out_start, int32_t *out_stop, uint64_t *ticket);
int kid_classify_wait(kid_sample *s, uint64_t ticket);
/* pinned (page-locked) host memory for the buffers above, placed on the NUMA node `device` is attached to */
int kid_host_alloc(int device, uint64_t nbytes, void **ptr);
int kid_host_free(void *ptr);
/* Device-resident form, asynchronous on `stream` (a hipStream_t, NULL = default stream).
 *   d_bases    the ASCII read text in HBM; the classify kernels read it AS IT IS (there is no packed copy): it must be
 *              16-byte aligned, its allocation must extend at least 16 bytes past bases_nbytes (= offsets[n_reads]), and it
 *              must stay untouched until the batch is through -- i.e. until everything queued on `stream` up to this
 *              call has run, not just until the call returns
 *   d_offsets, d_start, d_stop  read by a small kernel in front of the classify kernels (read descriptors, range checks);
 *              the same lifetime.  The library keeps three sets of descriptor scratch and uses them in turn; the classify
 *              kernels of a sample's batches never overlap each other (batches handed over on different streams are ordered
 *              behind each other by an event wait).  Under KID_OPT_INPUTS_READY the descriptor kernel of batch b + 1 may
 *              run while batch b is being classified.
 *   Records of more than KID_OPT_LONG_RECORD_KMERS k-mers are sorted out on the device and take the long-record kernels. */
int kid_classify_batch_device(kid_sample *s, const void *d_bases, uint64_t bases_nbytes,
                              const void *d_offsets, const void *d_start, const void *d_stop,
                              uint64_t n_reads, void *d_out_final_targ, void *stream);
/* Fixed-length reads laid out back to back (read r = bases[r*read_len, (r+1)*read_len)), whole reads, no offsets
 * array: the layout of the synthetic roofline runs.  One kernel launch per batch (no descriptors, nothing in front of
 * the classify kernel); the same alignment / lifetime rules for d_bases.                                            */
int kid_classify_fixed_device(kid_sample *s, const void *d_bases, uint32_t read_len,
                              uint64_t n_reads, void *d_out_final_targ, void *stream);

/* process_qual (newkmer_10nx.cpp:714-760) for a batch: quals laid out like bases.
 * keep[r] = 1 if the reference would call process_read (stop-start >= k).        */
int kid_trim_batch(kid_db *db, const uint8_t *quals, const uint64_t *offsets, uint64_t n_reads,
                   int32_t *start, int32_t *stop, uint8_t *keep);

/* ---- results -----------------------------------------------------------------
 * gcount[ntar], ucount[ntar] as written to <prefix>_result.txt (:1040-1043).
 * Synchronises the sample's outstanding work first.                             */
int kid_sample_end(kid_sample *s, int64_t *gcount, int64_t *ucount);
/* The same for ONE sample whose batches were dealt out over n DISTINCT kid_sample objects (a sample named twice is
 * KID_ERR_ARG: its reads would count twice), one per GPU, each on its own replica of the database (kid_db_replicate):
 * gcount summed, ucount from the union of the seen-bitmaps (copied peer to peer to samples[0]'s GPU).  Replaces the
 * per-sample reset + write of main (newkmer_10nx.cpp:1015-1045) around N devices; the result is identical for any N
 * and any way of dealing the batches.                                                                           */
int kid_sample_end_merged(kid_sample **samples, int n, int64_t *gcount, int64_t *ucount);
/* {reads, k-mer lookups, table cells read, k-mer hits} so far (synchronises) */
int kid_sample_stats(kid_sample *s, uint64_t out[4]);

/* Kernel timing for benchmarks: when enabled, every batch records a HIP event pair around its
 * kid_classify_kernel launches (on the launch stream).  kid_sample_kernel_time synchronises, adds
 * up the elapsed times since the last call and returns the number of batches they belong to.  */
int kid_sample_set_timing(kid_sample *s, int enabled);
int kid_sample_kernel_time(kid_sample *s, double *total_ms, uint64_t *launches);
/* The same interval on the device's own clock, always on: the first workgroup of kid_classify_kernel to
 * start and the last one to finish stamp the 100 MHz realtime counter.  Sum over the batches since
 * the last call (or kid_sample_reset) and their number.  No event overhead, comparable with a
 * rocprofv3 kernel trace.                                                                      */
int kid_sample_kernel_time_device(kid_sample *s, double *total_ms, uint64_t *launches);

/* ---- multi-GPU merge helpers (reads sharded over ranks, DB replicated) ---------
 * ucount is |distinct DB k-mers hit| and is not additive over shards: ranks
 * exchange slices of the "seen" bitmap, OR them, and count their slice.  The bitmap
 * has one bit per DB ENTRY: bit o = "the key whose first insert was entry o (the o-th
 * (key, target) pair handed to kid_db_build) was hit".  It does not depend on where the
 * builder placed the cells, so bitmaps of samples on different kid_db objects built
 * from the same entries (one per GPU, either placement) can be OR-ed.              */
int kid_sample_seen_bytes(const kid_sample *s, uint64_t *nbytes);
int kid_sample_seen_export(kid_sample *s, uint64_t byte_off, uint64_t nbytes, void *dst, int dst_on_device);
int kid_sample_seen_or(kid_sample *s, uint64_t byte_off, uint64_t nbytes, const void *src, int src_on_device);
int kid_sample_gcount(kid_sample *s, int64_t *gcount);
/* ucount contribution of the bitmap bits (entry ordinals) [bit_begin, bit_end): multiples of 128, at most
 * 8 * kid_sample_seen_bytes (byte ranges of the bitmap helpers above are multiples of 16) */
int kid_sample_ucount_range(kid_sample *s, uint64_t bit_begin, uint64_t bit_end, int64_t *ucount);

/* (The synthetic workload generators, the random-gather probe and the device memory helpers that bench.py and the
 *  tests use live in kmer_id_amd_bench.h: they are not part of the boundary.)                                        */
#ifdef __cplusplus
}
#endif
#include "kmer_id_amd_bench.h"
#endif /* KMER_ID_AMD_H */
