/* kmer_id_amd_bench.h -- what bench.py, the tests and tools/ need beside the boundary (kmer_id_amd.h): the seeded
 * synthetic workload (DESIGN.md "synthetic workload"), the random-line probe the kernel is priced against, and plain
 * device memory helpers for host languages without a HIP binding.  Exported by the same library.                    */
#ifndef KMER_ID_AMD_BENCH_H
#define KMER_ID_AMD_BENCH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
struct kid_db;
/* ---- synthetic workload generators (bench + tests; deterministic, seeded) ------
 * DB key j = canonical(splitmix64(seed + j) mod 4^k); target of key j follows
 * cum[] (cum[t] <= j < cum[t+1]).  Reads: see DESIGN.md "synthetic workload".    */
int kid_synth_db_keys_host(uint64_t seed, int k, const uint64_t *cum, int32_t ntar,
                           uint64_t j0, uint64_t n, uint64_t *keys, uint32_t *targets);
int kid_synth_db_keys_device(uint64_t seed, int k, const uint64_t *cum_host, int32_t ntar,
                             uint64_t j0, uint64_t n, void *d_keys, void *d_targets, int device);
int kid_synth_reads_host(uint64_t db_seed, uint64_t read_seed, int k, const uint64_t *cum,
                         const int32_t *parent, int32_t ntar, uint64_t r0, uint64_t n_reads,
                         uint32_t read_len, uint8_t *bases);
int kid_synth_reads_device(uint64_t db_seed, uint64_t read_seed, int k, const uint64_t *cum_host,
                           const int32_t *parent_host, int32_t ntar, uint64_t r0, uint64_t n_reads,
                           uint32_t read_len, void *d_bases, int device);

/* random gather micro-benchmark over the DB's own table: the measured ceiling the
 * lookup kernel is priced against.  inflight = 101 / 108: random 128-byte LINES,
 * one 16-byte load per lane as the classify kernel issues it (64 distinct lines per
 * load / runs of 8 lanes on a line), four loads in flight per lane; *loads_out = the
 * distinct line requests of one launch.  inflight = 1,2,4,8: n_loads random cells,
 * that many independent loads per lane, of which the compiler keeps two 4-byte
 * loads per cell (the round-1 probe: twice the load instructions per line, it tops
 * out at 39 G cells/s where the line probe reaches 49 G lines/s).
 * *ms_out = milliseconds per launch.                                               */
int kid_bench_gather(struct kid_db *db, uint64_t n_loads, int inflight, int iters, float *ms_out, uint64_t *loads_out);

/* device memory helpers so that a host language without a HIP binding can stage buffers */
int kid_dev_alloc(int device, uint64_t nbytes, void **d_ptr);
int kid_dev_free(int device, void *d_ptr);
int kid_dev_upload(int device, void *d_dst, const void *src, uint64_t nbytes);
int kid_dev_download(int device, void *dst, const void *d_src, uint64_t nbytes);
int kid_dev_sync(int device);

#ifdef __cplusplus
}
#endif
#endif /* KMER_ID_AMD_BENCH_H */
