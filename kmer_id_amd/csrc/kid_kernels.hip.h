// kid_kernels.hip.h -- gfx950 (MI355X, CDNA4) kernels of the k-mer read classifier.
//
// All work here is integer / indexing work bounded by random reads of the hash
// table in HBM; there is no MFMA-shaped computation on this path.
//
// Data layout in HBM
//   bases   the caller's ASCII read text; the classify kernels read it as it is and pack in registers
//   table   uint4[2^log2_slots]   {key lo, key hi, target (0 = empty), insertion ordinal+1}
//                                 one 16-byte cell = one global_load_dwordx4 per probe
//   rows    uint4[ntar]           8 x u16 per taxonomy node: e0 = depth, e[d] = ancestor of
//                                 the node at depth d (d = 1..7, the node itself at d = depth)
//   parent  int32[ntar], depth int32[ntar]   (fallback for trees deeper than 8 levels)
//   seen    u32[n_entries / 32]     one bit per DB entry (insertion ordinal), per sample (-> ucount)
//   ord_target u32[n_entries]       target of entry o (what the builder was handed)
//   gcount  u64[ntar], stats u64[8] per sample
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "kid_common.h"

#define KID_WAVE 64
#define KID_TOMBSTONE_HI 0xFFFFFFFFu // high key word of a duplicate entry that lost to an earlier insert (keys are < 2^62)
#define KID_SEG_KMERS 960  // k-mers per read segment: 960 + 30 bases + 15 alignment slack <= 64 chunks of 16 B
#define KID_WAVE_LDS_WORDS 104 // per wave, general loops: 66 packed-base words + 34 invalid-mask words + 4 counters
// pair kernel (no strips): 4 counters, the queue of unresolved lookups (3 words + 1 tag byte per entry) and the
// results of 64 reads
#define KID_CQ_CAP 192   // a read appends at most 128 entries to fewer than KID_CQ_FLUSH queued ones
#define KID_CQ_FLUSH 64
#define KID_PAIR_LDS_WORDS (4 + 3 * KID_CQ_CAP + KID_CQ_CAP / 4 + 64 + 128) // ... + the batch read numbers of 128 result slots
#define KID_TAPER 6u // the workgroups dispatched first take this many times the reads of those dispatched last
                     // (profiles/r02/ab_taper.txt: a launch ends when the slowest of its last workgroups does)
// A lookup whose header shows a fingerprint match fetches its candidate cell on the spot -- the cell sits in the
// 128-byte line the header has just brought in -- and the queue carries {target, entry ordinal} of verified hits; the
// resolver then needs neither the header nor the cell again (by then the line is long gone from every cache: one HBM
// line and two dependent round trips per hit saved).  Lookups that cannot be settled by their first candidate (a
// fingerprint false positive, a full line) are queued with their key.  Only while hits are sparse:
#define KID_EARLY_MAX 4u // ... tiles with at most this many flagged lookups
#define KID_EARLY_QN 32u // ... while fewer than this many lookups are queued (profiles/r02/clumped_ec.txt)
#define KID_GEN_ML_LDS_WORDS (104 + 3 * KID_CQ_CAP + KID_CQ_CAP / 4 + 64) // general loops on the minimizer-localised table: strip, counters, queue, results
#define KID_CLASSIFY_OCC 8 // waves per SIMD the register allocator must leave room for

typedef uint32_t kid_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t kid_u2 __attribute__((ext_vector_type(2)));

// One table cell.  Reference placement: non-temporal -- a probe touches a random 16 bytes of a 16 GiB
// table once, so the line is not worth keeping in L2 / Infinity Cache (tools/gather_policy.hip: 49 ->
// 54 G random cells/s on a 16 GiB region).  Minimizer-localised placement: plain loads -- the header
// of a line is read again by the resolver and the matching entry sits in the same 128 bytes, so the
// line should stay cached (measured: 1.29 -> 1.23 ms per launch).
// Fire-and-forget global writes of the classify kernel (per-read result, seen-bitmap bits).  Issued
// from inline assembly so that the compiler's waitcnt insertion does not know them: on gfx9 a pending
// store or no-return atomic shares vmcnt with the loads and makes the next wait a full vmcnt(0) drain,
// i.e. every read would sit out the write acknowledgement of the one before.  Nothing in the kernel
// reads these locations back; they complete before the kernel does.
__device__ __forceinline__ void kid_store_u32_nowait(uint32_t *p, uint32_t v)
{
    asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void kid_atomic_or_nowait(uint32_t *p, uint32_t v)
{
    asm volatile("global_atomic_or %0, %1, off" : : "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ uint4 kid_load_cell(const uint4 *table, uint32_t idx) { return table[idx]; }
__device__ __forceinline__ uint4 kid_load_cell_nt(const uint4 *table, uint32_t idx)
{
    const kid_u4 v = __builtin_nontemporal_load(reinterpret_cast<const kid_u4 *>(table) + idx);
    return make_uint4(v.x, v.y, v.z, v.w);
}

struct KidDevDb {
    const uint4 *table;
    uint64_t nslots;
    uint32_t slot_mask;
    uint32_t max_probes;
    int k;
    uint32_t u_is_t;
    uint32_t minloc; // 1: minimizer-localised geometry (kid_common.h), 0: the reference's fmix64 + triangular probing
    uint32_t line_shift; // minloc: 32 - log2(lines)
    uint32_t line_mask;  // minloc: lines - 1
    const uint4 *rows; // null when the tree does not fit the row encoding
    const int32_t *parent;
    const int32_t *depth;
    int32_t ntar;
};

struct KidBatch { // what the caller handed over
    const uint8_t *bases;
    const uint64_t *offsets; // null: fixed_len layout
    const int32_t *start;    // nullable
    const int32_t *stop;     // nullable
    uint32_t *out_final;     // nullable
    uint64_t n;
    uint32_t fixed_len;
};

// per-read descriptor written by kid_prepare_kernel: absolute index of the first base of the
// classified range and the number of k-mer windows in it (<= 0: none)
struct KidReadDesc {
    uint64_t first_base;
    int32_t n_kmers;
    uint32_t pad;
};

// Very long records (FASTA contigs classified whole, kmer_read_vf6.cpp:803-861).  The classify kernels give a read to ONE
// wave; a record of more than `long_cut` k-mers goes to the long-record kernels instead (kid_long_*): kid_prepare_kernel
// hides it from the classify kernels (n_kmers = 0) and puts it on this list, kid_long_plan_kernel turns the list into
// the records' places in the hit array.  All on the device: the batch may be one whose offsets the host never saw.
#define KID_LONG_MAX 1024u
struct KidLongList {
    uint32_t n;       // records flagged so far (beyond KID_LONG_MAX: left to the classify kernels)
    uint32_t pad[3];
    struct Item { uint64_t first_base; uint32_t n_kmers; uint32_t read; } e[KID_LONG_MAX];
};
struct KidLongRec {
    uint64_t first_base; // absolute index of the record's first classified base in the batch text
    uint64_t hits_off;   // where its hits start in the hits array
    uint32_t n_kmers;
    uint32_t read;       // its number in the batch
    uint64_t tile0;      // number of 256-position tiles of the records before it
};
struct KidLongPlan {
    uint32_t n_recs;
    uint32_t pad;
    uint64_t n_tiles;
    uint64_t total_kmers;
    KidLongRec recs[KID_LONG_MAX];
};

// What a launch of the classify kernels reads: the ASCII read text as the caller handed it over -- the kernels turn it
// into 2-bit codes in registers (kid_pack4 / kid_pack16), there is no packed image of the batch in memory -- and the
// reads' descriptors.  Fixed layout (kid_classify_fixed_*): no descriptors either, read r of the launch is
// bases[(read0 + r) * fixed_len ...) with fixed_nk k-mers (KidRareArgs).
struct KidInput {
    const uint8_t *bases;    // 16-byte aligned; readable up to 16 bytes past the last read
    const KidReadDesc *desc; // of this launch's reads; null: fixed layout
    uint32_t *out_final;     // of this launch's reads; nullable
    uint64_t n;              // reads of this launch
};

struct KidSampleDev {
    unsigned long long *gcount;
    uint32_t *seen;
    unsigned long long *stats; // [0] reads [1] lookups [2] probes [3] hits [4] argument errors [5] FASTQ records dropped by process_qual
                               // [6] device-clock ticks [7] launches banked [8] quality lines shorter than their sequence
                               // [29] workgroups through [30] first start [31] last end of the running launch
};

// What only rare paths of the classify kernel need (hit cells, the flush at the end): kept in device
// memory and loaded where used, so it does not occupy scalar registers across the hot loop.
struct KidRareArgs {
    unsigned long long *gcount;
    unsigned long long *stats;
    uint32_t line_mask;
    uint32_t fixed_len;           // fixed layout: bases per read (0: the launch has descriptors)
    unsigned long long batch_max; // (batch sequence number << 32) | largest n_kmers of the batch: kid_prepare_kernel
    unsigned long long read0;     // fixed layout: number of the launch's first read within the batch text
    int32_t fixed_nk;             // fixed layout: k-mers per read
    uint32_t pad;
    // what the resolver / the 64-read flush of the pair kernels need (rare paths by now, bulk work: a scalar
    // load there is cheaper than four scalar registers held across the hot loop)
    const uint4 *rows;
    uint32_t *seen;
    uint32_t *out_final;   // of the current launch: written by kid_prepare_kernel / kid_rebase_kernel
    const void *desc;      // of the current launch (KidReadDesc *; null: fixed layout)
    // the hit log (see kid_seenlog_*): the resolver appends the entry ordinals of its hits instead of setting their bits
    // in `seen` with one memory-side atomic each; null: atomics
    uint32_t *seen_log;      // KID_LOG_SHARDS regions of seen_log_cap entries
    uint32_t *seen_log_tail; // their fill counters, 64 bytes apart
    uint32_t seen_log_cap;
    uint32_t pad2;
};
#define KID_LOG_SHARDS 64u      // regions (blockIdx & 63), each with its own fill counter: a resolver chunk costs one fetch-and-add on
                                // its region's counter, and with 16 hits per read (reads from genomes the database holds) there are
                                // 600 k chunks per 1 M pairs -- on 8 counters they queued for 3 ms (profiles/r03/clumped_db_log_shards.txt)
#define KID_LOG_NONE 0xFFFFFFFFu // a log place whose lookup turned out not to be a hit
#define KID_LOG_BIN_BITS 18     // the apply pass owns the bitmap in pieces of 2^18 bits = 32 KiB of LDS
#define KID_LOG_WGS 1024u       // workgroups of the counting / scattering kernels: 16 per region

// ------------------------------------------------------------------ hash lookup
// Hashtable::getHash, newkmer_10nx.cpp:204-233 (+ probe cap kmer_read_m3.cpp:232)
// Candidate entries of a line header for fingerprint fp, as a bit set: bit (15 - d) = entry 2d,
// bit (31 - d) = entry 2d + 1 (d = header dword 0..3).  A superset of the true matches (the classic
// zero-halfword test can also flag the upper half of a dword whose lower half matched); every
// candidate is verified against the full key, so a false one only costs a load.
__device__ __forceinline__ uint32_t kid_hdr_cand(const uint4 &h, uint32_t fp)
{
    const uint32_t rep = fp * 0x00010001u, one = 0x00010001u, top = 0x80008000u;
    const uint32_t a = h.x ^ rep, b = h.y ^ rep, c = h.z ^ rep, d = (h.w ^ rep) | 0xFFFF0000u; // upper half of .w = count
    return (((a - one) & ~a) & top) | ((((b - one) & ~b) & top) >> 1) | ((((c - one) & ~c) & top) >> 2) |
           ((((d - one) & ~d) & top) >> 3);
}
// the same question as a yes/no, written as seven 16-bit compares: hipcc turns them into
// v_cmp_eq_u16 with sub-dword operand selects, whose lane masks combine on the scalar unit
__device__ __forceinline__ bool kid_hdr_any(const uint4 &h, uint32_t fp)
{
    const uint16_t f = (uint16_t)fp;
    return ((uint16_t)h.x == f) | ((uint16_t)(h.x >> 16) == f) | ((uint16_t)h.y == f) | ((uint16_t)(h.y >> 16) == f) |
           ((uint16_t)h.z == f) | ((uint16_t)(h.z >> 16) == f) | ((uint16_t)h.w == f);
}
__device__ __forceinline__ uint32_t kid_cand_entry(uint32_t bit) // bit index from kid_hdr_cand -> entry 0..6
{
    return 2u * (15u - (bit & 15u)) + (bit >> 4);
}

// Lookup in the minimizer-localised table: header, then the candidate entries, then the next line
// while the chain continues.  Returns the target (0 = absent); ncell = cells read.
__device__ __forceinline__ uint32_t kid_bucket_lookup(const KidDevDb &db, uint64_t key, uint32_t g, uint32_t &slot, uint32_t &ncell)
{
    uint32_t line = kid_minloc_line(g, db.line_shift);
    const uint32_t fp = kid_key_fp(key);
    ncell = 0;
    slot = 0;
    for (;;) {
        const uint32_t base = line * KID_LINE_CELLS;
        const uint4 h = db.table[base];
        ncell++;
        uint32_t m = kid_hdr_cand(h, fp);
        while (m) {
            const uint32_t j = kid_cand_entry((uint32_t)__builtin_ctz(m));
            m &= m - 1;
            const uint4 c = db.table[base + 1u + j];
            ncell++;
            if (c.z != 0 && c.x == (uint32_t)key && c.y == (uint32_t)(key >> 32)) { slot = c.w - 1u; return c.z; }
        }
        if (!kid_hdr_continues(h.w, fp)) return 0;
        line = (line + 1u) & db.line_mask;
    }
}

__device__ __forceinline__ uint32_t kid_dev_lookup(const KidDevDb &db, uint64_t key, uint32_t &slot, uint32_t &nprobe)
{
    uint32_t i = 0, res = 0;
    slot = 0;
    if (db.minloc) return kid_bucket_lookup(db, key, kid_minimizer_of_key(key, db.k), slot, nprobe);
    const uint64_t hash = kid_fmix64(key);
    uint64_t reprobe = 0;
    do {
        const uint32_t idx = ((uint32_t)hash + (uint32_t)reprobe) & db.slot_mask;
        reprobe += ++i;
        const uint4 c = db.table[idx];
        if (c.z == 0) break;
        if (c.x == (uint32_t)key && c.y == (uint32_t)(key >> 32)) { res = c.z; slot = c.w - 1u; break; }
    } while (reprobe < db.nslots && (db.max_probes == 0 || i < db.max_probes));
    nprobe = i;
    return res;
}

// ------------------------------------------------------------------ sliding-window minimum over a wavefront
// 16-lane DPP rows double as the blocks of the van Herk / Gil-Werman scheme: with P = prefix
// minimum and S = suffix minimum inside each row, min(a[p..p+15]) = min(S[p], P[p+15]).
// (a lane whose DPP source lies outside its row is disabled for that instruction: it keeps its own value.  A VGPR written
//  by a VALU instruction may be read through DPP two wait states later: s_nop 1 between dependent steps of ONE scan;
//  kid_row_scans interleaves four scans instead)
__device__ __forceinline__ uint32_t kid_row_prefix_min(uint32_t x)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xF, 0xF, false); x = x < t ? x : t; // row_shr:1
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xF, 0xF, false); x = x < t ? x : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xF, 0xF, false); x = x < t ? x : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xF, 0xF, false); x = x < t ? x : t;
    return x;
}
__device__ __forceinline__ uint32_t kid_row_suffix_min(uint32_t x)
{
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x101, 0xF, 0xF, false); x = x < t ? x : t; // row_shl:1
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x102, 0xF, 0xF, false); x = x < t ? x : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x104, 0xF, 0xF, false); x = x < t ? x : t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, 0x108, 0xF, 0xF, false); x = x < t ? x : t;
    return x;
}
// prefix and suffix minima of two tiles at once: p0 = s0 = hashes of the first tile, p1 = s1 = those of the second
__device__ __forceinline__ void kid_row_scans(uint32_t &p0, uint32_t &s0, uint32_t &p1, uint32_t &s1)
{
    p0 = kid_row_prefix_min(p0); s0 = kid_row_suffix_min(s0);
    p1 = kid_row_prefix_min(p1); s1 = kid_row_suffix_min(s1);
}

// ------------------------------------------------------------------ taxonomy
// Tree1::msca, newkmer_10nx.cpp:118-144, on the ancestor-row encoding.
// Let L = deepest common node of the two root paths.  The reference returns x
// when y lies on x's root path (L == y), y when x lies on y's (L == x), else L.
__device__ __forceinline__ uint32_t kid_row_entry(const uint4 &r, uint32_t d)
{
    const uint32_t w = d < 2 ? r.x : d < 4 ? r.y : d < 6 ? r.z : r.w;
    return (d & 1) ? (w >> 16) : (w & 0xFFFFu);
}

__device__ __forceinline__ uint32_t kid_msca_rows(uint32_t x, const uint4 &rx, uint32_t y, const uint4 &ry, uint4 &rout)
{
    const uint32_t dx = rx.x & 0xFFFFu, dy = ry.x & 0xFFFFu;
    const uint64_t lo = ((uint64_t)(rx.y ^ ry.y) << 32) | ((rx.x ^ ry.x) & 0xFFFF0000u);
    const uint64_t hi = ((uint64_t)(rx.w ^ ry.w) << 32) | (rx.z ^ ry.z);
    uint32_t f = lo ? (uint32_t)(__builtin_ctzll(lo) >> 4) : hi ? 4u + (uint32_t)(__builtin_ctzll(hi) >> 4) : 8u;
    uint32_t c = f - 1; // entries 1..f-1 agree
    c = c < dx ? c : dx;
    c = c < dy ? c : dy;
    if (c == 7 && dx == 8 && dy == 8 && x == y) c = 8; // depth-8 nodes are not stored in their own row
    if (c == dy) { rout = rx; return x; }
    if (c == dx) { rout = ry; return y; }
    rout = rx;
    rout.x = (rx.x & 0xFFFF0000u) | c;
    return c == 0 ? 1u : kid_row_entry(rx, c);
}

// same function by climbing parent[]/depth[] (any tree shape)
__device__ __forceinline__ uint32_t kid_msca_climb(const KidDevDb &db, uint32_t x, uint32_t y)
{
    uint32_t a = x, b = y;
    int32_t da = db.depth[a], dd = db.depth[b];
    while (da > dd) { a = (uint32_t)db.parent[a]; da--; }
    while (dd > da) { b = (uint32_t)db.parent[b]; dd--; }
    while (a != b) { a = (uint32_t)db.parent[a]; b = (uint32_t)db.parent[b]; }
    if (a == y) return x;
    if (a == x) return y;
    return a;
}

// ------------------------------------------------------------------ base packing
// ASCII bases -> 2-bit codes plus a mask of the bytes that are not ACGTacgt (those reset the reference's
// rolling window, newkmer_10nx.cpp:520-524).  4 bytes at a time, no per-byte branches.
// kid_pack4: the 4 bases of one dword (byte 0 = first base) -> codes in bits 7:0 (first base in bits 7:6),
// invalid flags in bits 3:0 of inv (bit j = byte j).
__device__ __forceinline__ void kid_pack4(const uint32_t x, const uint32_t u_is_t, uint32_t &codes, uint32_t &inv)
{
    const uint32_t c = ((x >> 1) ^ (x >> 2)) & 0x03030303u; // A,C,G,T -> 0,1,2,3 in every byte
    codes = (c * 0x40100401u) >> 24;                        // byte0 -> bits 7:6 ... byte3 -> bits 1:0
    // rebuild the upper-case letter each code stands for and compare
    const uint32_t c0 = c & 0x01010101u, c1 = (c >> 1) & 0x01010101u, t = c0 & c1;
    const uint32_t expect = 0x40404040u | (0x01010101u ^ t) | ((c0 ^ c1) << 1) | (c1 << 2) | (t << 4);
    uint32_t diff = (x & 0xDFDFDFDFu) ^ expect;
    if (u_is_t) diff &= ~t; // 'U' = 'T' ^ 1 and decodes to code 3
    const uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
    inv = (((nz >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// The same split in two for the pair loops, where almost no read has a base that is not ACGT: the codes and a word
// `diff` that is non-zero in every byte that is not ACGTacgt(Uu) -- seven instructions (v_perm_b32 looks up the
// letter a code stands for, v_dot4_u32_u8 gathers the four codes) --, and the 4-bit mask from `diff` when somebody
// needs it.
__device__ __forceinline__ uint32_t kid_codes4(const uint32_t x, const uint32_t u_is_t, uint32_t &diff)
{
    const uint32_t c = ((x >> 1) ^ (x >> 2)) & 0x03030303u;            // A,C,G,T -> 0,1,2,3 in every byte
    const uint32_t expect = __builtin_amdgcn_perm(0u, 0x54474341u, c); // byte j = "ACGT"[code of byte j]
    diff = (x & 0xDFDFDFDFu) ^ expect;
    if (u_is_t) diff &= ~(c & (c >> 1) & 0x01010101u);                 // 'U' = 'T' ^ 1 and decodes to code 3
    return __builtin_amdgcn_udot4(c, 0x01041040u, 0u, false);          // byte0 -> bits 7:6 ... byte3 -> bits 1:0
}
__device__ __forceinline__ uint32_t kid_inv4(const uint32_t diff)
{
    const uint32_t nz = (((diff & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | diff) & 0x80808080u;
    return (((nz >> 7) * 0x00204081u) >> 21) & 0xFu;
}
// 16 bases -> one packed word (first base in the top bits) + 16-bit mask
__device__ __forceinline__ void kid_pack16(const uint4 v, const uint32_t u_is_t, uint32_t &codes, uint32_t &inv)
{
    const uint32_t in[4] = {v.x, v.y, v.z, v.w};
    codes = 0;
    inv = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        uint32_t c8, i4;
        kid_pack4(in[q], u_is_t, c8, i4);
        codes = (codes << 8) | c8;
        inv |= i4 << (4 * q);
    }
}

// ------------------------------------------------------------------ batch preparation
// What a launch of the classify kernels finds in its device argument block: its share of the descriptors (or the fixed
// layout) and of the result array.  Set in classify-stream order.
__device__ __forceinline__ void kid_rebase(KidRareArgs *rare, const KidReadDesc *desc, uint32_t *out_final, unsigned long long read0,
                                           uint32_t fixed_len, int32_t fixed_nk)
{
    if (threadIdx.x == 0) {
        rare->desc = desc;
        rare->out_final = out_final;
        rare->read0 = read0;
        rare->fixed_len = fixed_len;
        rare->fixed_nk = fixed_nk;
    }
}

// [start, stop] -> descriptor; the range checks the reference leaves to string::at() happen here
__global__ void kid_prepare_kernel(const KidBatch b, int k, KidReadDesc *desc, unsigned long long *stats, KidRareArgs *rare, uint32_t seq,
                                   uint32_t long_cut, KidLongList *long_list, int rebase)
{
    // (the launch's descriptor / result pointers are set in classify-stream order: by kid_rebase_kernel when this kernel
    //  may run while the batch before is being classified, else -- `rebase`, same stream -- right here for the batch's
    //  first launch: one launch and its gap less per batch)
    if (rebase && blockIdx.x == 0) kid_rebase(rare, desc, b.out_final, 0ull, 0u, 0);
    uint32_t bad = 0, mx = 0;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < b.n; r += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t off;
        int64_t rl;
        if (b.offsets) { off = b.offsets[r]; rl = (int64_t)(b.offsets[r + 1] - off); }
        else { off = r * (uint64_t)b.fixed_len; rl = b.fixed_len; }
        int64_t s0 = b.start ? (int64_t)b.start[r] : 0;
        int64_t e0 = b.stop ? (int64_t)b.stop[r] : rl - 1;
        if (s0 <= e0 && (s0 < 0 || e0 >= rl)) { // never read outside the read; reported as KID_ERR_ARG
            bad++;
            if (s0 < 0) s0 = 0;
            if (e0 >= rl) e0 = rl - 1;
        }
        KidReadDesc d;
        int64_t nk = e0 - s0 + 1 - (k - 1);
        if (s0 > e0) nk = 0;
        d.first_base = off + (uint64_t)(s0 > 0 ? s0 : 0);
        d.n_kmers = (int32_t)(nk > 0x7FFFFFFF ? 0x7FFFFFFF : nk);
        d.pad = 0;
        // a record that goes to the long-record kernels (kid_long_hits_kernel / kid_long_fold_kernel): the classify
        // kernels see a read without k-mers (counted under target 0 until the fold corrects that)
        if (long_cut && nk > (int64_t)long_cut) {
            const uint32_t at = atomicAdd(&long_list->n, 1u);
            if (at < KID_LONG_MAX) {
                KidLongList::Item it;
                it.first_base = d.first_base; it.n_kmers = (uint32_t)d.n_kmers; it.read = (uint32_t)r;
                long_list->e[at] = it;
                d.n_kmers = 0;
            }
        }
        desc[r] = d;
        if (d.n_kmers > 0 && (uint32_t)d.n_kmers > mx) mx = (uint32_t)d.n_kmers;
    }
    // largest read of the batch, tagged with the batch number so that the word never needs a reset: one
    // atomic per workgroup at most (none once a workgroup sees a value as large as its own; a batch of
    // equal reads is announced by workgroup 0 alone)
    __shared__ uint32_t s_mx;
    if (threadIdx.x == 0) s_mx = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t y = (uint32_t)__shfl_xor((int)mx, o); mx = y > mx ? y : mx; }
    if ((threadIdx.x & 63u) == 0 && mx) atomicMax(&s_mx, mx);
    __syncthreads();
    const bool uniform_batch = !b.offsets && !b.start && !b.stop;
    if (threadIdx.x == 0 && (!uniform_batch || blockIdx.x == 0)) {
        const unsigned long long v = ((unsigned long long)seq << 32) | s_mx;
        if (v > *reinterpret_cast<volatile unsigned long long *>(&rare->batch_max)) atomicMax(&rare->batch_max, v);
    }
    if (bad) atomicAdd(&stats[4], (unsigned long long)bad);
}

// Sets a launch's argument block without kid_prepare_kernel: the later launches of a batch classified in several, and
// every launch of a fixed-layout batch (no descriptors: the host knows the one read length; batch_max = (seq << 32) | k-mers)
__global__ void kid_rebase_kernel(KidRareArgs *rare, const KidReadDesc *desc, uint32_t *out_final, unsigned long long read0,
                                  uint32_t fixed_len, int32_t fixed_nk, unsigned long long batch_max)
{
    kid_rebase(rare, desc, out_final, read0, fixed_len, fixed_nk);
    if (threadIdx.x == 0 && batch_max) rare->batch_max = batch_max;
}

// ------------------------------------------------------------------ classify
// One wavefront per read (DESIGN.md section 4 has the long version).  What a read needs:
//   1. lane i extracts the k-mer window starting at base i from the read's packed words (16 bases
//      each; general loops: staged in an LDS strip, pair kernels: fetched with ds_bpermute) with
//      two shifts -- no serial rolling --, derives the reverse complement with a bit reversal and
//      takes min(); with the minimizer-localised table the wave also computes the sliding-window
//      minimum of the hashed m-mers (DPP row scans + one cross-lane fetch), which selects the line;
//   2. the table is probed in HBM -- U windows per lane in flight at once.  Minimizer-localised
//      table: one header per lookup, which settles ~99 % of them; the others are queued in LDS and
//      resolved 64 at a time (candidate cell, ancestor row);
//   3. hits are folded with msca in read-position order (the fold is not associative:
//      newkmer_10nx.cpp:588-595);
//   4. hit cells are marked in the sample's seen-bitmap (ucount, :596-603).
// gcount is accumulated in an LDS histogram per workgroup and flushed once.
template <int U>
struct KidGroup {      // a group of U*64 windows between its two halves
    uint64_t key[U];
    uint32_t hlo[U];   // reference geometry: first cell of the probe sequence; minloc: the table line
    uint4 hd[U];       // minloc: header of that line (in flight)
    uint32_t fpp;      // minloc: the 16-bit key fingerprints of the U lookups, packed
    bool act[U];
};

// Three instantiations per configuration share this body, picked by the longest read of the batch
// (kid_prepare_kernel leaves it in rare->batch_max): PAIRK = 1 holds the hand-pipelined pair loop for
// batches of single-group reads (<= U*64 k-mers), PAIRK = 2 the same loop with the two groups of one
// read as the pair (<= 2*U*64 k-mers), PAIRK = 0 the general loops.  The host launches the one the
// batch is for when it knows the longest read, else all three: the others return at once.  Separate
// kernels, because each loop wants all 64 vector registers of an 8-waves-per-SIMD kernel for itself.
template <int U, bool ROWS, bool HIST, bool MINLOC, int KFIX, int PAIRK>
__global__ __launch_bounds__(512, KID_CLASSIFY_OCC) void kid_classify_kernel(const KidDevDb db, const KidInput b, const KidSampleDev s,
                                                            const uint32_t hist_words,
                                                            const KidReadDesc *__restrict__ const descs,
                                                            const KidRareArgs *__restrict__ const rare)
{
    static_assert(!PAIRK || (MINLOC && U == 2), "the pair loop exists for the minimizer-localised table, two windows per lane");
    if (MINLOC) { // wave-uniform, before anything else
        const uint32_t longest = (uint32_t)rare->batch_max;
        const int mode = longest <= (uint32_t)(U * 64) ? 1 : longest <= (uint32_t)(2 * U * 64) ? 2 : 0;
        if (mode != PAIRK) return;
    }
    // the launch on the device's own clock (kid_sample_kernel_time_device): first workgroup to start ... last one to end
    if (threadIdx.x == 0) atomicMin(&s.stats[30], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    // (descs == b.desc, passed once more as a restrict-qualified argument: the wave-uniform
    //  descriptor loads then become scalar loads, which stay in flight until first use)
    extern __shared__ uint32_t kid_smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform
    const uint32_t wpb = blockDim.x >> 6;
    uint32_t *hist = kid_smem;
    uint32_t *WA = kid_smem + hist_words + wib * (PAIRK ? KID_PAIR_LDS_WORDS : MINLOC ? KID_GEN_ML_LDS_WORDS : KID_WAVE_LDS_WORDS); // strip: 66 packed words + 34 mask words

    // the workgroup's totals for kid_sample_stats (behind the waves' areas): one set of global atomics per workgroup --
    // four per WAVE, all on one line, were 32 768 atomics per launch queueing up on one L2 atomic unit (~0.1 ms)
    unsigned long long *const WGS = reinterpret_cast<unsigned long long *>(
        kid_smem + hist_words + wpb * (PAIRK ? KID_PAIR_LDS_WORDS : MINLOC ? KID_GEN_ML_LDS_WORDS : KID_WAVE_LDS_WORDS));
    if (threadIdx.x < 4) WGS[threadIdx.x] = 0;
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < hist_words; i += blockDim.x) hist[i] = 0;
    }
    __syncthreads();

    const int k = KFIX ? KFIX : db.k; // KFIX = 30: the reference's KSIZE folded into the shifts and masks
    const uint32_t win = (uint32_t)kid_min_window(k); // m-mers per k-mer window: 15, 16 or 17
    const int mlen = kid_min_mlen(k);
    // lane classes of the sliding minimum.  With q = lane mod 16 the window of a lane leaves its 16-lane
    // row iff q + win > 16: then it is min(S[p], P[p+win-1]); inside one row it is the prefix P[p+win-1]
    // alone (q = 0) or the suffix alone (q = 1, win = 15)
    const uint32_t bp_src = ((lane + win - 1u) & 63u) << 2; // ds_bpermute address of lane + win - 1
    const bool p_same = lane + win - 1u < 64u;              // that lane is in the same tile
    const bool p_cross = (lane & 15u) + win > 16u, p_q0 = (lane & 15u) == 0u;
    const uint64_t gw = (uint64_t)blockIdx.x * wpb + wib;
    const uint64_t nw = (uint64_t)gridDim.x * wpb;
    // Counters for kid_sample_stats.  Wave-uniform state that only rare paths touch is kept out of
    // scalar registers (80 per wave at this occupancy, and the hot loop wants them all): lookups and
    // hits accumulate in the wave's LDS words, the msca row of the running result in lanes 0-3 of a
    // vector register, the probes beyond the first in a per-lane counter.
    uint32_t *const WC = WA + (PAIRK ? 0 : 100); // [0..1] lookups (64 bits), [2] hits, [3] probes beyond the first of a lookup
    unsigned long long *const WL = reinterpret_cast<unsigned long long *>(WC);
    if (lane < 4) WC[lane] = 0;
    uint32_t pend_t = 0, pend_n = 0;       // !HIST: run-length buffer in front of the global gcount atomics

    // ---- 1. stage one packed segment in a strip; returns "no base of it resets a window" (wave-uniform)
    auto stage = [&](uint32_t *W, const uint32_t codes, const uint32_t inv) -> bool {
        uint32_t *IM = W + 66;
        W[lane] = codes;
        reinterpret_cast<uint16_t *>(IM)[lane] = (uint16_t)inv;
        if (lane < 2) { W[64 + lane] = 0; IM[32 + lane] = 0; }
        const bool clean = (__ballot(inv != 0) == 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return clean;
    };

    // ---- 2, 3a. windows [t0, t0 + U*64) of the staged segment: keys, table line, header loads issued
    auto group_front = [&](const uint32_t *W, const uint32_t sh, const uint32_t nb, const uint32_t segk, const uint32_t t0,
                           const bool seg_clean, KidGroup<U> &g, uint32_t &n_bad, const bool issue_loads = true,
                           const uint32_t wcodes = 0, const uint32_t winv = 0) {
        // W == nullptr (pair kernels): no LDS strip; the lanes hold the read's packed words and masks in (wcodes, winv)
        // -- word c of the read in the LW = 4 (pair kernel) or 2 (duo kernel) lanes that loaded its 16 bases, see
        // pack_lanes -- and a window fetches its three words with ds_bpermute: one LDS round trip instead of
        // write + barrier + read.  Needs the whole read inside 64 / LW words, which one (two) group(s) of a read are.
        const bool direct = (W == nullptr);
        constexpr uint32_t LWS = PAIRK == 1 ? 2u : 1u; // log2 of the lanes per word
        auto word = [&](const uint32_t idx) -> uint32_t {
            return direct ? (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << (2u + LWS)), (int)wcodes) : W[idx];
        };
        auto mask32 = [&](const uint32_t idx) -> uint32_t { // invalid-mask bits of bases 32 idx .. 32 idx + 31
            if (!direct) return (W + 66)[idx];
            const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idx << (3u + LWS)), (int)winv);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx << (3u + LWS)) + (4u << LWS)), (int)winv);
            return (lo & 0xFFFFu) | (hi << 16);
        };
        uint32_t P[U + 1], S[U]; // minloc: row prefix / suffix minima of the hashed m-mers
        const uint32_t pmax = sh + nb - (uint32_t)mlen; // last m-mer start inside the segment
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint32_t i = t0 + (uint32_t)u * 64u + lane;
            // one window extraction serves the k-mer AND the m-mer that starts at the same base;
            // lanes past the last k-mer still hash their m-mer (the windows of earlier lanes reach
            // 14 positions ahead), clamped to the last one that lies inside the segment
            uint32_t p = sh + i;
            p = MINLOC ? (p < pmax ? p : pmax) : sh + (i < segk ? i : 0u);
            const uint32_t w0 = p >> 4, o2 = (p & 15u) * 2u;
            const uint64_t A = ((uint64_t)word(w0) << 32) | word(w0 + 1);
            const uint64_t B = word(w0 + 2);
            const uint64_t x = (A << o2) | ((B << o2) >> 32);
            const uint64_t keyF = x >> (64 - 2 * k);
            // one reversal of the 32-base window serves both reverse complements: base j of the window
            // lands in bits [2j+1, 2j], so the low 2k bits are the k-mer's and the low 2m bits the m-mer's
            const uint64_t nrv = ~kid_rev2(x);
            const uint64_t keyR = nrv & (~0ull >> (64 - 2 * k));
            bool valid = (i < segk);
            if (!seg_clean) { // rare: some base of the segment is not ACGTacgt
                const uint64_t im = (((uint64_t)mask32((p >> 5) + 1) << 32) | mask32(p >> 5)) >> (p & 31u);
                const bool ok = ((im & ((1ull << k) - 1ull)) == 0);
                n_bad += (uint32_t)__popcll(__ballot(valid && !ok));
                valid = valid && ok;
            }
            g.key[u] = keyF < keyR ? keyF : keyR; // newkmer_10nx.cpp:528
            if (!MINLOC) g.hlo[u] = (uint32_t)kid_fmix64(g.key[u]) & db.slot_mask;
            g.act[u] = valid;
            if (MINLOC) {
                const uint32_t h = kid_mmer_hash2((uint32_t)(x >> (64 - 2 * mlen)), (uint32_t)nrv & (0xFFFFFFFFu >> (32 - 2 * mlen)));
                P[u] = h;
                S[u] = h;
            }
        }
        if (MINLOC) {
            if constexpr (U == 2) kid_row_scans(P[0], S[0], P[1], S[1]);
            else {
#pragma unroll
                for (int u = 0; u < U; u++) { P[u] = kid_row_prefix_min(P[u]); S[u] = kid_row_suffix_min(S[u]); }
            }
        }
        if (MINLOC) {
            g.fpp = 0;
#pragma unroll
            for (int u = 0; u < U; u++) g.fpp |= kid_key_fp(g.key[u]) << (16 * u);
            // the win-1 m-mers behind the last k-mer of the group -- if any window of the group reaches that
            // far (wave-uniform: a 100-bp read ends inside the group, m-mers and all)
            P[U] = 0xFFFFFFFFu;
            if (segk + win - 1u > t0 + (uint32_t)U * 64u) {
                uint32_t p = sh + t0 + (uint32_t)U * 64u + lane;
                p = p < pmax ? p : pmax;
                const uint32_t w0 = p >> 4, o2 = (p & 15u) * 2u;
                const uint64_t A = ((uint64_t)word(w0) << 32) | word(w0 + 1);
                P[U] = kid_row_prefix_min(kid_mmer_hash((uint32_t)((A << o2) >> (64 - 2 * mlen)), mlen));
            }
            // ... and their minimum over every window a[p..p+win-1].  With q = p mod 16: the window
            // leaves its 16-lane row iff q + win > 16, then it is min(S[p], P[p+win-1]); inside
            // one row it is exactly the prefix P[p+win-1] (q = 0) or the suffix S[p] (win = 15, q = 1)
            uint32_t nxt = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bp_src, (int)P[0]);
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t same = nxt;
                nxt = (uint32_t)__builtin_amdgcn_ds_bpermute((int)bp_src, (int)P[u + 1]);
                const uint32_t pn = p_same ? same : nxt;
                const uint32_t both = S[u] < pn ? S[u] : pn;
                const uint32_t mz = p_cross ? both : (p_q0 ? pn : S[u]);
                // one 16-byte header per lookup settles every absent key; lanes that share a
                // minimizer read the same header (one sector for all of them)
                g.hlo[u] = kid_minloc_line(mz, db.line_shift);
                g.hd[u] = make_uint4(0, 0, 0, 0);
                if (issue_loads && g.act[u]) g.hd[u] = kid_load_cell(db.table, g.hlo[u] * KID_LINE_CELLS);
            }
        }
    };

    // ---- 3b, 4, 5. the rest of the probe sequences, seen-bitmap, ordered fold into final_t (vfrow:
    // the ancestor row of final_t, in lanes 0-3)
    auto group_back = [&](KidGroup<U> &g, uint32_t &final_t, uint32_t &vfrow) {
        uint32_t tgt[U], slot[U];
#pragma unroll
        for (int u = 0; u < U; u++) { tgt[u] = 0; slot[u] = 0; }
        if (MINLOC) {
            uint32_t mm[U], fp[U];
            bool more = false;
#pragma unroll
            for (int u = 0; u < U; u++) {
                fp[u] = (g.fpp >> (16 * u)) & 0xFFFFu;
                mm[u] = (g.act[u] && (kid_hdr_any(g.hd[u], fp[u]) || kid_hdr_continues(g.hd[u].w, fp[u]))) ? 1u : 0u;
                more |= mm[u] != 0;
            }
            if (__ballot(more) == 0) return; // (wave-uniform) ~99 % of the lanes are settled by their header
            // fingerprint matches (almost always the key itself): the first candidate cell of every
            // pending lookup is requested at once -- one round trip for the whole group.  From here on a
            // header is only its candidate set and its "line continues" flag.
            uint32_t m[U], idx[U];
            bool full[U];
            uint4 c[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                m[u] = mm[u] ? kid_hdr_cand(g.hd[u], fp[u]) : 0u;
                full[u] = mm[u] && kid_hdr_continues(g.hd[u].w, fp[u]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                idx[u] = 0;
                c[u] = make_uint4(0, 0, 0, 0);
                if (m[u]) {
                    idx[u] = g.hlo[u] * KID_LINE_CELLS + 1u + kid_cand_entry((uint32_t)__builtin_ctz(m[u]));
                    m[u] &= m[u] - 1;
                    c[u] = kid_load_cell(db.table, idx[u]);
                    atomicAdd(&WC[3], 1u);
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                bool go = mm[u] != 0;
                if (idx[u] && c[u].z != 0 && c[u].x == (uint32_t)g.key[u] && c[u].y == (uint32_t)(g.key[u] >> 32)) {
                    tgt[u] = c[u].z; slot[u] = c[u].w - 1u; go = false;
                }
                // leftovers (a second candidate: 1e-4 of the lookups; a chained line: 1e-5)
                uint32_t ln = g.hlo[u], mu = m[u];
                bool fu = full[u];
                while (go) {
                    if (mu) {
                        const uint32_t j = kid_cand_entry((uint32_t)__builtin_ctz(mu));
                        mu &= mu - 1;
                        const uint32_t ix = ln * KID_LINE_CELLS + 1u + j;
                        const uint4 cc = kid_load_cell(db.table, ix);
                        atomicAdd(&WC[3], 1u);
                        if (cc.z != 0 && cc.x == (uint32_t)g.key[u] && cc.y == (uint32_t)(g.key[u] >> 32)) { tgt[u] = cc.z; slot[u] = cc.w - 1u; go = false; }
                    } else if (fu) {
                        ln = (ln + 1u) & rare->line_mask;
                        const uint4 h = kid_load_cell(db.table, ln * KID_LINE_CELLS);
                        atomicAdd(&WC[3], 1u);
                        mu = kid_hdr_cand(h, fp[u]);
                        fu = kid_hdr_continues(h.w, fp[u]);
                    } else go = false;
                }
            }
        } else {
            uint64_t rp[U];
            uint32_t step[U];
#pragma unroll
            for (int u = 0; u < U; u++) { rp[u] = 0; step[u] = 0; }
            bool any = false;
#pragma unroll
            for (int u = 0; u < U; u++) any |= g.act[u];
            while (any) {
                uint4 c[U];
                uint32_t idx[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    idx[u] = (g.hlo[u] + (uint32_t)rp[u]) & db.slot_mask;
                    c[u] = make_uint4(0, 0, 0, 0);
                    if (g.act[u]) c[u] = kid_load_cell_nt(db.table, idx[u]);
                }
                any = false;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (g.act[u]) {
                        step[u]++;
                        rp[u] += step[u];
                        if (step[u] > 1) atomicAdd(&WC[3], 1u);
                        if (c[u].z == 0) g.act[u] = false;
                        else if (c[u].x == (uint32_t)g.key[u] && c[u].y == (uint32_t)(g.key[u] >> 32)) {
                            tgt[u] = c[u].z; slot[u] = c[u].w - 1u; g.act[u] = false;
                        } else if (!(rp[u] < db.nslots) || (db.max_probes != 0 && step[u] >= db.max_probes)) g.act[u] = false;
                    }
                    any |= g.act[u];
                }
            }
        }
        uint4 row[U];
        uint64_t hitm[U];
        uint32_t nh = 0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            row[u] = make_uint4(0, 0, 0, 0);
            if (tgt[u] > 0) {
                if (ROWS) row[u] = db.rows[tgt[u]];
                if (tgt[u] > 1) kid_atomic_or_nowait(&s.seen[slot[u] >> 5], 1u << (slot[u] & 31u));
            }
            hitm[u] = __ballot(tgt[u] > 0);
            nh += (uint32_t)__popcll(hitm[u]);
        }
        if (nh == 0) return;
        if (lane == 0) atomicAdd(&WC[2], nh);
#pragma unroll
        for (int u = 0; u < U; u++) {
            uint64_t m = hitm[u];
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t x = (uint32_t)__builtin_amdgcn_readlane((int)tgt[u], j);
                if (x == final_t) continue; // msca(x,x) = x
                uint4 rx = make_uint4(0, 0, 0, 0);
                if (ROWS) {
                    rx.x = (uint32_t)__builtin_amdgcn_readlane((int)row[u].x, j);
                    rx.y = (uint32_t)__builtin_amdgcn_readlane((int)row[u].y, j);
                    rx.z = (uint32_t)__builtin_amdgcn_readlane((int)row[u].z, j);
                    rx.w = (uint32_t)__builtin_amdgcn_readlane((int)row[u].w, j);
                }
                if (final_t != 0) { // :588-591
                    if (ROWS) {
                        uint4 fr, ro;
                        fr.x = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 0);
                        fr.y = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 1);
                        fr.z = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 2);
                        fr.w = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 3);
                        final_t = kid_msca_rows(x, rx, final_t, fr, ro);
                        rx = ro;
                    } else {
                        final_t = kid_msca_climb(db, x, final_t);
                    }
                } else {
                    final_t = x; // :592-595
                }
                if (ROWS)
                    vfrow = lane == 0 ? rx.x : lane == 1 ? rx.y : lane == 2 ? rx.z : lane == 3 ? rx.w : vfrow;
            }
        }
    };

    // ---- gcount[final]++ (:605) and the per-read output
    auto finish_read = [&](const uint64_t r, const uint32_t final_t, const uint32_t n_valid) {
        if (HIST) {
            if (lane == 0) { atomicAdd(&hist[final_t], 1u); atomicAdd(WL, (unsigned long long)n_valid); }
        } else {
            if (lane == 0) atomicAdd(WL, (unsigned long long)n_valid);
            if (final_t == pend_t) {
                pend_n++;
            } else {
                if (pend_n && lane == 0) atomicAdd(&rare->gcount[pend_t], (unsigned long long)pend_n);
                pend_t = final_t;
                pend_n = 1;
            }
        }
        if (lane == 0 && b.out_final) kid_store_u32_nowait(&b.out_final[r], final_t);
    };

    // ==== minimizer-localised table: the back half is deferred (both kernels) =======================
    // ---- back half, deferred.  A lookup whose header shows a fingerprint match (or a chained line)
    // is not followed up on the spot -- that would put two more dependent round trips (hit cell,
    // ancestor row) on every second read -- but queued in LDS; queued lookups are resolved 64 at a
    // time, one per lane, and folded read by read (entries are in read order, then window order).
    uint32_t *const CQ_klo = WA + (PAIRK ? 4 : 104), *const CQ_khi = CQ_klo + KID_CQ_CAP, *const CQ_lw = CQ_khi + KID_CQ_CAP;
    uint8_t *const CQ_tag = reinterpret_cast<uint8_t *>(CQ_lw + KID_CQ_CAP); // read number mod 64
    // per-read results wait in LDS for one scattered store per 64 reads: a pending store shares vmcnt
    // with the loads, and the explicit counts of the loop would have to sit out its acknowledgement
    uint32_t *const RB = CQ_lw + KID_CQ_CAP + KID_CQ_CAP / 4;
    const uint32_t gw32 = (uint32_t)gw, nw32 = (uint32_t)nw; // 32-bit read indices (n < 2^31): read number i of this wave is gw + i nw
    uint32_t rd_first = gw32, rd_stride = nw32; // ... except where a wave takes a contiguous range (set below): rd_first + i
    uint32_t gen_cnt = 0;                         // general loops: reads of this wave
    uint64_t rb_skip = 0;   // result slots of the current block of 64 reads that are not this pass's to store
    bool rb_direct = false; // second pass of the general loops (reads of more than one segment): results stored at once
    uint32_t *const RG = RB + 64; // pair kernel: which read of the batch a result slot belongs to (128 slots: the next block's are known early)
    auto flush_results = [&](const uint32_t i0, const uint32_t n) { // this wave's reads i0 .. i0+n-1 (n <= 64)
        uint32_t *const outp = rare->out_final;
        if (outp && lane < n && !((rb_skip >> ((i0 + lane) & 63u)) & 1ull))
            kid_store_u32_nowait(&outp[PAIRK == 1 ? RG[(i0 + lane) & 127u] : rd_first + (i0 + lane) * rd_stride], RB[(i0 + lane) & 63u]);
        rb_skip = 0;
        RB[lane] = 0; // (a read without any hit does not write its slot: see commit_zero)
    };
    if (MINLOC) RB[lane] = 0;
    uint32_t n_zero = 0;    // wave-uniform: reads classified as 0 that have not been added to gcount[0] yet
    uint32_t qn = 0;        // wave-uniform fill of the queue
    uint32_t n_lookups = 0; // wave-uniform; per wave and launch: stays below 2^32 for batches of < 2^31 reads x 128 k-mers (longer reads: mod 2^32 is accepted for this counter)
    // the read the resolver left open (its run ended a chunk, or it is still being classified), and its fold so far
    uint32_t cur_tag = 0xFFFFFFFFu, final_c = 0, vfrow_c = 0;
    auto commit = [&](const uint32_t i, const uint32_t final_t) { // gcount[final]++ (:605), per-read output
        if (HIST) {
            if (lane == 0) atomicAdd(&hist[final_t >> 1], 1u << (16u * (final_t & 1u)));
        } else if (final_t == pend_t) {
            pend_n++;
        } else {
            if (pend_n && lane == 0) atomicAdd(&rare->gcount[pend_t], (unsigned long long)pend_n);
            pend_t = final_t;
            pend_n = 1;
        }
        if (rb_direct) { if (lane == 0 && b.out_final) kid_store_u32_nowait(&b.out_final[rd_first + i * rd_stride], final_t); }
        else if (lane == 0) RB[i & 63u] = final_t;
    };
    // the common case, a read without a single candidate: counted in a scalar register, its result slot is
    // zero already
    auto commit_zero = [&](const uint32_t i) {
        n_zero++;
        if (rb_direct && lane == 0 && b.out_final) kid_store_u32_nowait(&b.out_final[rd_first + i * rd_stride], 0u);
    };
    // i_now: number of the newest queued read (all are within 63 of it); open_tag: a read that may still get
    // entries (general loops, between the groups of a read) -- its run is folded but not committed
    auto resolve_all = [&](const uint32_t i_now, const uint32_t open_tag) {
        // (the pair kernel never leaves a read open between calls: its carry lives and dies in here)
        uint32_t ctag = PAIRK == 1 ? 0xFFFFFFFFu : cur_tag, final_t = PAIRK == 1 ? 0u : final_c, vfrow = PAIRK == 1 ? 0u : vfrow_c;
        auto commit_tag = [&](const uint32_t tag, const uint32_t f) {
            commit(i_now - ((i_now - tag) & 63u), f);
        };
        for (uint32_t base = 0; base < qn; base += 64u) {
            const uint32_t n = qn - base < 64u ? qn - base : 64u;
            const bool valid = lane < n;
            // places in the hit log for this chunk: asked for now, looked at when the targets are known
            uint32_t *const slog = rare->seen_log;
            uint32_t log_at = 0;
            if (slog && lane == 0) log_at = atomicAdd(rare->seen_log_tail + (blockIdx.x & (KID_LOG_SHARDS - 1u)) * 16u, n);
            uint32_t klo = 0, khi = 0, ln = 0, tag = 0;
            if (valid) { klo = CQ_klo[base + lane]; khi = CQ_khi[base + lane]; ln = CQ_lw[base + lane]; tag = CQ_tag[base + lane]; }
            const bool verified = (tag & 0x80u) != 0; // {target, entry ordinal} fetched when the header came in
            tag &= 63u;
            // the header once more (the queue keeps only the line: working out the candidates at
            // queueing time would cost the hot loop ~45 instructions per read with a match)
            uint32_t tgt = 0, slot = 0, mu = 0;
            const uint32_t fp = kid_key_fp(((uint64_t)khi << 32) | klo);
            bool fu = false;
            if (verified) { tgt = klo; slot = khi; }
            if (valid && !verified) {
                const uint4 h = kid_load_cell(db.table, ln * KID_LINE_CELLS);
                mu = kid_hdr_cand(h, fp);
                fu = kid_hdr_continues(h.w, fp);
            }
            { // (cells read: one add for the wave, not one per lane on the same LDS word)
                const uint64_t cm = __ballot(mu != 0);
                if (cm && lane == 0) atomicAdd(&WC[3], (uint32_t)__popcll(cm));
            }
            if (mu) { // the first candidate of every entry in one round trip: almost always the key itself
                const uint32_t idx = ln * KID_LINE_CELLS + 1u + kid_cand_entry((uint32_t)__builtin_ctz(mu));
                mu &= mu - 1;
                const uint4 c = kid_load_cell(db.table, idx);
                if (c.z != 0 && c.x == klo && c.y == khi) { tgt = c.z; slot = c.w - 1u; }
            }
            bool go = valid && !verified && tgt == 0 && (mu != 0 || fu);
            while (go) { // a second candidate (1e-4 of the lookups) or a chained line (1e-5)
                if (mu) {
                    const uint32_t j = kid_cand_entry((uint32_t)__builtin_ctz(mu));
                    mu &= mu - 1;
                    const uint32_t ix = ln * KID_LINE_CELLS + 1u + j;
                    const uint4 cc = kid_load_cell(db.table, ix);
                    atomicAdd(&WC[3], 1u);
                    if (cc.z != 0 && cc.x == klo && cc.y == khi) { tgt = cc.z; slot = cc.w - 1u; go = false; }
                } else if (fu) {
                    ln = (ln + 1u) & rare->line_mask;
                    const uint4 h = kid_load_cell(db.table, ln * KID_LINE_CELLS);
                    atomicAdd(&WC[3], 1u);
                    mu = kid_hdr_cand(h, fp);
                    fu = kid_hdr_continues(h.w, fp);
                } else go = false;
            }
            uint4 row = make_uint4(0, 0, 0, 0);
            if (tgt > 0) {
                if (ROWS) row = rare->rows[tgt];
            }
            {
                // kmer_seen (:596-600).  The entries of a pass are in read and window order, and consecutive k-mers of a
                // genome carry consecutive entry ordinals when the database lists them in genome order (the reference's
                // builder does): the hits of a read then fall into two or three words of the bitmap.  Lanes that name the
                // same word as a lane 1, 2, 4 or 8 places before them in their 16-lane row take that lane's bits along
                // (bits of the same word, set by hits of this pass: OR-ing them in once more is harmless), and only the
                // last lane of a run issues the atomic.  60 hits per read: 121 M atomics per 2 M reads became 1/16 of
                // that; each one occupies an L2 channel for ~16 cycles (profiles/r02/ab_hitlog.txt).
                const bool sb = tgt > 1;
                // The log: one coalesced store per chunk instead of one memory-side atomic per hit -- 2.5 M of those per
                // 1 M read pairs, each worth several line fetches of channel time (8-10 % of the launch,
                // profiles/r03/ab_noseen.txt).  kid_seenlog_* sets the bits later, a bitmap piece at a time in LDS.
                bool logged = false;
                if (slog) {
                    const uint32_t cap = rare->seen_log_cap;
                    const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int)log_at);
                    uint32_t *const at_p = slog + (size_t)(blockIdx.x & (KID_LOG_SHARDS - 1u)) * cap + at;
                    if (at <= cap - n) { // (cap >= 64)
                        if (valid) kid_store_u32_nowait(at_p + lane, sb ? slot : KID_LOG_NONE);
                        logged = true;
                    } else if (at < cap && lane < cap - at) {
                        // the region is full: back to atomics until the log has been applied -- but the places that were
                        // handed out up to its end must not be left as they are (the pass reads everything below the end)
                        kid_store_u32_nowait(at_p + lane, KID_LOG_NONE);
                    }
                }
                if (!logged) {
                const uint32_t word = sb ? slot >> 5 : 0xFFFFFFF0u + (lane & 15u); // (no two neighbours without a hit alike)
                uint32_t bits = sb ? 1u << (slot & 31u) : 0u;
#define KID_SEEN_STEP(CTRL)                                                                                                    \
                {                                                                                                               \
                    const uint32_t w_ = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)word, CTRL, 0xF, 0xF, false); \
                    const uint32_t b_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, CTRL, 0xF, 0xF, false);               \
                    if (w_ == word) bits |= b_;                                                                                 \
                }
                KID_SEEN_STEP(0x111) KID_SEEN_STEP(0x112) KID_SEEN_STEP(0x114) KID_SEEN_STEP(0x118) // row_shr:1, 2, 4, 8
#undef KID_SEEN_STEP
                const uint32_t w_next = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)word, 0x101, 0xF, 0xF, false); // row_shl:1
                if (sb && w_next != word) kid_atomic_or_nowait(&rare->seen[word], bits);
                }
            }
            const uint64_t hitm = __ballot(tgt > 0);
            if (hitm && lane == 0) atomicAdd(&WC[2], (uint32_t)__popcll(hitm));
            // Fold read by read, one lane per read: entries of a read are neighbours (a change of tag
            // starts the next read); the first lane of a run walks its run in window order, fetching
            // (target, ancestor row) of entry lane + t with ds_bpermute.  The left fold of msca over
            // the hits of a read, newkmer_10nx.cpp:588-595.
            const uint32_t tprev = (uint32_t)__shfl_up((int)tag, 1);
            const bool head = valid && (lane == 0 || tag != tprev);
            const uint64_t hm = __ballot(head);
            const uint64_t above = lane < 63u ? (hm >> (lane + 1u)) : 0ull;
            const uint32_t len = above ? (uint32_t)__builtin_ctzll(above) + 1u : n - lane; // entries of this run (head lanes)
            uint32_t f = 0;
            uint4 fr = make_uint4(0, 0, 0, 0);
            if (lane == 0 && ctag != 0xFFFFFFFFu) {
                if (valid && tag == ctag) { // the run continues the read left open by the chunk before
                    f = final_t;
                    fr.x = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 0);
                    fr.y = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 1);
                    fr.z = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 2);
                    fr.w = (uint32_t)__builtin_amdgcn_readlane((int)vfrow, 3);
                }
            }
            if (ctag != 0xFFFFFFFFu && (uint32_t)__builtin_amdgcn_readfirstlane((int)tag) != ctag) commit_tag(ctag, final_t);
            if (__popcll(hm) <= 6) {
                // Few, long runs (reads with many hits): the runs in turn, all entries of a run at once.
                // Every lane works out the step its own entry would make from the run's current result;
                // entries up to the first one that changes the result leave it as it is -- which is all
                // the sequential fold would have done with them -- so the run jumps there and repeats.
                // One round per change of the result, a handful per read, instead of one per hit.
                uint64_t runs = hm;
                while (runs) {
                    const int h = __builtin_ctzll(runs);
                    runs &= runs - 1;
                    const uint32_t e = runs ? (uint32_t)__builtin_ctzll(runs) : n; // the run is [h, e)
                    uint32_t uf = (uint32_t)__builtin_amdgcn_readlane((int)f, h);
                    uint4 ufr;
                    ufr.x = (uint32_t)__builtin_amdgcn_readlane((int)fr.x, h);
                    ufr.y = (uint32_t)__builtin_amdgcn_readlane((int)fr.y, h);
                    ufr.z = (uint32_t)__builtin_amdgcn_readlane((int)fr.z, h);
                    ufr.w = (uint32_t)__builtin_amdgcn_readlane((int)fr.w, h);
                    uint64_t rem = hitm & (e >= 64u ? ~0ull : ((1ull << e) - 1ull)) & ~((1ull << h) - 1ull);
                    while (rem) {
                        uint32_t rj = tgt;
                        uint4 roj = row;
                        if (uf != 0 && tgt != uf) { // (first hit: :592-595; msca(x,x) = x)
                            if (ROWS) rj = kid_msca_rows(tgt, row, uf, ufr, roj);
                            else rj = kid_msca_climb(db, tgt > 0 ? tgt : uf, uf);
                        }
                        const uint64_t ch = __ballot(rj != uf) & rem;
                        if (!ch) break;
                        const int j = __builtin_ctzll(ch);
                        uf = (uint32_t)__builtin_amdgcn_readlane((int)rj, j);
                        ufr.x = (uint32_t)__builtin_amdgcn_readlane((int)roj.x, j);
                        ufr.y = (uint32_t)__builtin_amdgcn_readlane((int)roj.y, j);
                        ufr.z = (uint32_t)__builtin_amdgcn_readlane((int)roj.z, j);
                        ufr.w = (uint32_t)__builtin_amdgcn_readlane((int)roj.w, j);
                        rem &= j >= 63 ? 0ull : ~((2ull << j) - 1ull);
                    }
                    if (lane == (uint32_t)h) { f = uf; fr = ufr; }
                }
            } else
            for (uint32_t t = 0; __ballot(head && t < len) != 0; t++) {
                const int src = (int)(((lane + t) & 63u) << 2);
                const uint32_t x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)tgt);
                uint4 rx;
                rx.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)row.x);
                rx.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)row.y);
                rx.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)row.z);
                rx.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)row.w);
                if (head && t < len && x != 0 && x != f) { // msca(x,x) = x
                    if (f != 0) {
                        if (ROWS) {
                            uint4 ro;
                            f = kid_msca_rows(x, rx, f, fr, ro);
                            fr = ro;
                        } else {
                            f = kid_msca_climb(db, x, f);
                        }
                    } else {
                        f = x; // :592-595
                        fr = rx;
                    }
                }
            }
            // every run but the last one is a finished read; the last one too unless a chunk follows
            const uint32_t last = 63u - (uint32_t)__builtin_clzll(hm); // hm != 0: n >= 1
            const bool more_chunks = base + 64u < qn || (uint32_t)__builtin_amdgcn_readlane((int)tag, (int)last) == open_tag; // "keep the last run open"
            if (head && (lane != last || !more_chunks)) {
                if (HIST) atomicAdd(&hist[f >> 1], 1u << (16u * (f & 1u)));
                else atomicAdd(&rare->gcount[f], 1ull);
                if (rb_direct) { if (b.out_final) kid_store_u32_nowait(&b.out_final[rd_first + (i_now - ((i_now - tag) & 63u)) * rd_stride], f); }
                else RB[tag] = f; // tag = read number mod 64
            }
            if (more_chunks) {
                ctag = (uint32_t)__builtin_amdgcn_readlane((int)tag, (int)last);
                final_t = (uint32_t)__builtin_amdgcn_readlane((int)f, (int)last);
                const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)fr.x, (int)last), a1 = (uint32_t)__builtin_amdgcn_readlane((int)fr.y, (int)last);
                const uint32_t a2 = (uint32_t)__builtin_amdgcn_readlane((int)fr.z, (int)last), a3 = (uint32_t)__builtin_amdgcn_readlane((int)fr.w, (int)last);
                vfrow = lane == 0 ? a0 : lane == 1 ? a1 : lane == 2 ? a2 : lane == 3 ? a3 : vfrow;
            } else {
                ctag = 0xFFFFFFFFu;
            }
        }
        if (ctag != 0xFFFFFFFFu && ctag != open_tag) { commit_tag(ctag, final_t); ctag = 0xFFFFFFFFu; }
        if (PAIRK != 1) { cur_tag = ctag; final_c = final_t; vfrow_c = vfrow; }
        qn = 0;
        // a real s_waitcnt (not inline assembly): hipcc's waitcnt pass then knows that none of ITS loads is
        // pending when this rare path rejoins the loop, and does not drain vmcnt at the top of every trip
        __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0), expcnt and lgkmcnt untouched
    };
    // header test of one group of a read; its unsettled lookups go to the queue.  false: there were none
    auto back_deferred = [&](const KidGroup<U> &g, const uint32_t i) -> bool {
        uint32_t fp[U];
        bool mm[U], more = false;
#pragma unroll
        for (int u = 0; u < U; u++) {
            fp[u] = (g.fpp >> (16 * u)) & 0xFFFFu;
            // (a full line: word 3 >= 8 << 16 -- filter bits are only ever set on full lines.  Whether THIS key was
            //  pushed past it is looked at by the resolver: testing the filter bit here costs the hot loop a register)
            mm[u] = g.act[u] && (kid_hdr_any(g.hd[u], fp[u]) || (g.hd[u].w >> 16) >= KID_HDR_FULL);
            more |= mm[u];
        }
        if (__ballot(more) == 0) return false; // (wave-uniform) ~99 % of the lookups are settled by their header
#pragma unroll
        for (int u = 0; u < U; u++) {
            const uint64_t qm = __ballot(mm[u]);
            if (qm == 0) continue;
            // (only while hits are sparse: with many flagged lookups per tile the resolver runs every few reads, finds the
            //  lines still in the L2, and the wait for the candidate here would cost more than its second fetch --
            //  builder-shaped database, 15.9 hits per read: 2.15 vs 1.92 ms per 1 M pairs, profiles/r02/clumped_ec.txt)
            const bool early = (uint32_t)__popcll(qm) <= KID_EARLY_MAX && qn < KID_EARLY_QN;
            const uint32_t cm = (early && mm[u]) ? kid_hdr_cand(g.hd[u], fp[u]) : 0u;
            { // (cells read: one add for the wave)
                const uint64_t cmb = __ballot(cm != 0);
                if (cmb && lane == 0) atomicAdd(&WC[3], (uint32_t)__popcll(cmb));
            }
            if (mm[u]) {
                const uint32_t pos = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(qm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)qm, 0u));
                uint32_t w0 = (uint32_t)g.key[u], w1 = (uint32_t)(g.key[u] >> 32), tagv = i & 63u;
                if (cm) { // the first candidate, while its line is in the cache
                    const uint32_t idx = g.hlo[u] * KID_LINE_CELLS + 1u + kid_cand_entry((uint32_t)__builtin_ctz(cm));
                    const uint4 c = kid_load_cell(db.table, idx);
                    if (c.z != 0 && c.x == w0 && c.y == w1) { w0 = c.z; w1 = c.w - 1u; tagv |= 0x80u; } // verified: {target, ordinal}
                }
                CQ_klo[pos] = w0;
                CQ_khi[pos] = w1;
                CQ_lw[pos] = g.hlo[u];
                CQ_tag[pos] = (uint8_t)tagv;
            }
            qn += (uint32_t)__popcll(qm);
        }
        return true;
    };

    // ---- a whole read of any length on its own, given its descriptor and its first packed segment.
    // `prefetch` issues the loads for the reads behind this one; it is called right after this read's
    // first header loads went out (not before the loops: the compiler drains vmcnt in front of a loop)
    auto process_read = [&](const uint64_t r, const uint32_t i, const uint64_t first, const int64_t nk, const kid_u4 st_raw,
                            auto &&prefetch) {
        bool prefetched = false, had = false;
        uint32_t final_t = 0;
        uint32_t vfrow = 0, n_bad = 0;
        for (int64_t seg = 0; seg < nk; seg += KID_SEG_KMERS) {
            const uint32_t segk = (uint32_t)((nk - seg) < KID_SEG_KMERS ? (nk - seg) : KID_SEG_KMERS);
            const uint64_t b0 = first + (uint64_t)seg; // first base of the segment
            const uint32_t nb = segk + (uint32_t)k - 1;
            const uint64_t c0 = b0 >> 4;
            const uint32_t sh = (uint32_t)(b0 & 15ull);
            const uint32_t nchunks = (sh + nb + 15u) >> 4; // <= 64
            kid_u4 raw = st_raw;
            if (seg != 0) // long reads: later segments are fetched on the spot (lanes past the segment: its first chunk again)
                raw = *reinterpret_cast<const kid_u4 *>(b.bases + 16ull * (c0 + (lane < nchunks ? lane : 0u)));
            uint32_t codes, inv;
            kid_pack16(make_uint4(raw.x, raw.y, raw.z, raw.w), db.u_is_t, codes, inv);
            const bool seg_clean = stage(WA, codes, inv);
            for (uint32_t t0 = 0; t0 < segk; t0 += U * 64u) {
                KidGroup<U> g;
                group_front(WA, sh, nb, segk, t0, seg_clean, g, n_bad);
                if (!prefetched) { prefetch(); prefetched = true; }
                if constexpr (MINLOC) {
                    had |= back_deferred(g, i);
                    if (qn >= KID_CQ_FLUSH) resolve_all(i, i & 63u); // the read stays open: more groups may follow
                } else {
                    group_back(g, final_t, vfrow);
                }
            }
            __builtin_amdgcn_wave_barrier(); // strip is rewritten by the next segment / read
        }
        if constexpr (MINLOC) {
            n_lookups += (nk > 0 ? (uint32_t)nk : 0u) - n_bad;
            if (!had) commit_zero(i); // (else the resolver commits it, now that it is closed)
        } else {
            finish_read(r, final_t, (nk > 0 ? (uint32_t)nk : 0u) - n_bad);
        }
        if (!prefetched) prefetch();
    };

    auto fetch_desc = [&](uint32_t r, KidReadDesc &d) {
        d.first_base = 0; d.n_kmers = 0; d.pad = 0;
        if (r < (uint32_t)b.n) {
            if (b.desc) d = descs[r];
            else { d.first_base = (rare->read0 + r) * (unsigned long long)rare->fixed_len; d.n_kmers = rare->fixed_nk; } // fixed layout
        }
    };
    // the first segment of a read as text: lane c asks for the 16 bytes of chunk c.  Unconditional -- a fixed number of
    // loads keeps the compiler's vmcnt bookkeeping exact, so the waits for older loads do not drain these -- but never
    // beyond the chunk that holds the segment's last base: the lanes past it ask for its first chunk again (their words
    // are never looked at).  Wave-uniform base + 32-bit lane offset: the scalar-base addressing form
    auto fetch_words = [&](const KidReadDesc &d, kid_u4 &raw) {
        const uint32_t segk = d.n_kmers > KID_SEG_KMERS ? (uint32_t)KID_SEG_KMERS : d.n_kmers > 0 ? (uint32_t)d.n_kmers : 0u;
        const uint32_t nchunks = (((uint32_t)d.first_base & 15u) + segk + (uint32_t)k - 1u + 15u) >> 4;
        raw = *reinterpret_cast<const kid_u4 *>(reinterpret_cast<const char *>(b.bases + 16ull * (d.first_base >> 4)) + (lane < nchunks ? lane : 0u) * 16u);
    };
    auto uniform64 = [](uint64_t v) {
        return ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v);
    };
    // ---- phase 1: every read of one segment (<= 960 k-mers: all short-read data).  Longer ones are
    // left to phase 2, so that their loop nest and 64-bit bookkeeping stay out of this loop's registers
    bool any_long = false;
    auto short_read = [&](const uint32_t r, const uint32_t i, const uint64_t first, const int32_t nk, const kid_u4 st_raw,
                          auto &&prefetch) {
        if (nk > KID_SEG_KMERS) { any_long = true; rb_skip |= 1ull << (i & 63u); prefetch(); return; }
        bool prefetched = false, had = false;
        uint32_t final_t = 0;
        uint32_t vfrow = 0, n_bad = 0;
        if (nk > 0) {
            const uint32_t sh = (uint32_t)first & 15u, segk = (uint32_t)nk, nb = segk + (uint32_t)k - 1;
            uint32_t st_codes, st_inv;
            kid_pack16(make_uint4(st_raw.x, st_raw.y, st_raw.z, st_raw.w), db.u_is_t, st_codes, st_inv);
            const bool seg_clean = stage(WA, st_codes, st_inv);
            for (uint32_t t0 = 0; t0 < segk; t0 += U * 64u) {
                KidGroup<U> g;
                group_front(WA, sh, nb, segk, t0, seg_clean, g, n_bad);
                if (!prefetched) { prefetch(); prefetched = true; }
                if constexpr (MINLOC) {
                    had |= back_deferred(g, i);
                    if (qn >= KID_CQ_FLUSH) resolve_all(i, i & 63u); // the read stays open: more groups may follow
                } else {
                    group_back(g, final_t, vfrow);
                }
            }
            __builtin_amdgcn_wave_barrier(); // strip is rewritten by the next read
        }
        if constexpr (MINLOC) {
            n_lookups += (nk > 0 ? (uint32_t)nk : 0u) - n_bad;
            if (!had) commit_zero(i); // (else the resolver commits it, now that it is closed)
        } else {
            finish_read(r, final_t, (nk > 0 ? (uint32_t)nk : 0u) - n_bad);
        }
        if (!prefetched) prefetch();
    };
    uint32_t reads_done = gw < b.n ? (uint32_t)((b.n - gw + nw - 1) / nw) : 0u; // the wave's strided share (the pair kernel counts its own)
    // ---- batches whose reads all fit one group (<= U*64 k-mers: Illumina reads): hand-pipelined pairs.
    // A wave is a chain of dependent round trips (strip, header, hit cell, ancestor row) that eight waves
    // per SIMD do not cover, so two reads travel together: front half of A, front half of B (four header
    // loads per lane in flight), the packed words of the next pair, then the back halves.  The compiler's
    // waitcnt insertion drains vmcnt(0) whenever loads sit behind a branch or a store is pending, which
    // serialises exactly these round trips; the loads of this loop are therefore issued from inline
    // assembly, unconditionally (idle lanes read cell 0), and waited for with explicit counts.  Loads
    // return in order, and anything the compiler issues in between (hit cells, atomics, the out_final
    // store) only makes an explicit count stricter than needed.
    if constexpr (PAIRK != 0) {
        uint32_t cnt = 0, duo_first = 0; // duo kernel: the wave's reads are duo_first .. duo_first + cnt - 1
        if (PAIRK == 2) {
            // the duo kernel's waves take one contiguous range of reads each, in the pair kernel's shrinking shares (see
            // switch_block): KID_TAPER units of reads per wave in the first half of the workgroups, one in the second
            const uint32_t nb2 = (uint32_t)b.n, G = gridDim.x, half = G >> 1, wg = blockIdx.x;
            const uint32_t units = wpb * (KID_TAPER * half + (G - half));
            const uint32_t unit = (nb2 + units - 1u) / units;
            const uint32_t mine = wg < half ? KID_TAPER : 1u;
            const uint64_t ufirst = (uint64_t)wpb * (wg < half ? KID_TAPER * wg : KID_TAPER * half + (wg - half)) + (uint64_t)wib * mine;
            const uint64_t f64 = ufirst * unit;
            duo_first = f64 < nb2 ? (uint32_t)f64 : nb2;
            cnt = nb2 - duo_first < mine * unit ? nb2 - duo_first : mine * unit;
            rd_first = duo_first;
            rd_stride = 1u;
            reads_done = cnt;
        }
        // ---- the read text.  A read of the pair kernel (<= 128 k-mers, k <= 31, behind a shift of <= 15 bases) lies
        // in the first 176 bytes from its 16-byte boundary, one of the duo kernel (<= 256 k-mers) in the first 304:
        // lane l loads bytes [BPL l, BPL l + BPL) of them -- BPL = 4 bases per lane (pair) or 8 (duo), one load
        // instruction per read -- and only the lanes whose bytes hold bases of the read ask (exec is narrowed inside the
        // statement: the compiler must not see a load in a branch of its own).  Nothing beyond the dword that holds the
        // read's last classified base is touched, so the caller's buffer needs no padding beyond its own 16 bytes.
        constexpr uint32_t BPL = PAIRK == 1 ? 4u : 8u;
        typedef typename std::conditional<PAIRK == 1, uint32_t, kid_u2>::type Raw;
        const uint32_t lane_off = lane * BPL;
        auto ask_lanes = [&](const uint32_t sh, const uint32_t nk) -> uint64_t { // the lanes that hold bases of a read (never none: a load always goes out)
            const uint32_t nl = nk ? (sh + nk + (uint32_t)k - 1u + BPL - 1u) / BPL : 1u; // <= 44 (pair), 38 (duo)
            return (1ull << nl) - 1ull;
        };
        // descriptors, 64 at a time: lane l holds the one of read number blk + l of this wave -- the low word
        // of first_base, and its high word (< 2^16: a batch is smaller than 2^48 bytes) with n_kmers (<= 256
        // in these kernels) above it.  Fixed layout: worked out on the spot, there are no descriptors in memory.
        uint32_t blk = 0, dv_lo = 0, dv_hn = 0;
        auto load_descs = [&](const uint32_t first, const uint32_t len) { // reads first .. first + len - 1 of the launch -> lanes 0 .. len - 1
            KidReadDesc d;
            d.first_base = 0; d.n_kmers = 0; d.pad = 0;
            const uint32_t rr = first + lane;
            const KidReadDesc *const dp = static_cast<const KidReadDesc *>(rare->desc);
            if (lane < len) {
                if (dp) d = dp[rr];
                else { d.first_base = (rare->read0 + rr) * (unsigned long long)rare->fixed_len; d.n_kmers = rare->fixed_nk; }
            }
            dv_lo = (uint32_t)d.first_base;
            dv_hn = ((uint32_t)(d.first_base >> 32) & 0xFFFFu) | ((d.n_kmers > 0 ? (uint32_t)d.n_kmers : 0u) << 16);
            if (PAIRK == 1 && lane < len) RG[(blk + lane) & 127u] = rr;
        };
        auto issue_words = [&](const uint32_t idx, Raw &x) {
            const uint32_t hn = (uint32_t)__builtin_amdgcn_readlane((int)dv_hn, (int)idx); // [15:0] first_base >> 32, [30:16] n_kmers
            const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)dv_lo, (int)idx);
            const uint64_t w0 = (((uint64_t)(hn & 0xFFFFu) << 32) | lo) >> 4;
            // (said once more that this is wave-uniform: under register pressure hipcc has moved such address arithmetic
            //  to the vector ALU and then handed the "s" operand below a VGPR pair -- a build error, seen once)
            const uint8_t *const p = reinterpret_cast<const uint8_t *>(uniform64((uint64_t)(b.bases + 16ull * w0)));
            const uint64_t lanes = ask_lanes(lo & 15u, (hn >> 16) & 0x7FFFu);
            uint64_t save;
            if constexpr (PAIRK == 1)
                asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tglobal_load_dword %0, %2, %3\n\ts_mov_b64 exec, %1"
                             : "+v"(x), "=&s"(save) : "v"(lane_off), "s"(p), "s"(lanes) : "memory");
            else
                asm volatile("s_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\tglobal_load_dwordx2 %0, %2, %3\n\ts_mov_b64 exec, %1"
                             : "+v"(x), "=&s"(save) : "v"(lane_off), "s"(p), "s"(lanes) : "memory");
        };
        // ASCII -> packed words, in registers.  A lane packs its own 4 (8) bases (kid_codes4), shifts them to their place
        // in the read's 16-base word, and the 4 (2) lanes of a word OR their parts together over DPP quad permutes:
        // afterwards every lane of the group holds the whole word, which is where group_front's ds_bpermute looks for
        // it.  Returns "no base of the read resets a window": decided on the scalar unit from the ballot of the lanes
        // whose bytes hold something that is not ACGT -- the lanes that did not ask hold stale text, and the first and
        // the last lane of a read also hold bytes in front of / behind it (the neighbouring read; the line ends of a
        // FASTQ block), which must not count.  Only for a read that does have such a base are the 16-bit masks of its
        // words put together (wi; else 0: nobody looks).
        const uint32_t part_c = PAIRK == 1 ? 24u - 8u * (lane & 3u) : 16u - 16u * (lane & 1u); // where a lane's codes go in its word
        auto pack_lanes = [&](const Raw x, const uint32_t sh, const uint32_t nk, uint32_t &wc, uint32_t &wi) -> bool {
            uint32_t c, d0, d1 = 0;
            if constexpr (PAIRK == 1) c = kid_codes4(x, db.u_is_t, d0);
            else c = (kid_codes4(x.x, db.u_is_t, d0) << 8) | kid_codes4(x.y, db.u_is_t, d1);
            c <<= part_c;
            c |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0xB1, 0xF, 0xF, false);   // quad_perm:[1,0,3,2]
            if constexpr (PAIRK == 1) c |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x4E, 0xF, 0xF, false); // quad_perm:[2,3,0,1]
            wc = c;
            wi = 0;
            const uint64_t bad = __ballot((d0 | d1) != 0) & ask_lanes(sh, nk);
            if (bad == 0 || nk == 0) return true;
            // (rare, wave-uniform from here on) bytes [sh, end) of the loaded text are the read's
            const uint32_t end = sh + nk + (uint32_t)k - 1u, lf = sh / BPL, ll = (end - 1u) / BPL;
            uint64_t inside = bad & ~((1ull << lf) | (1ull << ll));
            if (inside == 0) { // only the two lanes at the read's ends: look at their bytes
                auto lane_bytes = [&](const uint32_t l) -> uint64_t {
                    return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)d1, (int)l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)d0, (int)l);
                };
                const uint64_t all = BPL == 4u ? 0xFFFFFFFFull : ~0ull;
                const uint64_t from = all << (8u * (sh % BPL)), upto = all >> (8u * (BPL - 1u - ((end - 1u) % BPL)));
                inside = lf == ll ? (lane_bytes(lf) & from & upto) : ((lane_bytes(lf) & from) | (lane_bytes(ll) & upto));
            }
            if (inside == 0) return true;
            uint32_t iv = PAIRK == 1 ? kid_inv4(d0) << (4u * (lane & 3u)) : (kid_inv4(d0) | (kid_inv4(d1) << 4)) << (8u * (lane & 1u));
            iv |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)iv, 0xB1, 0xF, 0xF, false);
            if constexpr (PAIRK == 1) iv |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)iv, 0x4E, 0xF, 0xF, false);
            wi = iv;
            return false;
        };
        auto issue_header = [&](const KidGroup<U> &g, const int u) -> kid_u4 {
            const uint4 *const p = db.table + (g.act[u] ? g.hlo[u] * KID_LINE_CELLS : 0u);
            kid_u4 v;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory");
            return v;
        };
        Raw xA{}, xB{};
        if constexpr (PAIRK == 2) {
            load_descs(duo_first, cnt < 64u ? cnt : 64u);
            issue_words(0u, xA);
            issue_words(1u, xB);
            // ---- reads of two groups (129..256 k-mers: 2 x 250 bp): the pair is the two groups of ONE read.
            // They share the read's words, their queue entries carry the same tag, and the read stays open in the
            // resolver between them.  Two text registers alternate, each requested two trips ahead (a trip is one
            // read here).
            auto duo = [&](const uint32_t i, Raw &xC) { // xC: the register that holds the text of read i
                const uint32_t ia = i - blk;
                const uint32_t hnC = (uint32_t)__builtin_amdgcn_readlane((int)dv_hn, (int)ia);
                const uint32_t nk = (hnC >> 16) & 0x7FFFu;
                const uint32_t sh = (uint32_t)__builtin_amdgcn_readlane((int)dv_lo, (int)ia) & 15u;
                const uint32_t nb = nk + (uint32_t)k - 1;
                KidGroup<U> gA, gB;
                uint32_t bad = 0;
                if (ia + 2u > 63u) { // (this read's descriptor is in scalars by now)
                    blk = i + 1u;
                    load_descs(duo_first + blk, cnt - blk < 64u ? cnt - blk : 64u);
                }
                asm volatile("s_waitcnt vmcnt(1)" : "+v"(xC) : : "memory"); // behind: the text of the next read (other register)
                uint32_t cC, iC;
                const bool cl = pack_lanes(xC, sh, nk, cC, iC);
                group_front(nullptr, sh, nb, nk, 0u, cl, gA, bad, false, cC, iC);
                kid_u4 hA0 = issue_header(gA, 0), hA1 = issue_header(gA, 1);
                group_front(nullptr, sh, nb, nk, (uint32_t)(U * 64), cl, gB, bad, false, cC, iC);
                kid_u4 hB0 = issue_header(gB, 0), hB1 = issue_header(gB, 1);
                issue_words(i + 2u - blk, xC); // this register is used up: the read after the next, two trips ahead
                n_lookups += nk - bad;
                asm volatile("s_waitcnt vmcnt(3)" : "+v"(hA0), "+v"(hA1) : : "memory"); // behind: the headers of B, the text just requested
                gA.hd[0] = make_uint4(hA0.x, hA0.y, hA0.z, hA0.w);
                gA.hd[1] = make_uint4(hA1.x, hA1.y, hA1.z, hA1.w);
                bool had = back_deferred(gA, i);
                if (qn >= KID_CQ_FLUSH) resolve_all(i, i & 63u); // the read stays open
                asm volatile("s_waitcnt vmcnt(1)" : "+v"(hB0), "+v"(hB1) : : "memory"); // behind: the text just requested
                gB.hd[0] = make_uint4(hB0.x, hB0.y, hB0.z, hB0.w);
                gB.hd[1] = make_uint4(hB1.x, hB1.y, hB1.z, hB1.w);
                had |= back_deferred(gB, i);
                if (!had) commit_zero(i);
                else if (qn >= KID_CQ_FLUSH) resolve_all(i, 0xFFFFFFFFu);
                if (((i + 1u) & 63u) == 0u) { // tags and result slots are read numbers mod 64
                    if (qn || cur_tag != 0xFFFFFFFFu) resolve_all(i, 0xFFFFFFFFu);
                    flush_results(i + 1u - 64u, 64u);
                }
            };
            for (uint32_t i = 0; i < cnt; i += 2) {
                duo(i, xA);
                if (i + 1u < cnt) duo(i + 1u, xB);
            }
            if (qn || cur_tag != 0xFFFFFFFFu) resolve_all(cnt - 1u, 0xFFFFFFFFu);
            if (cnt & 63u) flush_results(cnt & ~63u, cnt & 63u);
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(xA), "+v"(xB) : : "memory");
        } else {
        // ---- the pair kernel.  A wave takes blocks of 64 CONSECUTIVE reads: the text (9.6 KB at 150 bp) and the results
        // (256 bytes) of a block are lines that no other wave -- no other L2 -- touches; its blocks are one contiguous
        // share of the launch.  Shares shrink: the first half of the workgroups (dispatched first) take KID_TAPER units
        // of reads per wave, the second half one.  Workgroups take 200-330 us for the same number of reads (the SIMDs
        // serve their oldest waves first), and a launch ends when the slowest of its last workgroups does: with equal
        // shares the chip idles through half the spread of a 250-us workgroup at the end of a 1-ms launch, with small
        // last shares through a third of that (profiles/r02/ab_taper.txt, ab_block_cyclic.txt).
        const uint32_t n32 = (uint32_t)b.n;
        uint32_t seq = 0;        // wave-local number of the pair's first read (even)
        uint32_t blen = 0;       // reads in the current block [blk, blk + blen)
        uint32_t nreal = 0;      // reads classified
        uint32_t sblk = 0;       // number of the wave's next block
        // the next block (length 0: none) and its descriptors: lane l holds the one of wave-local read blk + l
        auto switch_block = [&]() {
            const uint32_t G = gridDim.x, half = G >> 1, wg = blockIdx.x;
            const uint32_t units = wpb * (KID_TAPER * half + (G - half));    // shares of one unit per wave
            const uint32_t unit = (((n32 + units - 1u) / units) + 1u) & ~1u; // reads per unit (even)
            const uint32_t mine = wg < half ? KID_TAPER : 1u;
            const uint64_t ufirst = (uint64_t)wpb * (wg < half ? KID_TAPER * wg : KID_TAPER * half + (wg - half)) + (uint64_t)wib * mine;
            const uint64_t wave_first = ufirst * unit, off = (uint64_t)sblk * 64u, wave_cnt = (uint64_t)mine * unit;
            sblk++;
            const uint64_t f64 = wave_first + off;
            const uint32_t first = f64 < n32 ? (uint32_t)f64 : n32;
            uint32_t len = off < wave_cnt ? (uint32_t)(wave_cnt - off < 64u ? wave_cnt - off : 64u) : 0u;
            if (len > n32 - first) len = n32 - first;
            load_descs(first, len);
            blen = len;
        };
        switch_block();
        bool more = blen != 0;
        if (more) {
            issue_words(0u, xA);
            issue_words(1u, xB);
        }
        while (more) {
            const uint32_t i = seq;
            const uint32_t ia = i - blk;
            const uint32_t hnA = (uint32_t)__builtin_amdgcn_readlane((int)dv_hn, (int)ia), hnB = (uint32_t)__builtin_amdgcn_readlane((int)dv_hn, (int)(ia + 1u));
            const uint32_t nkA = (hnA >> 16) & 0x7FFFu, nkB = (hnB >> 16) & 0x7FFFu;
            const uint32_t shA = (uint32_t)__builtin_amdgcn_readlane((int)dv_lo, (int)ia) & 15u;
            const uint32_t shB = (uint32_t)__builtin_amdgcn_readlane((int)dv_lo, (int)(ia + 1u)) & 15u;
            const bool realB = ia + 1u < blen; // (a block with an odd number of reads -- the last of a share: B is a phantom of zero k-mers)
            KidGroup<U> gA, gB;
            uint32_t badA = 0, badB = 0;
            // the block's last pair: the next block's descriptors (this pair's are in scalars by now)
            if (ia + 2u >= blen) {
                blk = i + 2u;
                switch_block(); // (no block: all-zero descriptors, the requests below fetch four bytes nobody needs)
                more = blen != 0;
            }

            // The text of the next pair is requested as soon as this pair's is used up, a whole trip ahead:
            // it comes from HBM (a batch is larger than the L2).
            asm volatile("s_waitcnt vmcnt(1)" : "+v"(xA) : : "memory"); // behind: the text of B
            uint32_t cA, iA;
            const bool clA = pack_lanes(xA, shA, nkA, cA, iA); // no base of the read resets a window
            group_front(nullptr, shA, nkA + (uint32_t)k - 1, nkA, 0u, clA, gA, badA, false, cA, iA);
            kid_u4 hA0 = issue_header(gA, 0), hA1 = issue_header(gA, 1);
            issue_words(i + 2u - blk, xA);

            asm volatile("s_waitcnt vmcnt(3)" : "+v"(xB) : : "memory"); // behind: headers of A, next text of A
            uint32_t cB, iB;
            const bool clB = pack_lanes(xB, shB, nkB, cB, iB);
            group_front(nullptr, shB, nkB + (uint32_t)k - 1, nkB, 0u, clB, gB, badB, false, cB, iB);
            kid_u4 hB0 = issue_header(gB, 0), hB1 = issue_header(gB, 1);
            issue_words(i + 3u - blk, xB);

            asm volatile("s_waitcnt vmcnt(4)" : "+v"(hA0), "+v"(hA1) : : "memory"); // behind: next text of A, headers and next text of B
            gA.hd[0] = make_uint4(hA0.x, hA0.y, hA0.z, hA0.w);
            gA.hd[1] = make_uint4(hA1.x, hA1.y, hA1.z, hA1.w);
            n_lookups += nkA - badA;
            if (!back_deferred(gA, i)) commit_zero(i);
            else if (qn >= KID_CQ_FLUSH) resolve_all(i, 0xFFFFFFFFu);

            asm volatile("s_waitcnt vmcnt(1)" : "+v"(hB0), "+v"(hB1) : : "memory"); // behind: the next text of B
            gB.hd[0] = make_uint4(hB0.x, hB0.y, hB0.z, hB0.w);
            gB.hd[1] = make_uint4(hB1.x, hB1.y, hB1.z, hB1.w);
            nreal += realB ? 2u : 1u;
            if (realB) {
                n_lookups += nkB - badB;
                if (!back_deferred(gB, i + 1u)) commit_zero(i + 1u);
                else if (qn >= KID_CQ_FLUSH) resolve_all(i + 1u, 0xFFFFFFFFu);
            }
            seq = i + 2u;
            if ((seq & 63u) == 0u) { // tags and result slots are wave-local read numbers mod 64
                if (qn) resolve_all(i + 1u, 0xFFFFFFFFu);
                flush_results(seq - 64u, 64u);
            }
        }
        // (the two requests behind the last pair fetched text nobody needs)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(xA), "+v"(xB) : : "memory");
        // a block of odd length ends the wave's work: seq counts its phantom, the flush below must not
        if (nreal & 1u) seq -= 1u;
        reads_done = seq;
        if (qn) resolve_all(seq - 1u, 0xFFFFFFFFu);
        if (seq & 63u) flush_results(seq & ~63u, seq & 63u);
        }
    }
    if constexpr (PAIRK == 0) {
        // Software pipeline, unrolled by two with two named register sets (A, B) so that nothing is
        // copied between stages: the descriptor of a read is requested two reads ahead (scalar loads)
        // and its first packed words one read ahead; the waits the compiler places in front of their
        // first use land behind a whole read's worth of work.  32-bit read indices (n < 2^31).
        const uint32_t n32 = (uint32_t)b.n;
        // the wave's reads: rd_first + i rd_stride, i < gen_cnt -- a strided share of the batch, or (KID_TAPER, minimizer-
        // localised table) one contiguous range in the pair kernel's shrinking shares
        gen_cnt = gw32 < n32 ? (n32 - gw32 + nw32 - 1u) / nw32 : 0u;
        if (MINLOC) {
            const uint32_t G = gridDim.x, half = G >> 1, wg = blockIdx.x;
            const uint32_t units = wpb * (KID_TAPER * half + (G - half));
            const uint32_t unit = (n32 + units - 1u) / units;
            const uint32_t mine = wg < half ? KID_TAPER : 1u;
            const uint64_t ufirst = (uint64_t)wpb * (wg < half ? KID_TAPER * wg : KID_TAPER * half + (wg - half)) + (uint64_t)wib * mine;
            const uint64_t f64 = ufirst * unit;
            rd_first = f64 < n32 ? (uint32_t)f64 : n32;
            rd_stride = 1u;
            gen_cnt = n32 - rd_first < mine * unit ? n32 - rd_first : mine * unit;
            reads_done = gen_cnt;
        }
        const uint32_t gf = rd_first, gs = rd_stride, gc = gen_cnt;
        KidReadDesc dA, dB;
        kid_u4 xA, xB;
        fetch_desc(gc > 0u ? gf : n32, dA);
        fetch_desc(gc > 1u ? gf + gs : n32, dB);
        fetch_words(dA, xA);
        for (uint32_t i = 0; i < gc; i += 2) { // i: number of the read within this wave
            const uint32_t r = gf + i * gs;
            short_read(r, i, uniform64(dA.first_base), __builtin_amdgcn_readfirstlane(dA.n_kmers), xA,
                       [&]() { fetch_words(dB, xB); fetch_desc(i + 2u < gc ? r + 2u * gs : n32, dA); });
            if (i + 1u >= gc) break;
            short_read(r + gs, i + 1u, uniform64(dB.first_base), __builtin_amdgcn_readfirstlane(dB.n_kmers), xB,
                       [&]() { fetch_words(dA, xA); fetch_desc(i + 3u < gc ? r + 3u * gs : n32, dB); });
            if (MINLOC && ((i + 2u) & 63u) == 0u) { // tags and result slots are read numbers mod 64
                if (qn || cur_tag != 0xFFFFFFFFu) resolve_all(i + 1u, 0xFFFFFFFFu);
                flush_results(i + 2u - 64u, 64u);
            }
        }
        if (MINLOC) {
            const uint32_t cnt = gc;
            if (qn || cur_tag != 0xFFFFFFFFu) resolve_all(cnt - 1u, 0xFFFFFFFFu);
            if (cnt & 63u) flush_results(cnt & ~63u, cnt & 63u);
        }
    }
    // ---- phase 2: the long reads this wave met (FASTA records, long-read data), one at a time
    if (PAIRK == 0 && any_long) {
        rb_direct = true; // the other reads' results are stored already: these go out one by one
        for (uint32_t i = 0; i < gen_cnt; i++) {
            const uint64_t r = (uint64_t)rd_first + (uint64_t)i * rd_stride;
            KidReadDesc d;
            fetch_desc((uint32_t)r, d);
            const int64_t nk = (int64_t)__builtin_amdgcn_readfirstlane(d.n_kmers);
            if (nk <= KID_SEG_KMERS) continue;
            kid_u4 x0;
            fetch_words(d, x0);
            process_read(r, i, uniform64(d.first_base), nk, x0, []() {});
            if (MINLOC && (qn || cur_tag != 0xFFFFFFFFu)) resolve_all(i, 0xFFFFFFFFu);
        }
    }
    if (MINLOC && lane == 0) atomicAdd(WL, (unsigned long long)n_lookups);
    if (MINLOC && n_zero && lane == 0) {
        if (HIST) atomicAdd(&hist[0], n_zero); // low half = target 0; a workgroup stays below 65536 reads
        else atomicAdd(&rare->gcount[0], (unsigned long long)n_zero);
    }
    if (!HIST && pend_n && lane == 0) atomicAdd(&rare->gcount[pend_t], (unsigned long long)pend_n);

    // ---- flush (with the minimizer-localised table the histogram packs two 16-bit counters per word:
    // the host keeps a workgroup below 65536 reads per launch)
    if (HIST) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < hist_words; i += blockDim.x) {
            const uint32_t v = hist[i];
            if (MINLOC) {
                if (v & 0xFFFFu) atomicAdd(&rare->gcount[2u * i], (unsigned long long)(v & 0xFFFFu));
                if (v >> 16) atomicAdd(&rare->gcount[2u * i + 1u], (unsigned long long)(v >> 16));
            } else if (v) atomicAdd(&rare->gcount[i], (unsigned long long)v);
        }
    }
    if (lane == 0) {
        const unsigned long long tl = *WL, n_reads = reads_done;
        const uint32_t n_hits = WC[2];
        const unsigned long long te = WC[3];
        if (n_reads) atomicAdd(&WGS[0], n_reads);
        if (tl) atomicAdd(&WGS[1], tl);
        if (tl + te) atomicAdd(&WGS[2], tl + te); // cells read: one per lookup plus the probes beyond the first
        if (n_hits) atomicAdd(&WGS[3], (unsigned long long)n_hits);
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const unsigned long long v = WGS[threadIdx.x];
        if (v) atomicAdd(&rare->stats[threadIdx.x], v);
    }
    if (threadIdx.x == 0) {
        // The launch's last workgroup banks the interval ([6] += ticks, [7] += 1) and re-arms the stamps for the next
        // launch.  Everything here is an atomic performed in memory, one after the other (each returns before the next is
        // issued): no fence -- a release / acquire fence at device scope writes back and invalidates the L2 of the
        // wave's XCD (buffer_wbl2 / buffer_inv sc1), and 4096 of those per launch cost 0.2 ms of 1.1
        // (profiles/r03/bench_fence_in_epilogue.json).
        unsigned long long prev = atomicMax(&s.stats[31], (unsigned long long)__builtin_amdgcn_s_memrealtime());
        asm volatile("" : "+v"(prev)); // (the stamp is in memory before this workgroup counts itself out)
        unsigned long long done = atomicAdd(&s.stats[29], 1ull);
        if (done + 1ull == (unsigned long long)gridDim.x) {
            const unsigned long long a = atomicAdd(&s.stats[30], 0ull), z = atomicAdd(&s.stats[31], 0ull);
            if (z > a) { atomicAdd(&s.stats[6], z - a); atomicAdd(&s.stats[7], 1ull); }
            unsigned long long t0 = atomicExch(&s.stats[30], ~0ull), t1 = atomicExch(&s.stats[31], 0ull);
            asm volatile("" : "+v"(t0), "+v"(t1));
            atomicExch(&s.stats[29], 0ull);
        }
    }
}

// ------------------------------------------------------------------ unit probes (parity tests)
__global__ void kid_lookup_kernel(const KidDevDb db, const uint64_t *keys, uint64_t n, uint32_t *targets, uint32_t *probes)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t slot, np;
        targets[i] = kid_dev_lookup(db, keys[i], slot, np);
        if (probes) probes[i] = np;
    }
}

// Hashtable::integerHash (newkmer_10nx.cpp:189-197) as the device computes it
__global__ void kid_fmix_kernel(const uint64_t *keys, uint64_t n, uint64_t *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = kid_fmix64(keys[i]);
}

__global__ void kid_msca_kernel(const KidDevDb db, const int32_t *x, const int32_t *y, uint64_t n, int32_t *out)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = (uint32_t)x[i], c = (uint32_t)y[i];
        uint32_t r;
        if (db.rows) {
            uint4 ro;
            r = kid_msca_rows(a, db.rows[a], c, db.rows[c], ro);
        } else {
            r = kid_msca_climb(db, a, c);
        }
        out[i] = (int32_t)r;
    }
}

// process_qual, newkmer_10nx.cpp:714-760: the quality string of one read (len = length of its SEQUENCE, :716) ->
// [start, stop].  A sequential scan by nature: one read per thread.
__device__ __forceinline__ void kid_process_qual(const signed char *q, const int len, int &start, int &stop)
{
    start = 0;
    stop = len - 1;
    if (len <= 0) return;
    while (q[start] < 49 && start < stop) start++;
    while (q[stop] < 49 && stop > start) stop--;
    if (start < stop - 4) {
        int w = 0;
        for (int i = 0; i < 4; i++) w += q[start + i] - 32;
        while (w < 68 && start < stop - 4) { w += q[start + 4] - q[start]; start++; }
    }
    if (start < stop - 4) {
        int w = 0;
        for (int i = 0; i < 4; i++) w += q[stop - i] - 32;
        while (w < 68 && start < stop - 4) { w += q[stop - 4] - q[stop]; stop--; }
    }
}

__global__ void kid_trim_kernel(const uint8_t *quals, const uint64_t *offsets, uint64_t n, int k,
                                int32_t *start_out, int32_t *stop_out, uint8_t *keep)
{
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const int len = (int)(offsets[r + 1] - offsets[r]);
        int start, stop;
        kid_process_qual(reinterpret_cast<const signed char *>(quals + offsets[r]), len, start, stop);
        start_out[r] = start;
        stop_out[r] = stop;
        keep[r] = (len > 0 && stop - start >= k) ? 1 : 0;
    }
}

// A block of FASTQ text whose lines the host has found (kid_classify_fastq_async): process_qual + the ">= k" test
// of :757 + the read descriptor, per record.  A record process_qual drops is not handed to process_read in the
// reference, i.e. it is counted nowhere: the classify kernels see it as a read without k-mers (gcount[0]++), which
// this kernel takes back (stats[5] counts them).  stats[8]: records whose quality line is shorter than the sequence
// (qual.at() throws, :727).
struct KidFastqRec {
    uint32_t seq_off, seq_len, qual_off, qual_len; // byte offsets into the block's text
};
__global__ void kid_prepare_fastq_kernel(const uint8_t *text, const KidFastqRec *recs, uint64_t n, int k, KidReadDesc *desc,
                                         int32_t *start_out, int32_t *stop_out, uint32_t *out_final, unsigned long long *stats,
                                         unsigned long long *gcount, KidRareArgs *rare, uint32_t seq, int rebase)
{
    if (rebase && blockIdx.x == 0) kid_rebase(rare, desc, out_final, 0ull, 0u, 0);
    uint32_t mx = 0, dropped = 0, bad = 0;
    for (uint64_t r = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; r < n; r += (uint64_t)gridDim.x * blockDim.x) {
        const KidFastqRec rc = recs[r];
        int start = 0, stop = -1;
        if (rc.qual_len < rc.seq_len) bad++;
        else kid_process_qual(reinterpret_cast<const signed char *>(text + rc.qual_off), (int)rc.seq_len, start, stop);
        const bool keep = rc.seq_len > 0 && rc.qual_len >= rc.seq_len && stop - start >= k;
        KidReadDesc d;
        d.first_base = (uint64_t)rc.seq_off + (uint64_t)(start > 0 ? start : 0);
        d.n_kmers = keep ? stop - start + 1 - (k - 1) : 0;
        d.pad = 0;
        desc[r] = d;
        start_out[r] = start;
        stop_out[r] = stop;
        if (!keep) dropped++;
        if (d.n_kmers > 0 && (uint32_t)d.n_kmers > mx) mx = (uint32_t)d.n_kmers;
    }
    __shared__ uint32_t s_mx, s_dropped, s_bad;
    if (threadIdx.x == 0) { s_mx = 0; s_dropped = 0; s_bad = 0; }
    __syncthreads();
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t y = (uint32_t)__shfl_xor((int)mx, o);
        mx = y > mx ? y : mx;
        dropped += (uint32_t)__shfl_xor((int)dropped, o);
        bad += (uint32_t)__shfl_xor((int)bad, o);
    }
    if ((threadIdx.x & 63u) == 0) {
        if (mx) atomicMax(&s_mx, mx);
        if (dropped) atomicAdd(&s_dropped, dropped);
        if (bad) atomicAdd(&s_bad, bad);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long v = ((unsigned long long)seq << 32) | s_mx;
        if (v > *reinterpret_cast<volatile unsigned long long *>(&rare->batch_max)) atomicMax(&rare->batch_max, v);
        if (s_dropped) { atomicAdd(&gcount[0], 0ull - (unsigned long long)s_dropped); atomicAdd(&stats[5], (unsigned long long)s_dropped); }
        if (s_bad) atomicAdd(&stats[8], (unsigned long long)s_bad);
    }
}

// ------------------------------------------------------------------ table build on the GPU
// Pass 1: every entry claims the first free cell on its probe path (same path
// as Hashtable::add_kmer, newkmer_10nx.cpp:235-263) with a CAS on the ordinal
// word.  Like the reference there is no key comparison: duplicates take
// separate cells.  Entries with target 0 are skipped: in the reference they
// leave their cell "empty" (value == 0), i.e. invisible to every lookup.
// Pass 1: every entry claims a cell.  Reference geometry: the first free cell on the reference's
// probe path (Hashtable::add_kmer, newkmer_10nx.cpp:235-263), claimed with a CAS on the ordinal
// word; like the reference there is no key comparison, duplicates take separate cells.
// Minimizer-localised geometry: the next free entry of the key's line (count in the header,
// CAS), chaining into the following line when 7 entries are taken; the key's 16-bit fingerprint
// goes into the header.  Entries with target 0 are skipped: in the reference they leave their
// cell "empty" (value == 0), i.e. invisible to every lookup.
__global__ void kid_build_insert_kernel(uint4 *table, uint32_t slot_mask, const uint64_t *keys, const uint32_t *targets,
                                        uint64_t n, uint32_t ntar, unsigned long long *n_occupied, int k, uint32_t minloc,
                                        uint32_t line_shift, uint32_t line_mask)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t t = targets[e];
        if (t == 0) continue;
        if (t >= ntar) { atomicAdd(n_occupied + 1, 1ull); continue; } // reported as KID_ERR_TARGET
        const uint64_t key = keys[e];
        if (minloc) {
            uint32_t line = kid_minloc_line(kid_minimizer_of_key(key, k), line_shift);
            const uint32_t fp = kid_key_fp(key);
            for (;;) {
                uint32_t *hdr = reinterpret_cast<uint32_t *>(table + (uint64_t)line * KID_LINE_CELLS);
                uint32_t old = __hip_atomic_load(hdr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t cnt;
                for (;;) {
                    cnt = kid_hdr_count(old);
                    if (cnt >= KID_HDR_FULL) break;
                    const uint32_t seen = atomicCAS(hdr + 3, old, old + 0x10000u);
                    if (seen == old) break;
                    old = seen;
                }
                if (cnt < KID_LINE_ENTRIES) { // entry number cnt of this line is mine
                    uint32_t *c = reinterpret_cast<uint32_t *>(table + (uint64_t)line * KID_LINE_CELLS + 1u + cnt);
                    c[0] = (uint32_t)key;
                    c[1] = (uint32_t)(key >> 32);
                    c[2] = t;
                    c[3] = (uint32_t)e + 1u;
                    atomicOr(hdr + (cnt >> 1), fp << (16u * (cnt & 1u)));
                    atomicAdd(n_occupied, 1ull);
                    break;
                }
                atomicOr(hdr + 3, 1u << kid_ovf_bit(fp)); // pushed past this line: leave a trace for the lookups
                line = (line + 1u) & line_mask; // cnt == 7 just became 8 (chain marker) or was 8 already
            }
            continue;
        }
        const uint32_t h = (uint32_t)kid_fmix64(key) & slot_mask;
        uint32_t rp = 0, i = 0;
        for (;;) {
            const uint32_t idx = (h + rp) & slot_mask;
            rp += ++i;
            uint32_t *ordp = reinterpret_cast<uint32_t *>(table + idx) + 3;
            if (atomicCAS(ordp, 0u, (uint32_t)e + 1u) == 0u) {
                uint32_t *c = reinterpret_cast<uint32_t *>(table + idx);
                c[0] = (uint32_t)key;
                c[1] = (uint32_t)(key >> 32);
                c[2] = t;
                atomicAdd(n_occupied, 1ull);
                break;
            }
        }
    }
}

// Pass 2: the reference's lookup returns the FIRST-inserted copy of a key (earlier inserts sit
// earlier on the path).  Pass 1 placed duplicate copies in arbitrary order, so every entry walks
// its whole chain, finds the smallest ordinal among the cells holding its key and, unless that is
// its own, turns ITS OWN cell into a tombstone (a key no lookup can ask for: keys are below 2^62).
// Afterwards the first insert is the only copy a lookup can match, wherever it sits -- and the
// ordinal it carries (cell word 3) names the key independently of the placement, which is what
// the per-sample seen-bitmap is indexed by.  A tombstone keeps its cell occupied and its
// fingerprint in the header, like the reference's unreachable duplicates keep theirs.  The first
// insert is never touched, so concurrent walkers always see it; each thread writes one word of
// its own cell only.
__global__ void kid_build_firstwins_kernel(uint4 *table, uint32_t slot_mask, const uint64_t *keys, const uint32_t *targets,
                                           uint64_t n, int k, uint32_t minloc, uint32_t line_shift, uint32_t line_mask)
{
    for (uint64_t e = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; e < n; e += (uint64_t)gridDim.x * blockDim.x) {
        if (targets[e] == 0 || table == nullptr) continue;
        const uint64_t key = keys[e];
        uint32_t my_idx = 0, min_ord = 0xFFFFFFFFu, copies = 0;
        if (minloc) {
            uint32_t line = kid_minloc_line(kid_minimizer_of_key(key, k), line_shift);
            for (;;) {
                const uint32_t base = line * KID_LINE_CELLS;
                const uint32_t w3 = reinterpret_cast<const uint32_t *>(table + base)[3];
                const uint32_t cnt = kid_hdr_count(w3);
                const uint32_t ne = cnt < KID_LINE_ENTRIES ? cnt : KID_LINE_ENTRIES;
                for (uint32_t j = 0; j < ne; j++) {
                    const uint32_t *c = reinterpret_cast<const uint32_t *>(table + base + 1u + j);
                    if (c[0] == (uint32_t)key && c[1] == (uint32_t)(key >> 32)) {
                        const uint32_t ord = c[3];
                        if (ord == (uint32_t)e + 1u) my_idx = base + 1u + j;
                        min_ord = ord < min_ord ? ord : min_ord;
                        copies++;
                    }
                }
                if (!kid_hdr_continues(w3, kid_key_fp(key))) break; // (every copy of this key left its trace on the way)
                line = (line + 1u) & line_mask;
            }
        } else {
            const uint32_t h = (uint32_t)kid_fmix64(key) & slot_mask;
            uint32_t rp = 0, i = 0;
            for (;;) {
                const uint32_t idx = (h + rp) & slot_mask;
                rp += ++i;
                const uint32_t *c = reinterpret_cast<const uint32_t *>(table + idx);
                const uint32_t ord = c[3];
                if (ord == 0) break;
                if (c[0] == (uint32_t)key && c[1] == (uint32_t)(key >> 32)) {
                    if (ord == (uint32_t)e + 1u) my_idx = idx;
                    min_ord = ord < min_ord ? ord : min_ord;
                    copies++;
                }
            }
        }
        if (copies > 1 && min_ord != (uint32_t)e + 1u) reinterpret_cast<uint32_t *>(table + my_idx)[1] = KID_TOMBSTONE_HI;
    }
}

// ------------------------------------------------------------------ very long records (FASTA contigs classified whole)
// A record is one left fold over its hits (newkmer_10nx.cpp:588-595; msca is not associative), which the classify
// kernels run inside ONE wave: 9.7 ms per megabase when the batch holds few records.  When the host sees few long
// records in a batch it hands them to two kernels instead:
//   kid_long_hits_kernel   every k-mer of every long record is looked up by a lane of its own, all over the chip; a hit
//                          leaves its target in hits[position] (and its bit in the seen-bitmap)
//   kid_long_fold_kernel   one workgroup per record compacts the hits in position order and folds them, 64 at a time,
//                          jumping from change to change of the running result like the resolver does
// Plain code: this path runs a few hundred times per batch, not a hundred million times.
// The list -> the plan: where every long record's hits go.  One thread: the list is short.  A record that does not fit
// the hit array any more is handed back to the classify kernels (its descriptor gets its k-mers back).
__global__ void kid_long_plan_kernel(KidLongList *list, KidLongPlan *plan, KidReadDesc *desc, KidRareArgs *rare, uint32_t seq,
                                     uint64_t hits_cap, uint64_t tiles_cap)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t n = list->n < KID_LONG_MAX ? list->n : KID_LONG_MAX;
    uint64_t off = 0, tiles = 0;
    uint32_t m = 0, back = 0;
    for (uint32_t i = 0; i < n; i++) {
        const KidLongList::Item it = list->e[i];
        const uint64_t nt = ((uint64_t)it.n_kmers + 255u) / 256u;
        if (off + it.n_kmers > hits_cap || tiles + nt > tiles_cap) {
            desc[it.read].n_kmers = (int32_t)it.n_kmers;
            back = it.n_kmers > back ? it.n_kmers : back;
            continue;
        }
        KidLongRec r;
        r.first_base = it.first_base; r.hits_off = off; r.n_kmers = it.n_kmers; r.read = it.read; r.tile0 = tiles;
        plan->recs[m++] = r;
        off += it.n_kmers;
        tiles += nt;
    }
    plan->n_recs = m;
    plan->n_tiles = tiles;
    plan->total_kmers = off;
    list->n = 0; // for the batch that uses this set next
    if (back) { // (the kernels pick themselves by the longest read of the batch)
        const unsigned long long v = ((unsigned long long)seq << 32) | back;
        if (v > rare->batch_max) rare->batch_max = v;
    }
}

// Every k-mer of every long record is looked up by a lane of its own.  A tile = 256 consecutive k-mers of one record:
// its 256 + k - 1 bases are read as text (16 bytes per thread by the first 20 threads), packed in registers and staged
// in LDS, like the classify kernels' general loops stage a segment.
__global__ __launch_bounds__(256) void kid_long_hits_kernel(const KidDevDb db, const uint8_t *bases, const KidLongPlan *plan,
                                                             uint32_t *hits, uint8_t *tile_any, uint32_t *seen,
                                                             unsigned long long *stats)
{
    __shared__ uint32_t mm[256 + 32];
    __shared__ uint32_t W[24], IM[24]; // the tile's packed words and invalid masks (20 chunks + what a window reads beyond)
    const int k = db.k;
    const uint32_t win = (uint32_t)kid_min_window(k);
    const int mlen = kid_min_mlen(k);
    const uint32_t n_recs = plan->n_recs;
    const uint64_t n_tiles = plan->n_tiles;
    const KidLongRec *recs = plan->recs;
    unsigned long long n_lookups = 0, n_cells = 0, n_hits = 0;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        // the record this tile belongs to (recs are few: binary search over tile0)
        uint32_t lo = 0, hi = n_recs;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (recs[mid].tile0 <= tile) lo = mid; else hi = mid; }
        const KidLongRec rc = recs[lo];
        const uint32_t t0 = (uint32_t)(tile - rc.tile0) * 256u; // first k-mer of the tile within the record
        const uint32_t j = threadIdx.x;
        // the tile's text: chunks c0 .. of 16 bases, never beyond the chunk that holds the record's last base
        const uint64_t c0 = (rc.first_base + t0) >> 4, c_last = (rc.first_base + (uint64_t)rc.n_kmers + (uint64_t)k - 2u) >> 4;
        if (j < 24u) {
            uint32_t cw = 0, ci = 0;
            if (c0 + j <= c_last && j < 21u) {
                const uint4 v = *reinterpret_cast<const uint4 *>(bases + 16ull * (c0 + j));
                kid_pack16(v, db.u_is_t, cw, ci);
            }
            W[j] = cw;
            IM[j] = ci;
        }
        __syncthreads();
        auto window = [&](uint64_t base) -> uint64_t { // 32 bases starting at `base`, first base in the top bits
            const uint32_t w0 = (uint32_t)((base >> 4) - c0);
            const uint32_t o2 = (uint32_t)(base & 15u) * 2u;
            const uint64_t A = ((uint64_t)W[w0] << 32) | W[w0 + 1];
            const uint64_t B = W[w0 + 2];
            return (A << o2) | ((B << o2) >> 32);
        };
        // hashed m-mers of positions t0 .. t0 + 255 + win - 1 (clamped to the last m-mer inside the record)
        const uint64_t last_m = rc.first_base + (uint64_t)rc.n_kmers + (uint64_t)k - 1u - (uint64_t)mlen;
        if (db.minloc) // (workgroup-uniform; the reference placement has no minimizers, and for k < 15 there is no m-mer to hash)
            for (uint32_t q = j; q < 256u + win - 1u; q += 256u) {
                uint64_t p = rc.first_base + t0 + q;
                p = p < last_m ? p : last_m;
                mm[q] = kid_mmer_hash((uint32_t)(window(p) >> (64 - 2 * mlen)), mlen);
            }
        __syncthreads();
        const uint32_t i = t0 + j;
        bool hit = false;
        uint32_t hit_t = 0;
        if (i < rc.n_kmers) {
            const uint64_t p = rc.first_base + i;
            // a window touching a base that is not ACGTacgt(Uu) holds no k-mer (newkmer_10nx.cpp:520-526,604)
            const uint32_t iw = (uint32_t)((p >> 4) - c0);
            uint64_t im = (uint64_t)IM[iw] | ((uint64_t)IM[iw + 1] << 16) | ((uint64_t)IM[iw + 2] << 32);
            im >>= (p & 15u);
            if ((im & ((1ull << k) - 1ull)) == 0) {
                const uint64_t keyF = window(p) >> (64 - 2 * k);
                const uint64_t key = kid_canonical(keyF, k);
                uint32_t slot = 0, nc = 0, tgt;
                if (db.minloc) {
                    uint32_t g = 0xFFFFFFFFu;
                    for (uint32_t w = 0; w < win; w++) g = mm[j + w] < g ? mm[j + w] : g;
                    tgt = kid_bucket_lookup(db, key, g, slot, nc);
                } else {
                    tgt = kid_dev_lookup(db, key, slot, nc);
                }
                n_lookups++;
                n_cells += nc;
                if (tgt > 0) {
                    n_hits++;
                    hit = true;
                    if (tgt > 1) atomicOr(&seen[slot >> 5], 1u << (slot & 31u));
                }
                hit_t = tgt;
            }
        }
        if (i < rc.n_kmers) hits[rc.hits_off + i] = hit_t; // (every position: the array is not cleared between batches)
        const int any = __syncthreads_or(hit ? 1 : 0); // (also: mm[], W[] and IM[] are free for the next tile)
        if (threadIdx.x == 0) tile_any[tile] = any ? 1 : 0; // the fold skips tiles without hits unseen
    }
    // one set of atomics per workgroup (see kid_classify_kernel)
    __shared__ unsigned long long tot[3];
    if (threadIdx.x < 3) tot[threadIdx.x] = 0;
    __syncthreads();
    if (n_lookups) atomicAdd(&tot[0], n_lookups);
    if (n_cells) atomicAdd(&tot[1], n_cells);
    if (n_hits) atomicAdd(&tot[2], n_hits);
    __syncthreads();
    if (threadIdx.x < 3 && tot[threadIdx.x]) atomicAdd(&stats[1 + threadIdx.x], tot[threadIdx.x]);
}

__global__ __launch_bounds__(256) void kid_long_fold_kernel(const KidDevDb db, const KidLongPlan *plan, const uint32_t *hits,
                                                             const uint8_t *tile_any, unsigned long long *gcount,
                                                             uint32_t *out_final)
{
    if (blockIdx.x >= plan->n_recs) return; // (a grid of KID_LONG_MAX workgroups: the host does not know how many there are)
    const KidLongRec *recs = plan->recs;
    __shared__ uint32_t list[256];
    __shared__ uint32_t wcount[4];
    __shared__ uint8_t flags[256];
    const KidLongRec rc = recs[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t uf = 0; // the running result (wave 0)
    uint4 ufr = make_uint4(0, 0, 0, 0);
    const uint32_t ntile = (rc.n_kmers + 255u) / 256u;
    for (uint32_t tg = 0; tg < ntile; tg += 256u) { // 256 tiles = 65 536 positions at a time: most hold no hit at all
      const uint32_t myt = tg + threadIdx.x;
      const uint8_t fl = myt < ntile ? tile_any[rc.tile0 + myt] : (uint8_t)0;
      flags[threadIdx.x] = fl;
      if (!__syncthreads_or(fl)) continue;
      for (uint32_t tt = 0; tt < 256u && tg + tt < ntile; tt++) {
        if (!flags[tt]) continue; // (workgroup-uniform)
        const uint32_t t0 = (tg + tt) * 256u;
        {
        const uint32_t i = t0 + threadIdx.x;
        const uint32_t h = i < rc.n_kmers ? hits[rc.hits_off + i] : 0u;
        const uint64_t bm = __ballot(h != 0);
        if (lane == 0) wcount[wv] = (uint32_t)__popcll(bm);
        __syncthreads();
        uint32_t before = 0, total = 0;
        for (uint32_t w = 0; w < 4; w++) { const uint32_t c = wcount[w]; if (w < wv) before += c; total += c; }
        if (h != 0) list[before + __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u))] = h;
        __syncthreads();
        if (wv == 0) {
            for (uint32_t c0 = 0; c0 < total; c0 += 64u) {
                const uint32_t n = total - c0 < 64u ? total - c0 : 64u;
                const uint32_t tgt = lane < n ? list[c0 + lane] : 0u;
                uint4 row = make_uint4(0, 0, 0, 0);
                if (db.rows && tgt) row = db.rows[tgt];
                uint64_t rem = n >= 64u ? ~0ull : ((1ull << n) - 1ull);
                while (rem) { // every lane: the step its own hit would make from the current result; jump to the first change
                    uint32_t rj = tgt;
                    uint4 roj = row;
                    if (uf != 0 && tgt != uf && tgt != 0) {
                        if (db.rows) rj = kid_msca_rows(tgt, row, uf, ufr, roj);
                        else rj = kid_msca_climb(db, tgt, uf);
                    }
                    const uint64_t ch = __ballot(rj != uf) & rem;
                    if (!ch) break;
                    const int jj = __builtin_ctzll(ch);
                    uf = (uint32_t)__builtin_amdgcn_readlane((int)rj, jj);
                    ufr.x = (uint32_t)__builtin_amdgcn_readlane((int)roj.x, jj);
                    ufr.y = (uint32_t)__builtin_amdgcn_readlane((int)roj.y, jj);
                    ufr.z = (uint32_t)__builtin_amdgcn_readlane((int)roj.z, jj);
                    ufr.w = (uint32_t)__builtin_amdgcn_readlane((int)roj.w, jj);
                    rem &= jj >= 63 ? 0ull : ~((2ull << jj) - 1ull);
                }
            }
        }
        __syncthreads();
        }
      }
      __syncthreads(); // flags[] is rewritten by the next round
    }
    if (threadIdx.x == 0) {
        // the classify kernels counted the record under target 0 (they saw it without k-mers)
        if (uf != 0) {
            atomicAdd(&gcount[uf], 1ull);
            atomicAdd(&gcount[0], ~0ull); // - 1
        }
        if (out_final) out_final[rc.read] = uf;
    }
}

// ------------------------------------------------------------------ the hit log -> seen-bitmap
// The classify kernels append the entry ordinals of their hits to KID_LOG_SHARDS log regions.  Setting 2.5 M random
// bits of a 13.6 MB bitmap costs one memory-side atomic each however it is done from the classify kernel; here the
// entries are first sorted by bitmap piece (a counting sort: count, scan, scatter), then ONE workgroup per piece sets
// its bits in LDS and ORs the piece into the bitmap as its only writer.  Run every few dozen launches (and before
// anybody reads the bitmap), over everything logged since: ~16 bytes of traffic per logged hit + one pass over the bitmap.
struct KidLogArgs {
    const uint32_t *log;
    const uint32_t *tail;
    uint32_t cap;
    uint32_t nbins;        // pieces of 2^KID_LOG_BIN_BITS bits
    uint32_t *counts;      // [bin][workgroup]: entries of the bin in the workgroup's share of the log
    uint32_t *bin_total;   // [bin]
    uint32_t *sorted;
    uint32_t *seen;
    uint64_t seen_words;
    unsigned long long *host_total; // mapped host memory: log places asked for per 1024 reads, as of this pass (the host paces the passes by it)
    // Many hits per read (reads from genomes the database holds): neighbouring lookups name neighbouring bits, the
    // resolver merges them over DPP and one atomic sets up to 16 -- cheaper than logging every hit and sorting the log
    // (profiles/r03/dense_hits.txt: 15 hits per read 1-2 %, 60 hits 4.5 %, the builder-shaped database 6 %).  The
    // pass decides: more than 8 log places per read of the `reads` it covers, and it takes the log out of the sample's
    // argument blocks -- in stream order, in front of the next launch, without a trip to the host.
    unsigned long long reads;
    KidRareArgs *blocks[4];
    unsigned int *off_flag;         // mapped host memory: set when the pass has switched the log off
};
// the share of the log a counting / scattering workgroup owns: region blockIdx / 16, 1/16 of its entries (whole
// groups of 64 entries: a share starts on a 16-byte boundary)
__device__ __forceinline__ void kid_log_share(const KidLogArgs &a, uint32_t &begin, uint32_t &end)
{
    const uint32_t per = KID_LOG_WGS / KID_LOG_SHARDS, sh = blockIdx.x / per, c = blockIdx.x % per;
    const uint32_t t = a.tail[sh * 16u], filled = t < a.cap ? t : a.cap;
    const uint32_t len = ((filled + per - 1u) / per + 63u) & ~63u;
    begin = sh * a.cap + (c * len < filled ? c * len : filled);
    end = sh * a.cap + ((c + 1u) * len < filled ? (c + 1u) * len : filled);
}
// every entry of [b0, b1) to f, four at a time (16-byte loads: these kernels are a stream over the log)
template <class F>
__device__ __forceinline__ void kid_log_for_each(const uint32_t *log, uint32_t b0, uint32_t b1, F &&f)
{
    const uint32_t n4 = (b1 - b0) >> 2;
    const uint4 *p = reinterpret_cast<const uint4 *>(log + b0);
    for (uint32_t i = threadIdx.x; i < n4; i += blockDim.x) {
        const uint4 v = p[i];
        if (v.x != KID_LOG_NONE) f(v.x);
        if (v.y != KID_LOG_NONE) f(v.y);
        if (v.z != KID_LOG_NONE) f(v.z);
        if (v.w != KID_LOG_NONE) f(v.w);
    }
    for (uint32_t i = b0 + (n4 << 2) + threadIdx.x; i < b1; i += blockDim.x) {
        const uint32_t e = log[i];
        if (e != KID_LOG_NONE) f(e);
    }
}
__global__ __launch_bounds__(256) void kid_seenlog_count_kernel(const KidLogArgs a)
{
    extern __shared__ uint32_t kid_lh[];
    for (uint32_t i = threadIdx.x; i < a.nbins; i += blockDim.x) kid_lh[i] = 0;
    __syncthreads();
    uint32_t b0, b1;
    kid_log_share(a, b0, b1);
    kid_log_for_each(a.log, b0, b1, [&](const uint32_t e) { atomicAdd(&kid_lh[e >> KID_LOG_BIN_BITS], 1u); });
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < a.nbins; i += blockDim.x) a.counts[i * KID_LOG_WGS + blockIdx.x] = kid_lh[i];
}
// per bin: exclusive scan of the workgroups' counts (in place) and the bin's total
__global__ __launch_bounds__(1024) void kid_seenlog_scan_kernel(const KidLogArgs a)
{
    __shared__ uint32_t part[KID_LOG_WGS];
    uint32_t *c = a.counts + (size_t)blockIdx.x * KID_LOG_WGS;
    const uint32_t v = c[threadIdx.x];
    part[threadIdx.x] = v;
    __syncthreads();
    for (uint32_t o = 1; o < KID_LOG_WGS; o <<= 1) {
        const uint32_t add = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += add;
        __syncthreads();
    }
    c[threadIdx.x] = part[threadIdx.x] - v;
    if (threadIdx.x == KID_LOG_WGS - 1u) a.bin_total[blockIdx.x] = part[threadIdx.x];
}
// where the bins start in `sorted`: exclusive scan of bin_total (<= 1024 bins) into LDS; returns the grand total
__device__ __forceinline__ uint32_t kid_log_bin_bases(const KidLogArgs &a, uint32_t *bases /* nbins + 1 */)
{
    // (one wave does it: the totals are few)
    if (threadIdx.x < 64u) {
        uint32_t run = 0;
        for (uint32_t i0 = 0; i0 < a.nbins; i0 += 64u) {
            const uint32_t i = i0 + threadIdx.x;
            uint32_t v = i < a.nbins ? a.bin_total[i] : 0u, x = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)x, o); if ((int)threadIdx.x >= o) x += y; }
            if (i < a.nbins) bases[i] = run + x - v;
            run += (uint32_t)__shfl((int)x, 63);
        }
        if (threadIdx.x == 0) bases[a.nbins] = run;
    }
    __syncthreads();
    return bases[a.nbins];
}
// exclusive scan of v[0 .. n) (n <= 4 x blockDim) into out[0 .. n), by a workgroup of 256 threads; returns the total
__device__ __forceinline__ uint32_t kid_block_exscan(const uint32_t *v, uint32_t *out, const uint32_t n, uint32_t *wave_tot /* [5] */)
{
    const uint32_t t = threadIdx.x, i0 = 4u * t;
    uint32_t x[4], sum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { x[j] = i0 + j < n ? v[i0 + j] : 0u; sum += x[j]; }
    uint32_t inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t y = (uint32_t)__shfl_up((int)inc, o); if ((int)(t & 63u) >= o) inc += y; }
    if ((t & 63u) == 63u) wave_tot[t >> 6] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < (t >> 6); w++) before += wave_tot[w];
    uint32_t run = before + inc - sum;
#pragma unroll
    for (int j = 0; j < 4; j++) { if (i0 + j < n) out[i0 + j] = run; run += x[j]; }
    const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return total;
}
// The log -> `sorted`, bin by bin.  A workgroup takes its share in tiles of KID_LOG_TILE entries and sorts a tile in LDS
// first (count, scan, place), so that what goes out to memory are runs of neighbouring entries of one bin -- scattering
// the entries one by one took 7 ps each (47 M partial-line writes per pass, profiles/r03/seen_log_first_kernel_stats.csv).
#define KID_LOG_TILE 4096u
__global__ __launch_bounds__(256) void kid_seenlog_scatter_kernel(const KidLogArgs a)
{
    extern __shared__ uint32_t kid_lh[]; // bin bases [nbins + 1], next place of this workgroup per bin [nbins], the tile's counts and offsets [2 nbins], the tile
    const uint32_t nb = a.nbins;
    uint32_t *bases = kid_lh, *next = bases + nb + 1u, *hist = next + nb, *toff = hist + nb, *stage = toff + nb;
    __shared__ uint32_t wave_tot[5];
    kid_log_bin_bases(a, bases);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        // (by the places ASKED for -- hits, header matches that were none, and what no longer fitted: a log that
        // overflows holds fewer entries than there were hits.  Reported as ONE word, a rate: the host runs far ahead of
        // the device, and a count would meet the wrong number of reads there)
        unsigned long long asked = 0;
        for (uint32_t i = 0; i < KID_LOG_SHARDS; i++) asked += a.tail[i * 16u];
        if (a.host_total && a.reads) *a.host_total = ((asked << 10) / a.reads) | (1ull << 63);
        if (a.reads && asked > 8ull * a.reads) {
            for (int i = 0; i < 4; i++)
                if (a.blocks[i]) a.blocks[i]->seen_log = nullptr;
            if (a.off_flag) *a.off_flag = 1u;
        }
    }
    for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) next[i] = bases[i] + a.counts[i * KID_LOG_WGS + blockIdx.x];
    uint32_t b0, b1;
    kid_log_share(a, b0, b1);
    for (uint32_t t0 = b0; t0 < b1; t0 += KID_LOG_TILE) { // (workgroup-uniform)
        const uint32_t n = b1 - t0 < KID_LOG_TILE ? b1 - t0 : KID_LOG_TILE;
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) hist[i] = 0;
        __syncthreads();
        uint32_t e[KID_LOG_TILE / 256u], r[KID_LOG_TILE / 256u];
#pragma unroll
        for (uint32_t j = 0; j < KID_LOG_TILE / 256u; j++) {
            const uint32_t i = j * 256u + threadIdx.x;
            e[j] = i < n ? a.log[t0 + i] : KID_LOG_NONE;
        }
#pragma unroll
        for (uint32_t j = 0; j < KID_LOG_TILE / 256u; j++)
            r[j] = e[j] != KID_LOG_NONE ? atomicAdd(&hist[e[j] >> KID_LOG_BIN_BITS], 1u) : 0u; // its rank among the tile's entries of its bin
        __syncthreads();
        const uint32_t tile_n = kid_block_exscan(hist, toff, nb, wave_tot);
#pragma unroll
        for (uint32_t j = 0; j < KID_LOG_TILE / 256u; j++)
            if (e[j] != KID_LOG_NONE) stage[toff[e[j] >> KID_LOG_BIN_BITS] + r[j]] = e[j];
        __syncthreads();
        for (uint32_t p = threadIdx.x; p < tile_n; p += blockDim.x) { // neighbours in `stage` are neighbours in `sorted`
            const uint32_t v = stage[p], bin = v >> KID_LOG_BIN_BITS;
            a.sorted[next[bin] + (p - toff[bin])] = v;
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) next[i] += hist[i];
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void kid_seenlog_apply_kernel(const KidLogArgs a)
{
    extern __shared__ uint32_t kid_lh[]; // 2^KID_LOG_BIN_BITS bits of the bitmap, then the bin bases
    const uint32_t piece_words = 1u << (KID_LOG_BIN_BITS - 5);
    uint32_t *piece = kid_lh, *bases = kid_lh + piece_words;
    kid_log_bin_bases(a, bases);
    const uint32_t e0 = bases[blockIdx.x], e1 = bases[blockIdx.x + 1u];
    if (e0 == e1) return; // (workgroup-uniform)
    for (uint32_t i = threadIdx.x; i < piece_words; i += blockDim.x) piece[i] = 0;
    __syncthreads();
    auto set = [&](const uint32_t v) { const uint32_t e = v & ((1u << KID_LOG_BIN_BITS) - 1u); atomicOr(&piece[e >> 5], 1u << (e & 31u)); };
    // the bin's entries, 16 bytes per load between the first and the last 16-byte boundary
    const uint32_t up = (e0 + 3u) & ~3u, a0 = up < e1 ? up : e1, a1 = a0 + ((e1 - a0) & ~3u);
    for (uint32_t i = e0 + threadIdx.x; i < a0; i += blockDim.x) set(a.sorted[i]);
    const uint4 *p4 = reinterpret_cast<const uint4 *>(a.sorted + a0);
    for (uint32_t i = threadIdx.x; i < (a1 - a0) >> 2; i += blockDim.x) { const uint4 v = p4[i]; set(v.x); set(v.y); set(v.z); set(v.w); }
    for (uint32_t i = a1 + threadIdx.x; i < e1; i += blockDim.x) set(a.sorted[i]);
    __syncthreads();
    const uint64_t w0 = (uint64_t)blockIdx.x * piece_words;
    for (uint32_t i = threadIdx.x; i < piece_words; i += blockDim.x) {
        const uint32_t v = piece[i];
        if (v && w0 + i < a.seen_words) a.seen[w0 + i] |= v; // the only writer of this piece while the pass runs
    }
}

// ------------------------------------------------------------------ ucount from the seen-bitmap
// ucount[t] = number of distinct DB k-mers of target t seen in the sample
// (newkmer_10nx.cpp:596-603), counted over the entry ordinals [w_begin*32, w_end*32): bit o of the
// bitmap = "the key first inserted as entry o was hit", ord_target[o] = that entry's target
template <bool HIST>
__global__ void kid_ucount_kernel(const uint32_t *seen, uint64_t w_begin, uint64_t w_end, const uint32_t *ord_target,
                                  unsigned long long *ucount, uint32_t ntar)
{
    // the bitmap is almost empty: stream it 16 bytes per lane and only look inside non-zero words;
    // counts go to a per-workgroup LDS histogram first (millions of hits land on a few thousand targets)
    extern __shared__ uint32_t kid_uhist[];
    if (HIST) {
        for (uint32_t i = threadIdx.x; i < ntar; i += blockDim.x) kid_uhist[i] = 0;
        __syncthreads();
    }
    const uint64_t q_begin = w_begin >> 2, q_end = w_end >> 2; // callers pass 128-cell aligned ranges
    const uint4 *seen4 = reinterpret_cast<const uint4 *>(seen);
    for (uint64_t q = q_begin + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < q_end; q += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = seen4[q];
        if ((v.x | v.y | v.z | v.w) == 0) continue;
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t bits = words[i];
            while (bits) {
                const uint32_t bpos = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t t = ord_target[(q * 4ull + (uint64_t)i) * 32ull + bpos];
                if (HIST) atomicAdd(&kid_uhist[t], 1u);
                else atomicAdd(&ucount[t], 1ull);
            }
        }
    }
    if (HIST) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < ntar; i += blockDim.x) {
            const uint32_t c = kid_uhist[i];
            if (c) atomicAdd(&ucount[i], (unsigned long long)c);
        }
    }
}

__global__ void kid_or_kernel(uint32_t *dst, const uint32_t *src, uint64_t nwords)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < nwords; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] |= src[i];
}

// ------------------------------------------------------------------ synthetic data
__global__ void kid_synth_keys_kernel(uint64_t seed, int k, const uint64_t *cum, int32_t ntar, uint64_t j0, uint64_t n,
                                      uint64_t *keys, uint32_t *targets)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        keys[i] = kid_synth_db_key(seed, k, j0 + i);
        targets[i] = kid_synth_target_of(cum, ntar, j0 + i);
    }
}

__global__ void kid_synth_reads_kernel(uint64_t db_seed, uint64_t read_seed, int k, const uint64_t *cum,
                                       const int32_t *parent, int32_t ntar, uint64_t r0, uint64_t n, uint32_t len,
                                       uint8_t *bases)
{
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        kid_synth_read(db_seed, read_seed, k, cum, parent, ntar, r0 + i, len, bases + i * (uint64_t)len);
}

// ------------------------------------------------------------------ random-gather ceiling
// INF independent 16-byte loads per lane per round from uniformly random cells
// The same question asked the way the classify kernel asks it: random 128-byte LINES of the table (cell 0 of a line),
// RUN consecutive lanes on one line (1: 64 distinct lines per load; 8: what the headers of neighbouring k-mers look like),
// four loads in flight per lane, issued from inline assembly and waited for once.
template <int RUN, int MODE = 0> // MODE bit 0: a random cell of the line instead of cell 0; bit 1: the compiler's load and wait instead of inline assembly
__global__ __launch_bounds__(256) void kid_gather_lines_kernel(const uint4 *table, uint32_t line_mask, uint64_t rounds, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint64_t ctr = wave * 0x9E3779B97F4A7C15ULL + 12345;
    for (uint64_t r = 0; r < rounds; r++) {
        kid_u4 a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ctr += 0xD1B54A32D192ED03ULL;
            const uint32_t line = (uint32_t)kid_fmix64(ctr ^ ((uint64_t)(lane / (uint32_t)RUN) << 48)) & line_mask;
            const uint4 *p = table + (uint64_t)line * KID_LINE_CELLS + ((MODE & 1) ? (uint32_t)(ctr >> 40) & 7u : 0u);
            if (MODE & 2) { const uint4 v = *p; a[u] = kid_u4{v.x, v.y, v.z, v.w}; }
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(a[u]) : "v"(p) : "memory");
        }
        if (!(MODE & 2)) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : : "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= a[u].x ^ a[u].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int INF>
__global__ __launch_bounds__(256) void kid_gather_kernel(const uint4 *table, uint32_t slot_mask, uint64_t rounds, uint32_t *sink)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint32_t acc = 0;
    uint64_t ctr = tid * 0x9E3779B97F4A7C15ULL;
    for (uint64_t r = 0; r < rounds; r++) {
        uint4 c[INF];
#pragma unroll
        for (int u = 0; u < INF; u++) {
            ctr += 0xD1B54A32D192ED03ULL;
            c[u] = table[(uint32_t)kid_fmix64(ctr) & slot_mask];
        }
#pragma unroll
        for (int u = 0; u < INF; u++) acc ^= c[u].x ^ c[u].z;
    }
    if (acc == 0x12345678u) sink[0] = acc; // never true in practice; keeps the loads alive
}
