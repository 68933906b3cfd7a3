// kid_common.h -- integer helpers shared by host code and gfx950 device code.
// Everything here is exact integer arithmetic; there is no floating point on
// the classification path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define KID_HD __host__ __device__ __forceinline__
#else
#define KID_HD inline
#endif

// Hashtable::integerHash, newkmer_10nx.cpp:189-197 (MurmurHash3 64-bit finaliser)
KID_HD uint64_t kid_fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

// reverse the order of the 32 two-bit groups of x
KID_HD uint64_t kid_rev2(uint64_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // bit-reverse each half, swap the halves, then swap the two bits of every group -- all 32-bit ops
    uint32_t lo = __brev((uint32_t)(x >> 32)), hi = __brev((uint32_t)x);
    lo = ((lo >> 1) & 0x55555555u) | ((lo & 0x55555555u) << 1);
    hi = ((hi >> 1) & 0x55555555u) | ((hi & 0x55555555u) << 1);
    return ((uint64_t)hi << 32) | lo;
#else
    x = ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
    x = ((x >> 2) & 0x3333333333333333ULL) | ((x & 0x3333333333333333ULL) << 2);
    x = ((x >> 4) & 0x0F0F0F0F0F0F0F0FULL) | ((x & 0x0F0F0F0F0F0F0F0FULL) << 4);
    x = ((x >> 8) & 0x00FF00FF00FF00FFULL) | ((x & 0x00FF00FF00FF00FFULL) << 8);
    x = ((x >> 16) & 0x0000FFFF0000FFFFULL) | ((x & 0x0000FFFF0000FFFFULL) << 16);
    x = (x >> 32) | (x << 32);
    // a full bit reversal also swapped the two bits inside every group: undo that
    return ((x >> 1) & 0x5555555555555555ULL) | ((x & 0x5555555555555555ULL) << 1);
#endif
}

// The reference keeps two rolling keys (newkmer_10nx.cpp:480-519): keyF with the
// oldest base in the top group, keyR with the complement of the oldest base in
// the bottom group.  For a full window keyR is a pure function of keyF.
KID_HD uint64_t kid_revcomp(uint64_t keyF, int k)
{
    return (~kid_rev2(keyF)) >> (64 - 2 * k);
}

// key handed to getHash: min(keyF, keyR), newkmer_10nx.cpp:528
KID_HD uint64_t kid_canonical(uint64_t keyF, int k)
{
    uint64_t r = kid_revcomp(keyF, k);
    return keyF < r ? keyF : r;
}

// base -> 2-bit code, -1 for a byte that resets the window (newkmer_10nx.cpp:478-525;
// U/u only under KID_FLAG_U_IS_T, kmer_read_vf6.cpp:496-500,521-525)
KID_HD int kid_base_code(uint8_t c, bool u_is_t)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    case 'U': case 'u': return u_is_t ? 3 : -1;
    default: return -1;
    }
}

// ---------------------------------------------------------------- minimizer-localised table geometry
// The GPU-native table keeps the reference's 16-byte cells and its first-insert-wins lookup
// results but places a key by the MINIMIZER of its k-mer: all k-mers that share their smallest
// (hashed, strand-symmetric) 16-mer go to one 128-byte line = 8 cells.  Consecutive k-mers of a
// read share their minimizer for ~8 positions, so the lanes of a wavefront (consecutive read
// positions) touch ~9 distinct lines per 64 lookups instead of 64 -- the lookups coalesce.
// Inside a line, cell 0 is a header {7 x 16-bit key fingerprints, count}: one 16-byte load
// answers "absent" (99 % of all lookups); a fingerprint match costs one more load.  A full line
// (7 entries) chains into the next one; with 16-mers a minimizer owns < 1 DB key on average at
// bact10 scale, so full lines are a 1e-5 event (with 15-mers ~3 % of the lookups met one).
// Only legal where the probe loop is unbounded (newkmer_10nx, kmer_read_vf6): there the answer
// depends on the key -> first target map alone, not on where cells sit.
// minimizer length m and window w = k - m + 1 (m-mers per k-mer): w <= 17 is what the 16-lane row
// scans of the kernel support (a window may touch two rows, not three); k = 30 -> w = 17, m = 14 (28 bits)
// w = 17 (k >= 26; k = 30 -> m = 14) also fits the two-row scheme and a read would meet 11 % fewer distinct minimizers --
// but a k-mer's minimizer is the SMALLEST of its m-mer hashes, so the minimizers in use crowd into the lowest ~1/w of the
// hash space: at m = 14 that leaves ~13 M lines' worth of minimizers for 108.6 M keys, the lines overflow (1.24 cells
// per lookup instead of 1.01) and the kernel takes twice as long (profiles/r02/ab_min_window.txt).  m stays 16.
#ifndef KID_MIN_W
#define KID_MIN_W 15
#endif
KID_HD int kid_min_window(int k) { return (KID_MIN_W == 17 && k >= 26) ? 17 : (k >= 31 ? 16 : 15); }
KID_HD int kid_min_mlen(int k) { return k - kid_min_window(k) + 1; }
#define KID_LINE_CELLS 8    // header + 7 entries
#define KID_LINE_ENTRIES 7
#define KID_HDR_FULL 8u     // header count value: 7 entries stored and the chain continues in the next line
// Word 3 of a header: [15:0] fingerprint of entry 6, [19:16] count (0..7, 8 = full and chained), [31:20] a
// 12-bit filter of the fingerprints of the keys that were pushed past this line.  A lookup follows the chain
// only if the line is full AND its key's filter bit is set: lines fill up when a run of consecutive genome
// k-mers shares one minimizer (up to 15 keys on 7 entries), and without the filter every lookup that lands on
// a full line -- present or not -- would have to walk on.

KID_HD uint32_t kid_rev2_32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    x = __brev(x);
#else
    x = ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = ((x >> 8) & 0x00FF00FFu) | ((x & 0x00FF00FFu) << 8);
    x = (x >> 16) | (x << 16);
#endif
    return ((x >> 1) & 0x55555555u) | ((x & 0x55555555u) << 1);
}

// hash of the canonical form of the m-mer f (2m bits, m <= 16); identical for an m-mer and its
// reverse complement.  One multiply + xorshift: enough to break up lexicographic order.
KID_HD uint32_t kid_mmer_hash(uint32_t f, int m)
{
    const uint32_t r = (~kid_rev2_32(f)) >> (32 - 2 * m);
    uint32_t h = f < r ? f : r;
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    return h;
}

// the same with the reverse complement r of f already at hand
KID_HD uint32_t kid_mmer_hash2(uint32_t f, uint32_t r)
{
    uint32_t h = f < r ? f : r;
    h *= 0x9E3779B1u;
    h ^= h >> 15;
    return h;
}

// minimizer of a whole k-mer given as its 2k-bit forward key (brute force: table build, unit lookups)
KID_HD uint32_t kid_minimizer_of_key(uint64_t keyF, int k)
{
    const int w = kid_min_window(k), m = kid_min_mlen(k);
    const uint32_t mm = m >= 16 ? 0xFFFFFFFFu : ((1u << (2 * m)) - 1u);
    uint32_t g = 0xFFFFFFFFu;
    for (int j = 0; j < w; j++) {
        const uint32_t h = kid_mmer_hash((uint32_t)(keyF >> (2 * (w - 1 - j))) & mm, m);
        g = h < g ? h : g;
    }
    return g;
}

// line of a minimizer: top bits of one more product (line_shift = 32 - log2(number of lines))
KID_HD uint32_t kid_minloc_line(uint32_t g, uint32_t line_shift)
{
    return (g * 0xC2B2AE3Du) >> line_shift;
}

KID_HD uint32_t kid_hdr_count(uint32_t w3) { return (w3 >> 16) & 15u; }
KID_HD uint32_t kid_ovf_bit(uint32_t fp) { return 20u + ((fp * 12u) >> 16); } // fp < 2^16 -> bits 20..31
KID_HD bool kid_hdr_continues(uint32_t w3, uint32_t fp)
{
    return kid_hdr_count(w3) >= KID_HDR_FULL && ((w3 >> kid_ovf_bit(fp)) & 1u) != 0;
}

// 16-bit fingerprint of a key kept in the line header; never 0 (0 = unused header slot)
KID_HD uint32_t kid_key_fp(uint64_t key)
{
    const uint32_t hi = (uint32_t)(key >> 32);
    const uint32_t f = (((uint32_t)key ^ ((hi << 7) | (hi >> 25))) * 0x9E3779B1u) >> 16;
    return f ? f : 1u;
}

// ---------------------------------------------------------------- synthetic data
KID_HD uint64_t kid_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// j-th key of the synthetic DB: a uniformly random k-mer, made canonical like
// the builder's output (kmer_build_vf6.cpp:604,622)
KID_HD uint64_t kid_synth_db_key(uint64_t seed, int k, uint64_t j)
{
    uint64_t raw = kid_splitmix64(seed + j) & ((1ULL << (2 * k)) - 1);
    return kid_canonical(raw, k);
}

// target owning key ordinal j: cum[t] <= j < cum[t+1]
KID_HD uint32_t kid_synth_target_of(const uint64_t *cum, int32_t ntar, uint64_t j)
{
    int32_t lo = 0, hi = ntar; // invariant cum[lo] <= j < cum[hi]
    while (hi - lo > 1) {
        int32_t mid = (lo + hi) >> 1;
        if (cum[mid] <= j) lo = mid; else hi = mid;
    }
    return (uint32_t)lo;
}

KID_HD void kid_synth_put_kmer(uint8_t *dst, uint64_t keyF, int k, bool rc, bool lower)
{
    uint64_t v = rc ? kid_revcomp(keyF, k) : keyF;
    const char *al = lower ? "acgt" : "ACGT";
    for (int i = 0; i < k; i++) dst[i] = (uint8_t)al[(v >> (2 * (k - 1 - i))) & 3];
}

// One synthetic read (DESIGN.md "synthetic workload"): random bases; half of the
// reads carry 1-4 DB k-mers of one lineage at non-overlapping slots, ~2 % of
// those one more from anywhere in the DB, ~1 % of all reads one 'N', ~1 % are
// lower case.
KID_HD void kid_synth_read(uint64_t db_seed, uint64_t read_seed, int k, const uint64_t *cum,
                           const int32_t *parent, int32_t ntar, uint64_t r, uint32_t len, uint8_t *out)
{
    const uint64_t s = kid_splitmix64(read_seed ^ (r * 0xD1B54A32D192ED03ULL));
    const uint64_t d = kid_splitmix64(s + 1000);
    const bool lower = ((d >> 48) & 0xFF) < 3;
    const char *al = lower ? "acgt" : "ACGT";
    for (uint32_t i = 0; i < len; i += 32) {
        uint64_t w = kid_splitmix64(s + (i >> 5));
        uint32_t m = len - i < 32 ? len - i : 32;
        for (uint32_t b = 0; b < m; b++) out[i + b] = (uint8_t)al[(w >> (2 * b)) & 3];
    }
    const uint64_t n_keys = cum[ntar];
    const uint32_t nslots = len / (uint32_t)k;
    if (((d & 0xFF) < 128) && n_keys > 0 && nslots > 0) {
        const uint64_t j0 = kid_splitmix64(s + 1001) % n_keys;
        const uint32_t t0 = kid_synth_target_of(cum, ntar, j0);
        uint32_t n_imp = 1 + (uint32_t)((d >> 8) & 3);
        if (n_imp > nslots) n_imp = nslots;
        const uint32_t slot0 = (uint32_t)((d >> 16) & 0xFF) % nslots;
        for (uint32_t i = 0; i < n_imp; i++) {
            uint64_t j = j0;
            if (i > 0) {
                uint32_t t = t0;
                uint32_t up = (uint32_t)((kid_splitmix64(s + 1002 + i) >> 8) & 3);
                for (uint32_t u = 0; u < up; u++) {
                    uint32_t p = (t != 1 && t > 0) ? (uint32_t)parent[t] : 1u;
                    if (p == 1 || p >= (uint32_t)ntar) break;
                    t = p;
                }
                uint64_t cnt = cum[t + 1] - cum[t];
                if (cnt == 0) { t = t0; cnt = cum[t + 1] - cum[t]; }
                j = cum[t] + kid_splitmix64(s + 1010 + i) % cnt;
            }
            const bool rc = kid_splitmix64(s + 1020 + i) & 1;
            const uint32_t slot = (slot0 + i) % nslots;
            kid_synth_put_kmer(out + slot * (uint32_t)k, kid_synth_db_key(db_seed, k, j), k, rc, lower);
        }
        if ((((d >> 24) & 0xFF) < 5) && n_imp < nslots) {
            const uint64_t j = kid_splitmix64(s + 1030) % n_keys;
            const uint32_t slot = (slot0 + n_imp) % nslots;
            kid_synth_put_kmer(out + slot * (uint32_t)k, kid_synth_db_key(db_seed, k, j), k,
                               (kid_splitmix64(s + 1031) & 1) != 0, lower);
        }
    }
    if (((d >> 32) & 0xFF) < 3) out[(uint32_t)((d >> 40) & 0xFFFF) % len] = 'N';
}
