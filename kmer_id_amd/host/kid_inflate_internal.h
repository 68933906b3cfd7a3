// kid_inflate_internal.h -- the Huffman decode tables shared by the sequential reader (kid_inflate.cpp) and the
// pieces of the parallel one (kid_pargz.cpp).  Not an interface: both files are the same component.
#pragma once
#include <stdint.h>
#include <string.h>

namespace kidhost {

static inline uint64_t load64(const uint8_t *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v;
}

// ---------------------------------------------------------------- decode tables
// An entry: bits 0-7 the bits to drop for the symbol: its code word (in a second-level table: the part behind the first
// level's index) plus the extra bits of a length or distance; bits 8-11 the code word's share of that (the extra bits'
// value is what lies above it) or, for a pointer, the second-level table's index width; bits 12-15 what it is; bits
// 16-31 the literal / the base of the length or distance / the second-level table's position.  0 = no such code.
// E_LIT | E_BASE = TWO literals (the second in bits 24-31) whose code words fit the first-level index together: a symbol
// costs a dependent table load, this way a load yields up to two literals.
static const uint32_t E_LIT = 1u << 15, E_SUB = 1u << 14, E_EOB = 1u << 13, E_BASE = 1u << 12;
static const unsigned LROOT = 11, DROOT = 8;
static const unsigned LT_SIZE = (1u << LROOT) + 288 * 16, DT_SIZE = (1u << DROOT) + 32 * 128;

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline unsigned rev16(unsigned x)
{
    x = ((x & 0x5555) << 1) | ((x >> 1) & 0x5555);
    x = ((x & 0x3333) << 2) | ((x >> 2) & 0x3333);
    x = ((x & 0x0f0f) << 4) | ((x >> 4) & 0x0f0f);
    return ((x & 0xff) << 8) | (x >> 8);
}

static inline uint32_t litlen_entry(unsigned sym)
{
    if (sym < 256) return E_LIT | (sym << 16);
    if (sym == 256) return E_EOB;
    if (sym < 286) return E_BASE | ((uint32_t)LEN_BASE[sym - 257] << 16) | LEN_EXTRA[sym - 257];
    return 0;
}
static inline uint32_t dist_entry(unsigned sym)
{
    if (sym < 30) return E_BASE | ((uint32_t)DIST_BASE[sym] << 16) | DIST_EXTRA[sym];
    return 0;
}

// Canonical Huffman code -> table.  0 = a complete code, 1 = over-subscribed, 2 = incomplete (table usable: the missing
// code words are "no such code"), 3 = no code at all.
template <typename EntryOf>
static int build_table(uint32_t *tab, unsigned root, unsigned cap, const uint8_t *lens, unsigned n, EntryOf entry_of, unsigned *max_len_out)
{
    unsigned count[16] = {0};
    for (unsigned s = 0; s < n; s++) count[lens[s]]++;
    count[0] = 0;
    unsigned max_len = 15;
    while (max_len > 0 && count[max_len] == 0) max_len--;
    *max_len_out = max_len;
    memset(tab, 0, sizeof(uint32_t) << root);
    if (max_len == 0) return 3;
    int left = 1;
    for (unsigned l = 1; l <= 15; l++) {
        left = left * 2 - (int)count[l];
        if (left < 0) return 1;
    }
    unsigned next[16], code = 0;
    for (unsigned l = 1; l <= 15; l++) {
        code = (code + count[l - 1]) << 1;
        next[l] = code;
    }
    uint16_t rev_of[288];
    uint8_t longest[1u << LROOT]; // per first-level index: the longest code word that starts with it
    const unsigned mask = (1u << root) - 1;
    if (max_len > root) memset(longest, 0, (size_t)1 << root);
    for (unsigned s = 0; s < n; s++) {
        const unsigned l = lens[s];
        if (!l) continue;
        const unsigned r = rev16(next[l]++) >> (16 - l);
        rev_of[s] = (uint16_t)r;
        if (l > root && longest[r & mask] < l) longest[r & mask] = (uint8_t)l;
    }
    unsigned free_at = 1u << root;
    for (unsigned s = 0; s < n; s++) {
        const unsigned l = lens[s];
        if (!l) continue;
        const unsigned r = rev_of[s];
        const uint32_t e0 = entry_of(s); // (the number of extra bits in its low byte)
        if (l <= root) {
            const uint32_t e = (e0 & ~0xffu) | (l << 8) | (l + (e0 & 0xff));
            for (unsigned i = r; i <= mask; i += 1u << l) tab[i] = e;
            continue;
        }
        const unsigned pre = r & mask, sb = longest[pre] - root;
        if (tab[pre] == 0) {
            if (free_at + (1u << sb) > cap) return 1; // (cannot happen for a code that passed the check above)
            tab[pre] = E_SUB | (free_at << 16) | (sb << 8) | root;
            memset(tab + free_at, 0, sizeof(uint32_t) << sb);
            free_at += 1u << sb;
        }
        uint32_t *sub = tab + (tab[pre] >> 16);
        const uint32_t e = (e0 & ~0xffu) | ((l - root) << 8) | (l - root + (e0 & 0xff));
        for (unsigned i = r >> root; i < (1u << sb); i += 1u << (l - root)) sub[i] = e;
    }
    return left > 0 ? 2 : 0;
}

// Literal pairs: where a first-level entry is a literal whose code word leaves room in the index for another literal's
// whole code word, the entry becomes both.  (ONE-symbol decoding, at the very end of a file, uses the table without.)
static void pair_literals(uint32_t *tab, uint32_t *single)
{
    memcpy(single, tab, sizeof(uint32_t) << LROOT);
    for (unsigned i = 0; i < (1u << LROOT); i++) {
        const uint32_t e1 = single[i];
        if ((e1 & 0xf000) != E_LIT) continue;
        const unsigned l1 = e1 & 0xff;
        const uint32_t e2 = single[i >> l1];
        if ((e2 & 0xf000) != E_LIT) continue;
        const unsigned l2 = e2 & 0xff;
        if (l1 + l2 > LROOT) continue;
        tab[i] = E_LIT | E_BASE | (e1 & 0xff0000) | ((e2 & 0xff0000) << 8) | (l1 + l2);
    }
}

} // namespace kidhost
