// kid_pargz.cpp -- see kid_pargz.h
#include "kid_pargz.h"

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "kid_inflate.h"
#include "kid_inflate_internal.h"

namespace kidhost {
namespace {

const uint64_t NOWHERE = ~(uint64_t)0;
const size_t WIN = GzStream::kWindow;
const size_t TAIL_MARGIN = 1100; // a dynamic block header is at most 562 bytes: nothing is parsed this close to the end of the file

struct MemberEnd {
    size_t out_pos; // the member's text ends in front of this symbol of the piece
    uint32_t crc, isize;
};

// One piece of the file, inflated from a block header found from the outside into 16-bit symbols.
struct Chunk {
    uint64_t from_byte = 0, stop_byte = 0; // the header is searched from here; the piece behind searches from stop_byte
    uint64_t start_bit = NOWHERE, end_bit = 0;
    enum Status { REACHED, STREAM_END, GAVE_UP } status = GAVE_UP;
    uint16_t *sym = nullptr;
    size_t n = 0, cap = 0;
    unsigned min_marker = 0xffff; // the farthest reach back into the unknown window (256 + index), 0xffff = none
    std::vector<MemberEnd> ends;
    bool done = false;            // (under the reader's mutex)
    uint8_t table[256 + WIN];     // symbol -> byte, once the window is known
    ~Chunk() { free(sym); }
};

struct Bits {
    const uint8_t *ip;
    uint64_t b = 0;
    unsigned c = 0;
    inline void refill()
    {
        b |= load64(ip) << c;
        ip += (63 - c) >> 3;
        c |= 56;
    }
    inline unsigned take(unsigned n)
    {
        const unsigned v = (unsigned)(b & (((uint64_t)1 << n) - 1));
        b >>= n;
        c -= n;
        return v;
    }
    void to_byte_boundary()
    {
        take(c & 7);
        ip -= c >> 3;
        b = 0;
        c = 0;
    }
};

inline Bits cursor_at(const uint8_t *file, uint64_t bit)
{
    Bits z;
    z.ip = file + bit / 8;
    z.refill();
    z.take((unsigned)(bit % 8));
    return z;
}
inline uint64_t bit_position(const Bits &z, const uint8_t *file) { return (uint64_t)(z.ip - file) * 8 - z.c; }

// The lengths of a dynamic block's two codes (behind the three header bits).  false = not a header the sequential
// reader would take (kid_inflate.cpp block_header(): the same checks in the same order).
bool read_code_lengths(Bits &z, uint8_t *lens /*[288 + 32]*/, unsigned &nlit, unsigned &ndist)
{
    z.refill();
    nlit = z.take(5) + 257;
    ndist = z.take(5) + 1;
    const unsigned ncl = z.take(4) + 4;
    if (nlit > 286 || ndist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19] = {0};
    unsigned kraft = 0;
    for (unsigned i = 0; i < ncl; i++) {
        if ((i & 7) == 0) z.refill();
        const unsigned l = z.take(3);
        cl[order[i]] = (uint8_t)l;
        if (l) kraft += 128u >> l;
    }
    if (kraft != 128) return false; // (the code-length code has to be complete)
    uint32_t ct[128];
    unsigned mx;
    if (build_table(ct, 7, 128, cl, 19, [](unsigned s) { return E_LIT | (s << 16); }, &mx) != 0) return false;
    unsigned i = 0;
    while (i < nlit + ndist) {
        z.refill();
        const uint32_t e = ct[z.b & 127];
        z.take(e & 0xff);
        const unsigned sym = (e >> 16) & 0xff;
        if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
        unsigned rep, val = 0;
        if (sym == 16) {
            if (i == 0) return false;
            val = lens[i - 1];
            rep = 3 + z.take(2);
        } else if (sym == 17) rep = 3 + z.take(3);
        else rep = 11 + z.take(7);
        if (i + rep > nlit + ndist) return false;
        while (rep--) lens[i++] = (uint8_t)val;
    }
    if (lens[256] == 0) return false;
    memmove(lens + 288, lens + nlit, ndist);
    return true;
}

// complete, or the one-code-of-length-1 case zlib lets pass (0 = no code at all: fine for distances only)
bool code_is_usable(const uint8_t *lens, unsigned n, bool may_be_empty)
{
    unsigned count[16] = {0}, used = 0, mx = 0;
    for (unsigned s = 0; s < n; s++)
        if (lens[s]) { count[lens[s]]++; used++; if (lens[s] > mx) mx = lens[s]; }
    if (used == 0) return may_be_empty;
    int left = 1;
    for (unsigned l = 1; l <= 15; l++) {
        left = left * 2 - (int)count[l];
        if (left < 0) return false;
    }
    return left == 0 || mx == 1;
}

// The first bit position in [from_byte * 8, limit_byte * 8) where a dynamic block header that parses starts.
uint64_t find_block(const uint8_t *file, size_t flen, uint64_t from_byte, uint64_t limit_byte, const std::atomic<bool> &cancel)
{
    if (flen < TAIL_MARGIN + 16) return NOWHERE;
    if (limit_byte > flen - TAIL_MARGIN) limit_byte = flen - TAIL_MARGIN;
    uint8_t lens[288 + 32];
    for (uint64_t byte = from_byte; byte < limit_byte; byte++) {
        if ((byte & 4095u) == 0 && cancel.load(std::memory_order_relaxed)) return NOWHERE; // (nobody is going to ask)
        const uint64_t w = load64(file + byte);
        for (unsigned k = 0; k < 8; k++) {
            const uint64_t v = w >> k;
            if (((v >> 1) & 3) != 2) continue;                      // BTYPE
            if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue; // HLIT, HDIST
            Bits z = cursor_at(file, byte * 8 + k);
            z.take(3);
            unsigned nlit, ndist;
            if (!read_code_lengths(z, lens, nlit, ndist)) continue;
            if (!code_is_usable(lens, nlit, false) || !code_is_usable(lens + 288, ndist, true)) continue;
            return byte * 8 + k;
        }
    }
    return NOWHERE;
}

// A gzip member header at p (RFC 1952).  0 = not one that the sequential reader would take, else its length.
size_t member_header_length(const uint8_t *p, const uint8_t *end)
{
    if (end - p < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 0;
    const int flg = p[3];
    const uint8_t *q = p + 10;
    if (flg & 4) {
        if (end - q < 2) return 0;
        const size_t n = q[0] | (q[1] << 8);
        q += 2;
        if ((size_t)(end - q) < n) return 0;
        q += n;
    }
    for (int bit = 8; bit <= 16; bit <<= 1)
        if (flg & bit) {
            while (q < end && *q) q++;
            if (q == end) return 0;
            q++;
        }
    if (flg & 2) {
        if (end - q < 2) return 0;
        if ((uint32_t)(q[0] | (q[1] << 8)) != (crc32_fast(0, p, (size_t)(q - p)) & 0xffff)) return 0;
        q += 2;
    }
    return (size_t)(q - p);
}

bool grow(Chunk &c, size_t want)
{
    size_t cap = c.cap ? c.cap : ((size_t)1 << 20);
    while (cap < want) cap *= 2;
    uint16_t *p = (uint16_t *)realloc(c.sym, cap * sizeof(uint16_t));
    if (!p) return false;
    c.sym = p;
    c.cap = cap;
    return true;
}

// One Huffman block into symbols.  1 = end of block, -1 = something the sequential reader has to look at.
int spec_block(Bits &zz, const uint32_t *L, const uint32_t *D, Chunk &c, size_t &n_io, long lowest, const uint8_t *in_lim, size_t max_out)
{
    uint64_t b = zz.b;
    unsigned cbits = zz.c;
    const uint8_t *ip = zz.ip;
    size_t n = n_io;
    uint16_t *sym = c.sym;
    const unsigned LMASK = (1u << LROOT) - 1, DMASK = (1u << DROOT) - 1;
    int rc = -1;
#define KID_REFILL() do { b |= load64(ip) << cbits; ip += (63 - cbits) >> 3; cbits |= 56; } while (0)
#define KID_DROP(e) do { b >>= ((e) & 0xff); cbits -= ((e) & 0xff); } while (0)
#define KID_PUT(e) do { sym[n] = (uint16_t)(((e) >> 16) & 0xff); sym[n + 1] = (uint16_t)((e) >> 24); n += 1 + (((e) >> 12) & 1); } while (0)
#define KID_VALUE(e, from) (((e) >> 16) + (unsigned)(((from) & (((uint64_t)1 << ((e) & 0xff)) - 1)) >> (((e) >> 8) & 15)))
    KID_REFILL();
    uint32_t e = L[b & LMASK];
    for (;;) {
        if (ip >= in_lim) break;
        if (n + 320 > c.cap) {
            if (n > max_out || !grow(c, n + 320)) break;
            sym = c.sym;
        }
        if (e & E_LIT) {
            KID_DROP(e);
            KID_PUT(e);
            e = L[b & LMASK];
            if (e & E_LIT) {
                KID_DROP(e);
                KID_PUT(e);
                e = L[b & LMASK];
                if (e & E_LIT) {
                    KID_DROP(e);
                    KID_PUT(e);
                    KID_REFILL();
                    e = L[b & LMASK];
                    continue;
                }
            }
            KID_REFILL();
        }
        if (e & E_SUB) {
            KID_DROP(e);
            e = L[(e >> 16) + (b & ((1u << ((e >> 8) & 15)) - 1))];
            if (e & E_LIT) {
                KID_DROP(e);
                sym[n++] = (uint16_t)((e >> 16) & 0xff);
                KID_REFILL();
                e = L[b & LMASK];
                continue;
            }
        }
        if (!(e & E_BASE)) {
            if (e & E_EOB) { KID_DROP(e); rc = 1; }
            break;
        }
        const unsigned len = KID_VALUE(e, b);
        KID_DROP(e);
        e = D[b & DMASK];
        if (e & E_SUB) {
            KID_DROP(e);
            e = D[(e >> 16) + (b & ((1u << ((e >> 8) & 15)) - 1))];
        }
        if (!(e & E_BASE)) break;
        const unsigned dist = KID_VALUE(e, b);
        KID_DROP(e);
        const long src = (long)n - (long)dist;
        if (src < lowest) break;
        KID_REFILL();
        e = L[b & LMASK];
        uint16_t *o = sym + n;
        if (src >= 0) {
            const uint16_t *s = sym + src;
            if (dist >= 4) { // words of four symbols; a word's source lies at least a word behind it
                memcpy(o, s, 8);
                memcpy(o + 4, s + 4, 8);
                for (unsigned k = 8; k < len; k += 4) memcpy(o + k, s + k, 8);
            } else {
                for (unsigned k = 0; k < len; k++) o[k] = s[k];
            }
        } else { // starts in the text in front of this piece: symbols that stand for bytes of the unknown window
            const unsigned first = (unsigned)(256 + (long)WIN + src);
            if (first < c.min_marker) c.min_marker = first;
            unsigned k = 0;
            for (; k < len && src + (long)k < 0; k++) o[k] = (uint16_t)(first + k);
            for (; k < len; k++) o[k] = sym[src + (long)k];
        }
        n += len;
    }
#undef KID_REFILL
#undef KID_DROP
#undef KID_PUT
#undef KID_VALUE
    zz.b = b;
    zz.c = cbits;
    zz.ip = ip;
    n_io = n;
    return rc;
}

// Inflate from c.start_bit until a block boundary at or behind c.stop_byte where a dynamic block begins (the spot the
// piece behind is looking for), the end of the stream, or anything unusual -- then up to the last block boundary.
void spec_decode(const uint8_t *file, size_t flen, Chunk &c, size_t max_out)
{
    std::vector<uint32_t> tables(LT_SIZE + DT_SIZE + (1u << LROOT));
    uint32_t *const lt = tables.data(), *const dt = lt + LT_SIZE, *const scratch = dt + DT_SIZE;
    const uint8_t *const fend = file + flen, *const in_lim = fend - 64, *const parse_lim = fend - TAIL_MARGIN;
    Bits z = cursor_at(file, c.start_bit);
    long lowest = -(long)WIN;
    size_t n = 0;
    c.status = Chunk::GAVE_UP;
    c.end_bit = c.start_bit;
    if (!grow(c, (size_t)1 << 20)) return;
    for (;;) { // in front of a block header
        c.end_bit = bit_position(z, file);
        c.n = n;
        const size_t good_ends = c.ends.size();
        auto give_up = [&]() { c.ends.resize(good_ends); c.status = Chunk::GAVE_UP; };
        if (z.ip >= parse_lim) { give_up(); return; }
        z.refill();
        const unsigned type = (unsigned)(z.b >> 1) & 3;
        if (c.end_bit >= c.stop_byte * 8 && type == 2 && c.end_bit != c.start_bit) { c.status = Chunk::REACHED; return; }
        const bool final_block = z.take(1) != 0;
        z.take(2);
        if (type == 0) {
            z.to_byte_boundary();
            const unsigned len = z.ip[0] | (z.ip[1] << 8), nlen = z.ip[2] | (z.ip[3] << 8);
            if ((len ^ 0xffff) != nlen || z.ip + 4 + len >= in_lim) { give_up(); return; }
            if (n + len + 320 > c.cap && (n + len > max_out || !grow(c, n + len + 320))) { give_up(); return; }
            z.ip += 4;
            for (unsigned k = 0; k < len; k++) c.sym[n + k] = z.ip[k];
            n += len;
            z.ip += len;
        } else if (type == 3) {
            give_up();
            return;
        } else {
            uint8_t lens[288 + 32];
            unsigned nlit, ndist, mx;
            if (type == 1) {
                for (unsigned s = 0; s < 288; s++) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
                for (unsigned s = 0; s < 32; s++) lens[288 + s] = 5;
                nlit = 288;
                ndist = 32;
            } else if (!read_code_lengths(z, lens, nlit, ndist)) {
                give_up();
                return;
            }
            int rc = build_table(lt, LROOT, LT_SIZE, lens, nlit, litlen_entry, &mx);
            if (rc == 1 || rc == 3 || (rc == 2 && mx != 1)) { give_up(); return; }
            rc = build_table(dt, DROOT, DT_SIZE, lens + 288, ndist, dist_entry, &mx);
            if (rc == 1 || (rc == 2 && mx != 1)) { give_up(); return; }
            pair_literals(lt, scratch);
            if (spec_block(z, lt, dt, c, n, lowest, in_lim, max_out) != 1) { give_up(); return; }
        }
        if (!final_block) continue;
        // the end of a member: its sums, then another member or the end of the data
        z.to_byte_boundary();
        if (z.ip + 8 > fend) { give_up(); return; }
        MemberEnd me;
        me.out_pos = n;
        me.crc = (uint32_t)z.ip[0] | ((uint32_t)z.ip[1] << 8) | ((uint32_t)z.ip[2] << 16) | ((uint32_t)z.ip[3] << 24);
        me.isize = (uint32_t)z.ip[4] | ((uint32_t)z.ip[5] << 8) | ((uint32_t)z.ip[6] << 16) | ((uint32_t)z.ip[7] << 24);
        z.ip += 8;
        if (fend - z.ip < 2 || z.ip[0] != 0x1f || z.ip[1] != 0x8b) { // (what follows a member and is no header is ignored)
            c.ends.push_back(me);
            c.n = n;
            c.status = Chunk::STREAM_END;
            return;
        }
        const size_t hl = z.ip < parse_lim ? member_header_length(z.ip, parse_lim) : 0;
        if (hl == 0) { give_up(); return; } // (a damaged or cut-off header: the sequential reader says what it is)
        c.ends.push_back(me);
        z.ip += hl;
        lowest = (long)n;
    }
}

} // namespace

// ---------------------------------------------------------------- the reader
struct ParallelGz::Impl {
    std::string path;
    int threads;
    size_t chunk_bytes, piece_bytes, head;
    int fd = -1;
    const uint8_t *file = nullptr;
    size_t flen = 0;

    // workers
    std::vector<std::thread> pool;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    std::deque<std::function<void()>> urgent, normal;
    bool quit = false;
    std::atomic<bool> cancel{false}; // the pieces still being looked for are not wanted any more

    // pieces of the file
    std::vector<std::shared_ptr<Chunk>> chunks; // [0] unused: the file starts with the sequential reader
    size_t submitted = 1, expect = 1;            // chunks [1, submitted) are with the workers; `expect` should start at `pos_bit`
    size_t ahead = 8;
    unsigned unusable_in_a_row = 0, patience = 4; // pieces that did not continue the text in front of them (see accept_next)

    // the stream
    std::unique_ptr<GzStream> seq;
    bool seq_pending = false;   // go on sequentially at pos_bit as soon as everything queued has been handed out
    uint64_t seq_stop_byte = NOWHERE;
    bool finished = false;
    uint64_t pos_bit = 0;
    uint32_t crc = 0;           // of the current member's text HANDED OUT so far
    uint64_t member_out = 0;
    uint64_t accepted_member_out = 0; // the same as of pos_bit (text accepted, not yet all handed out)
    uint8_t window[WIN];        // the last text in front of pos_bit (right-aligned)
    uint64_t parallel_bytes = 0;

    // A stretch of a piece's text on its way to the caller.  It may run over the ends of gzip members (a bgzip-style file
    // has one every 64 KiB): the sums are kept per segment.
    struct Segment {
        size_t len = 0;          // text up to the member end (or the end of the item)
        uint32_t crc = 0;        // ... its CRC-32 (the worker's part)
        bool member_ends = false;
        uint32_t want_crc = 0, want_isize = 0;
    };
    struct Item {
        HostBuf buf;
        size_t len = 0;
        bool ready = false;
        std::vector<Segment> segs;
        std::shared_ptr<Chunk> chunk;
        size_t a = 0;
    };
    bool has_pending_error = false;
    Fatal pending_error{0, ""};
    std::deque<std::shared_ptr<Item>> items;

    void worker()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_work.wait(lk, [&] { return quit || !urgent.empty() || !normal.empty(); });
                if (quit) return;
                if (!urgent.empty()) { job = std::move(urgent.front()); urgent.pop_front(); }
                else { job = std::move(normal.front()); normal.pop_front(); }
            }
            job();
        }
    }

    void submit_chunks()
    {
        while (submitted < chunks.size() && submitted < expect + ahead) {
            std::shared_ptr<Chunk> c(new Chunk());
            c->from_byte = submitted * chunk_bytes;
            c->stop_byte = submitted + 1 < chunks.size() ? (submitted + 1) * chunk_bytes : NOWHERE / 8;
            chunks[submitted++] = c;
            std::lock_guard<std::mutex> lk(m);
            normal.push_back([this, c] {
                c->start_bit = find_block(file, flen, c->from_byte, c->from_byte + 2 * chunk_bytes, cancel);
                if (c->start_bit != NOWHERE && !cancel.load()) spec_decode(file, flen, *c, chunk_bytes * 16);
                std::lock_guard<std::mutex> lk2(m);
                c->done = true;
                cv_done.notify_all();
            });
            cv_work.notify_one();
        }
    }

    void start_sequential(uint64_t stop_byte)
    {
        GzResume r;
        r.bit_offset = pos_bit;
        r.window_len = member_out < WIN ? (size_t)member_out : WIN;
        r.window = window + (WIN - r.window_len);
        r.crc = crc;
        r.member_out = member_out;
        seq.reset(new GzStream(path, r));
        if (stop_byte != NOWHERE) seq->stop_at_dynamic_block_from(stop_byte);
        seq_pending = false;
    }

    // Take the piece that should start at pos_bit.  (Only with nothing else to do: its worker may still be at it.)
    void accept_next()
    {
        std::shared_ptr<Chunk> c = chunks[expect];
        {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&] { return c->done; });
        }
        const uint64_t next_stop = expect + 1 < chunks.size() ? (uint64_t)(expect + 1) * chunk_bytes : NOWHERE;
        chunks[expect].reset();
        expect++;
        submit_chunks();
        // a reach back in front of the member's first byte is damage: the sequential reader names it
        const uint64_t known = accepted_member_out < WIN ? accepted_member_out : WIN;
        const bool reach_ok = c->min_marker == 0xffff || (uint64_t)(c->min_marker - 256) >= WIN - known;
        if (c->start_bit != pos_bit || !reach_ok) { // not the continuation of what is in front: inflate this stretch in order
            seq_pending = true;
            seq_stop_byte = next_stop;
            // A file that keeps doing this is not made of dynamic-code blocks (stored blocks: data that does not compress):
            // looking for headers in it costs far more than reading it in order does.  The rest goes to the sequential reader.
            if (++unusable_in_a_row >= patience) {
                seq_stop_byte = NOWHERE;
                expect = submitted = chunks.size();
                cancel = true;
            }
            return;
        }
        unusable_in_a_row = 0;
        for (unsigned v = 0; v < 256; v++) c->table[v] = (uint8_t)v;
        memcpy(c->table + 256, window, WIN);
        // the text behind this piece's last 32 KiB: the window of the next
        if (c->n >= WIN) {
            for (size_t i = 0; i < WIN; i++) window[i] = c->table[c->sym[c->n - WIN + i]];
        } else {
            memmove(window, window + c->n, WIN - c->n);
            for (size_t i = 0; i < c->n; i++) window[WIN - c->n + i] = c->table[c->sym[i]];
        }
        // the text in stretches of up to piece_bytes, the member ends inside a stretch noted with it
        size_t at = 0, e = 0;
        while (at < c->n || e < c->ends.size()) {
            std::shared_ptr<Item> it(new Item());
            it->chunk = c;
            it->a = at;
            while (it->len < piece_bytes && (at < c->n || e < c->ends.size())) {
                const size_t upto = e < c->ends.size() ? c->ends[e].out_pos : c->n;
                Segment sg;
                sg.len = upto - at < piece_bytes - it->len ? upto - at : piece_bytes - it->len;
                at += sg.len;
                it->len += sg.len;
                accepted_member_out += sg.len;
                if (e < c->ends.size() && c->ends[e].out_pos == at) {
                    sg.member_ends = true;
                    sg.want_crc = c->ends[e].crc;
                    sg.want_isize = c->ends[e].isize;
                    accepted_member_out = 0;
                    e++;
                }
                it->segs.push_back(sg);
            }
            parallel_bytes += it->len;
            if (it->len) it->buf.resize(head + piece_bytes);
            items.push_back(it);
            {
                std::lock_guard<std::mutex> lk(m);
                urgent.push_back([this, it] {
                    uint8_t *dst = it->len ? (uint8_t *)it->buf.data() + head : nullptr;
                    const uint16_t *s = it->chunk->sym + it->a;
                    const uint8_t *t = it->chunk->table;
                    for (size_t i = 0; i < it->len; i++) dst[i] = t[s[i]];
                    size_t o = 0;
                    for (Segment &sg : it->segs) {
                        sg.crc = crc32_fast(0, dst + o, sg.len);
                        o += sg.len;
                    }
                    it->chunk.reset();
                    std::lock_guard<std::mutex> lk2(m);
                    it->ready = true;
                    cv_done.notify_all();
                });
                cv_work.notify_one();
            }
        }
        pos_bit = c->end_bit;
        if (c->status == Chunk::STREAM_END) { finished = true; return; }
        if (c->status == Chunk::GAVE_UP) {
            seq_pending = true;
            seq_stop_byte = next_stop;
        }
    }
};

ParallelGz::ParallelGz(const std::string &path, int threads, size_t chunk_bytes, size_t piece_bytes, size_t head, unsigned patience) : impl_(new Impl())
{
    Impl &z = *impl_;
    z.patience = patience < 1 ? 1 : patience;
    z.path = path;
    z.threads = threads;
    z.chunk_bytes = chunk_bytes < 4096 ? 4096 : chunk_bytes;
    z.piece_bytes = piece_bytes < GzStream::kMinRead ? GzStream::kMinRead : piece_bytes;
    z.head = head < WIN ? WIN : head;
    z.seq.reset(new GzStream(path)); // (throws when the file cannot be opened)
    struct stat st;
    z.fd = ::open(path.c_str(), O_RDONLY);
    if (z.fd >= 0 && fstat(z.fd, &st) == 0 && S_ISREG(st.st_mode)) z.flen = (size_t)st.st_size;
    const size_t n_chunks = z.flen / z.chunk_bytes;
    if (threads >= 2 && n_chunks >= 2) {
        void *p = mmap(nullptr, z.flen, PROT_READ, MAP_PRIVATE, z.fd, 0);
        if (p != MAP_FAILED) z.file = (const uint8_t *)p;
    }
    if (!z.file) return; // small, or no second thread: the sequential reader does the whole file
    z.chunks.resize(n_chunks); // (made when they are handed to the workers: `ahead` of them exist at a time)
    z.ahead = (size_t)threads * 2;
    z.seq->stop_at_dynamic_block_from(z.chunk_bytes);
    for (int t = 0; t < threads; t++) z.pool.emplace_back([this] { impl_->worker(); });
    z.submit_chunks();
}

ParallelGz::~ParallelGz()
{
    Impl &z = *impl_;
    z.cancel = true;
    {
        std::lock_guard<std::mutex> lk(z.m);
        z.quit = true;
        z.cv_work.notify_all();
    }
    for (std::thread &t : z.pool) t.join();
    z.items.clear();
    z.chunks.clear();
    if (z.file) munmap((void *)z.file, z.flen);
    if (z.fd >= 0) ::close(z.fd);
}

uint64_t ParallelGz::bytes_in_parallel() const { return impl_->parallel_bytes; }

void ParallelGz::close()
{
    if (impl_->seq) impl_->seq->close();
}

bool ParallelGz::next(HostBuf &buf, size_t &len)
{
    Impl &z = *impl_;
    for (;;) {
        // keep the queue fed: pieces whose workers are through are taken in while earlier text is still being handed out
        while (!z.seq && !z.seq_pending && !z.finished && z.expect < z.submitted && z.items.size() < z.ahead * 2) {
            {
                std::lock_guard<std::mutex> lk(z.m);
                if (!z.chunks[z.expect]->done) break;
            }
            z.accept_next();
        }
        if (z.has_pending_error) throw z.pending_error;
        if (!z.items.empty()) { // hand out what is queued, in order
            std::shared_ptr<Impl::Item> it = z.items.front();
            {
                std::unique_lock<std::mutex> lk(z.m);
                z.cv_done.wait(lk, [&] { return it->ready; });
            }
            z.items.pop_front();
            size_t good = 0; // text in front of a member whose sums are wrong is handed out first, like the sequential reader does
            for (const Impl::Segment &sg : it->segs) {
                z.crc = (uint32_t)crc32_combine(z.crc, sg.crc, (z_off_t)sg.len);
                z.member_out += sg.len;
                good += sg.len;
                if (!sg.member_ends) continue;
                const char *what = sg.want_crc != z.crc ? "incorrect data check" : sg.want_isize != (uint32_t)z.member_out ? "incorrect length check" : nullptr;
                if (what) {
                    z.pending_error = Fatal{255, z.path + ": " + what};
                    z.has_pending_error = true;
                    z.items.clear();
                    break;
                }
                z.crc = 0;
                z.member_out = 0;
            }
            if (good == 0) continue; // (member ends only, or nothing in front of the damage)
            buf = std::move(it->buf);
            len = good;
            return true;
        }
        if (z.finished) return false;
        if (z.seq_pending) z.start_sequential(z.seq_stop_byte);
        if (z.seq) {
            if (buf.size() < z.head + z.piece_bytes) buf.resize(z.head + z.piece_bytes);
            const size_t got = z.seq->read((uint8_t *)buf.data() + z.head, z.piece_bytes);
            if (z.seq->stopped() && z.file) { // from here the pieces inflated side by side take over
                z.pos_bit = z.seq->stopped_at_bit();
                z.crc = z.seq->member_crc();
                z.member_out = z.accepted_member_out = z.seq->member_length();
                uint8_t h[WIN];
                const size_t hn = z.seq->history(h);
                memcpy(z.window + (WIN - hn), h, hn);
                z.seq.reset();
            } else if (got == 0) {
                z.finished = true; // (the reader is kept: close() asks it whether the file was cut off)
                return false;
            }
            if (got) { len = got; return true; }
            continue;
        }
        // between pieces: the next one
        if (z.expect < z.submitted) z.accept_next();
        else { z.seq_pending = true; z.seq_stop_byte = NOWHERE; }
    }
}

} // namespace kidhost
