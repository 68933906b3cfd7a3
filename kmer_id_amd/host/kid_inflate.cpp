// kid_inflate.cpp -- see kid_inflate.h.  The format is RFC 1951 (deflate) inside RFC 1952 (gzip); the behaviour at
// the edges (members, garbage, truncation, messages) is zlib 1.2.11's gzread, which is what the reference calls.
#include "kid_inflate.h"

#include <errno.h>
#include <fcntl.h>
#include <immintrin.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <vector>

#include "kid_inflate_internal.h"
#include "kid_textio.h"

namespace kidhost {

// ---------------------------------------------------------------- CRC-32
// Folding with carry-less multiplies (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ",
// Intel 2009), constants for the reflected polynomial 0xEDB88320.  `crc` is the register as it is kept between calls
// (the bitwise NOT of the published value).  n >= 64 and a multiple of 16.
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_clmul(uint32_t crc, const uint8_t *p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    const __m128i low32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64;
    n -= 64;
    while (n >= 64) { // four lanes, each folded 512 bits ahead
        __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        n -= 64;
    }
    __m128i a;
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00); // the four lanes into one
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x2);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x3);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x4);
    while (n >= 16) {
        a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), _mm_loadu_si128((const __m128i *)p));
        p += 16;
        n -= 16;
    }
    // 128 -> 64 -> 32 bits (Barrett)
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_clmulepi64_si128(_mm_and_si128(x1, low32), k5, 0x00);
    x1 = _mm_xor_si128(x1, t);
    t = _mm_clmulepi64_si128(_mm_and_si128(x1, low32), poly, 0x10);
    t = _mm_clmulepi64_si128(_mm_and_si128(t, low32), poly, 0x00);
    x1 = _mm_xor_si128(x1, t);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n)
{
    static const bool have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && n >= 256) {
        const size_t body = n & ~(size_t)15;
        crc = ~crc32_clmul(~crc, p, body);
        p += body;
        n -= body;
    }
    while (n) { // (zlib takes an unsigned int)
        const size_t step = n > ((size_t)1 << 30) ? ((size_t)1 << 30) : n;
        crc = (uint32_t)::crc32(crc, p, (unsigned)step);
        p += step;
        n -= step;
    }
    return crc;
}

// ---------------------------------------------------------------- the stream
static const size_t IN_CAP = (size_t)4 << 20, IN_LOW = 4096, IN_PAD = 2048;
static const size_t OUT_SLACK = 288; // a pass of the decode loop writes at most 3 literals + a match of 258 + 15 bytes of copy overshoot


struct GzStream::Impl {
    std::string path;
    int fd = -1;
    bool file_eof = false;
    std::vector<uint8_t> ibuf;
    const uint8_t *in = nullptr, *in_end = nullptr;
    uint64_t total_read = 0, base_offset = 0; // bytes read from the file so far, and where in the file reading began
    uint64_t stop_from_bit = ~(uint64_t)0, stopped_bit = 0;
    bool is_stopped = false;
    uint64_t bb = 0;
    unsigned bc = 0;
    enum St { START, DIRECT, MEMBER_HEADER, BLOCK_HEADER, STORED, HUFF, TRAILER, END } st = START;
    bool final_block = false, had_member = false, truncated = false, has_pending = false;
    Fatal pending{0, ""};
    uint32_t stored_left = 0, crc = 0;
    uint64_t member_out = 0;
    size_t hist_len = 0;
    uint8_t hist[kWindow];
    uint32_t lt[LT_SIZE], dt[DT_SIZE];
    uint32_t lt1[1u << LROOT]; // the first level of lt without literal pairs
    const char *huff_msg = nullptr;

    void fill()
    {
        // (the 8 bytes in front of `in` stay: to_byte_boundary() gives back whole bytes that the bit register holds)
        const size_t back = (size_t)(in - ibuf.data()) < 8 ? (size_t)(in - ibuf.data()) : 8;
        const size_t rem = (size_t)(in_end - in) + back;
        if (in - back != ibuf.data()) memmove(ibuf.data(), in - back, rem);
        size_t have = rem;
        while (have < IN_CAP && !file_eof) {
            const ssize_t got = ::read(fd, ibuf.data() + have, IN_CAP - have);
            if (got < 0) {
                if (errno == EINTR) continue;
                throw Fatal{255, path + ": " + strerror(errno)}; // (gzread: Z_ERRNO, gzerror gives the same text)
            }
            if (got == 0) file_eof = true;
            have += (size_t)got;
            total_read += (uint64_t)got;
            if (have >= IN_LOW * 16) break; // enough to go on with; the next fill tops it up
        }
        in = ibuf.data() + back;
        in_end = ibuf.data() + have;
        memset(ibuf.data() + have, 0, IN_PAD);
    }
    size_t avail() const { return (size_t)(in_end - in); }
    [[noreturn]] void bad(const char *msg) { throw Fatal{255, path + ": " + msg}; }

    // bits
    inline void refill()
    {
        bb |= load64(in) << bc;
        in += (63 - bc) >> 3;
        bc |= 56;
    }
    inline unsigned take(unsigned n)
    {
        const unsigned v = (unsigned)(bb & (((uint64_t)1 << n) - 1));
        bb >>= n;
        bc -= n;
        return v;
    }
    void to_byte_boundary() // drop the rest of the current byte, give whole bytes back to the input
    {
        take(bc & 7);
        in -= bc >> 3;
        bb = 0;
        bc = 0;
    }
    bool overran() const { return in > in_end && (size_t)(in - in_end) * 8 > bc; }
    uint64_t bit_position() const // of the next unread bit, in the file
    {
        return (base_offset + total_read) * 8 - (uint64_t)((in_end - in) * 8) - bc;
    }

    int next_byte()
    {
        if (in == in_end) {
            if (!file_eof) fill();
            if (in == in_end) return -1;
        }
        return *in++;
    }

    bool member_header(); // false: the file ended inside the header
    bool block_header();  // false: the file ended inside the header
    template <bool ONE> int huff(uint8_t *&outp, uint8_t *out_lim, const uint8_t *in_lim, const uint8_t *hist_begin);
};

bool GzStream::Impl::member_header()
{
    std::vector<uint8_t> seen;
    auto get = [&]() -> int {
        const int c = next_byte();
        if (c >= 0) seen.push_back((uint8_t)c);
        return c;
    };
    int h[10];
    for (int i = 0; i < 10; i++)
        if ((h[i] = get()) < 0) return false;
    // (h[0], h[1] are the magic: START looked)
    if (h[2] != 8) bad("unknown compression method");
    const int flg = h[3];
    if (flg & 0xe0) bad("unknown header flags set");
    if (flg & 4) { // FEXTRA
        const int a = get(), b = get();
        if (a < 0 || b < 0) return false;
        for (int i = 0, n = a | (b << 8); i < n; i++)
            if (get() < 0) return false;
    }
    for (int bit = 8; bit <= 16; bit <<= 1) // FNAME, FCOMMENT: zero-terminated
        if (flg & bit) {
            int c;
            do {
                if ((c = get()) < 0) return false;
            } while (c != 0);
        }
    if (flg & 2) { // FHCRC: the low 16 bits of the CRC-32 of the header so far
        const uint32_t want = crc32_fast(0, seen.data(), seen.size()) & 0xffff;
        const int a = next_byte(), b = next_byte();
        if (a < 0 || b < 0) return false;
        if ((uint32_t)(a | (b << 8)) != want) bad("header crc mismatch");
    }
    return true;
}

bool GzStream::Impl::block_header()
{
    refill();
    final_block = take(1) != 0;
    const unsigned type = take(2);
    if (type == 0) {
        to_byte_boundary();
        if (in > in_end || avail() < 4) return false; // (fill() keeps IN_LOW bytes ahead unless the file has ended)
        const unsigned len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
        if ((len ^ 0xffff) != nlen) bad("invalid stored block lengths");
        in += 4;
        stored_left = len;
        st = STORED;
        return true;
    }
    if (type == 3) {
        if (overran()) return false;
        bad("invalid block type");
    }
    uint8_t lens[288 + 32];
    unsigned nlit, ndist, mx;
    if (type == 1) {
        for (unsigned s = 0; s < 288; s++) lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        for (unsigned s = 0; s < 32; s++) lens[288 + s] = 5;
        nlit = 288;
        ndist = 32;
    } else {
        nlit = take(5) + 257;
        ndist = take(5) + 1;
        const unsigned ncl = take(4) + 4;
        if (overran()) return false;
        if (nlit > 286 || ndist > 30) bad("too many length or distance symbols");
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (unsigned i = 0; i < ncl; i++) {
            refill();
            cl[order[i]] = (uint8_t)take(3);
        }
        uint32_t ct[128];
        const int rc = build_table(ct, 7, 128, cl, 19, [](unsigned s) { return E_LIT | (s << 16); }, &mx);
        if (overran()) return false;
        if (rc == 3) bad("invalid code -- missing end-of-block"); // (zlib reads every length as 0 from an empty code, then says this)
        if (rc != 0) bad("invalid code lengths set");
        unsigned i = 0;
        while (i < nlit + ndist) {
            refill();
            const uint32_t e = ct[bb & 127];
            take(e & 0xff);
            const unsigned sym = (e >> 16) & 0xff;
            if (overran()) return false;
            if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
            unsigned rep, val = 0;
            if (sym == 16) {
                if (i == 0) bad("invalid bit length repeat");
                val = lens[i - 1];
                rep = 3 + take(2);
            } else if (sym == 17) rep = 3 + take(3);
            else rep = 11 + take(7);
            if (overran()) return false;
            if (i + rep > nlit + ndist) bad("invalid bit length repeat");
            while (rep--) lens[i++] = (uint8_t)val;
        }
        if (lens[256] == 0) bad("invalid code -- missing end-of-block");
        // the distance lengths behind the literal/length ones -> their own array position
        memmove(lens + 288, lens + nlit, ndist);
    }
    int rc = build_table(lt, LROOT, LT_SIZE, lens, nlit, litlen_entry, &mx);
    if (rc == 1 || rc == 3 || (rc == 2 && mx != 1)) bad("invalid literal/lengths set");
    rc = build_table(dt, DROOT, DT_SIZE, lens + 288, ndist, dist_entry, &mx);
    if (rc == 1 || (rc == 2 && mx != 1)) bad("invalid distances set"); // (no distance code at all is fine until a match wants one)
    pair_literals(lt, lt1);
    st = HUFF;
    return true;
}

// The decode loop.  1 = end of block, 0 = a limit was reached (or, ONE, one symbol was decoded), -1 = bad data (huff_msg).
template <bool ONE>
int GzStream::Impl::huff(uint8_t *&outp, uint8_t *out_lim, const uint8_t *in_lim, const uint8_t *hist_begin)
{
    uint64_t b = bb;
    unsigned c = bc;
    const uint8_t *ip = in;
    uint8_t *out = outp;
    const uint32_t *const L = lt, *const D = dt, *const L1 = ONE ? lt1 : lt;
    const unsigned LMASK = (1u << LROOT) - 1, DMASK = (1u << DROOT) - 1;
    int rc = 0;
// 56 bits or more in the register.  It only adds bits above the valid ones: an entry looked up before stays the right one.
#define KID_REFILL() do { b |= load64(ip) << c; ip += (63 - c) >> 3; c |= 56; } while (0)
#define KID_DROP(e) do { b >>= ((e) & 0xff); c -= ((e) & 0xff); } while (0)
// one literal or two: both bytes are stored, the pointer moves past the ones that count
#define KID_PUT(e) do { const uint16_t two = (uint16_t)((e) >> 16); memcpy(out, &two, 2); out += 1 + (((e) >> 12) & 1); } while (0)
// base + the extra bits that lie above the code word in `from`
#define KID_VALUE(e, from) (((e) >> 16) + (unsigned)(((from) & (((uint64_t)1 << ((e) & 0xff)) - 1)) >> (((e) >> 8) & 15)))
    KID_REFILL();
    uint32_t e = L1[b & LMASK];
    while (ip < in_lim && out < out_lim) {
        if (e & E_LIT) { // up to three loads of one or two literals from the first-level table: at most 33 of the 56 bits
            KID_DROP(e);
            KID_PUT(e);
            if (ONE) break;
            e = L1[b & LMASK];
            if (e & E_LIT) {
                KID_DROP(e);
                KID_PUT(e);
                e = L1[b & LMASK];
                if (e & E_LIT) {
                    KID_DROP(e);
                    KID_PUT(e);
                    KID_REFILL();
                    e = L1[b & LMASK];
                    continue;
                }
            }
            KID_REFILL();
        }
        // 56 bits from here: a length (15 + 5) and a distance (15 + 13) need 48
        if (e & E_SUB) {
            KID_DROP(e);
            e = L[(e >> 16) + (b & ((1u << ((e >> 8) & 15)) - 1))];
            if (e & E_LIT) {
                KID_DROP(e);
                *out++ = (uint8_t)(e >> 16);
                if (ONE) break;
                KID_REFILL();
                e = L1[b & LMASK];
                continue;
            }
        }
        if (!(e & E_BASE)) {
            if (e & E_EOB) { KID_DROP(e); rc = 1; break; }
            huff_msg = "invalid literal/length code";
            rc = -1;
            break;
        }
        const unsigned len = KID_VALUE(e, b);
        KID_DROP(e);
        e = D[b & DMASK];
        if (e & E_SUB) {
            KID_DROP(e);
            e = D[(e >> 16) + (b & ((1u << ((e >> 8) & 15)) - 1))];
        }
        if (!(e & E_BASE)) {
            huff_msg = "invalid distance code";
            rc = -1;
            break;
        }
        const unsigned dist = KID_VALUE(e, b);
        KID_DROP(e);
        if (dist > (size_t)(out - hist_begin)) {
            huff_msg = "invalid distance too far back";
            rc = -1;
            break;
        }
        KID_REFILL(); // the next symbol's entry is on its way while the match is copied
        e = L1[b & LMASK];
        const uint8_t *s = out - dist;
        uint8_t *const end = out + len;
        if (dist >= 8) { // words; a word's source lies at least 8 bytes behind it, so overlapping matches come out right
            memcpy(out, s, 8);
            memcpy(out + 8, s + 8, 8);
            if (len > 16) {
                out += 16;
                s += 16;
                do {
                    memcpy(out, s, 8);
                    out += 8;
                    s += 8;
                } while (out < end);
            }
        } else if (dist == 1) {
            memset(out, *s, len);
        } else {
            do *out++ = *s++;
            while (out < end);
        }
        out = end;
        if (ONE) break;
    }
#undef KID_REFILL
#undef KID_DROP
#undef KID_PUT
#undef KID_VALUE
    bb = b;
    bc = c;
    in = ip;
    outp = out;
    return rc;
}

GzStream::GzStream(const std::string &path) : impl_(new Impl())
{
    impl_->path = path;
    impl_->fd = ::open(path.c_str(), O_RDONLY);
    if (impl_->fd < 0) throw Fatal{255, "cannot open " + path};
#ifdef POSIX_FADV_SEQUENTIAL
    posix_fadvise(impl_->fd, 0, 0, POSIX_FADV_SEQUENTIAL);
#endif
    impl_->ibuf.resize(IN_CAP + IN_PAD);
    impl_->in = impl_->in_end = impl_->ibuf.data();
}

GzStream::GzStream(const std::string &path, const GzResume &from) : GzStream(path)
{
    Impl &z = *impl_;
    z.base_offset = from.bit_offset / 8;
    if (lseek(z.fd, (off_t)z.base_offset, SEEK_SET) < 0) throw Fatal{255, path + ": " + strerror(errno)};
    z.fill();
    z.refill();
    z.take((unsigned)(from.bit_offset % 8));
    z.st = Impl::BLOCK_HEADER;
    z.had_member = true;
    z.crc = from.crc;
    z.member_out = from.member_out;
    z.hist_len = from.window_len < kWindow ? from.window_len : kWindow;
    memcpy(z.hist, from.window + (from.window_len - z.hist_len), z.hist_len);
}

void GzStream::stop_at_dynamic_block_from(uint64_t byte_offset) { impl_->stop_from_bit = byte_offset * 8; }
bool GzStream::stopped() const { return impl_->is_stopped; }
uint64_t GzStream::stopped_at_bit() const { return impl_->stopped_bit; }
uint32_t GzStream::member_crc() const { return impl_->crc; }
uint64_t GzStream::member_length() const { return impl_->member_out; }
size_t GzStream::history(uint8_t *dst) const
{
    memcpy(dst, impl_->hist, impl_->hist_len);
    return impl_->hist_len;
}

GzStream::~GzStream()
{
    if (impl_->fd >= 0) ::close(impl_->fd);
}

void GzStream::close()
{
    if (impl_->fd >= 0) {
        ::close(impl_->fd);
        impl_->fd = -1;
    }
    if (impl_->truncated) throw Fatal{255, "failed gzclose"};
}

uint64_t GzStream::bytes_in() const { return impl_->in < impl_->in_end ? impl_->total_read - impl_->avail() : impl_->total_read; }

size_t GzStream::read(uint8_t *dst, size_t cap)
{
    Impl &z = *impl_;
    if (z.has_pending) throw z.pending;
    if (z.st == Impl::END || z.is_stopped) return 0;
    if (cap < kMinRead) throw Fatal{1, "GzStream::read: room for less than 4096 bytes"};
    if (z.hist_len) memcpy(dst - z.hist_len, z.hist, z.hist_len);
    const uint8_t *hist_begin = dst - z.hist_len;
    uint8_t *out = dst, *const out_end = dst + cap, *const out_lim = out_end - OUT_SLACK;
    uint8_t *crc_from = dst;
    auto account = [&]() { // the CRC-32 and the length of the member's text so far
        if (out > crc_from) {
            z.crc = crc32_fast(z.crc, crc_from, (size_t)(out - crc_from));
            z.member_out += (uint64_t)(out - crc_from);
            crc_from = out;
        }
    };
    auto cut_short = [&]() { // the file ended inside a stream
        z.truncated = true;
        z.st = Impl::END;
    };
    try {
        bool more = true;
        while (more) {
            if (!z.file_eof && z.avail() < IN_LOW && z.in <= z.in_end) z.fill();
            switch (z.st) {
            case Impl::START:
                if (z.avail() == 0) { z.st = Impl::END; break; }
                if (z.avail() >= 2 && z.in[0] == 0x1f && z.in[1] == 0x8b) z.st = Impl::MEMBER_HEADER;
                else z.st = z.had_member ? Impl::END : Impl::DIRECT; // (what follows a member and is no header is ignored)
                break;
            case Impl::DIRECT: {
                if (z.avail() == 0) { z.st = Impl::END; break; }
                const size_t n = z.avail() < (size_t)(out_end - out) ? z.avail() : (size_t)(out_end - out);
                memcpy(out, z.in, n);
                out += n;
                crc_from = out;
                z.in += n;
                if (out == out_end) more = false;
                break;
            }
            case Impl::MEMBER_HEADER:
                account();
                if (!z.member_header()) { cut_short(); break; }
                z.had_member = true;
                z.crc = 0;
                z.member_out = 0;
                hist_begin = out;
                z.st = Impl::BLOCK_HEADER;
                break;
            case Impl::BLOCK_HEADER:
                if (z.bit_position() >= z.stop_from_bit && !(z.in > z.in_end)) {
                    z.refill(); // (a look at the three header bits; nothing is consumed)
                    if (((z.bb >> 1) & 3) == 2 && !z.overran()) {
                        z.is_stopped = true;
                        z.stopped_bit = z.bit_position();
                        more = false;
                        break;
                    }
                }
                if (!z.block_header()) cut_short();
                break;
            case Impl::STORED: {
                if (z.stored_left == 0) { z.st = z.final_block ? Impl::TRAILER : Impl::BLOCK_HEADER; break; }
                if (z.avail() == 0) { cut_short(); break; }
                size_t n = z.stored_left;
                if (n > z.avail()) n = z.avail();
                if (n > (size_t)(out_end - out)) n = (size_t)(out_end - out);
                memcpy(out, z.in, n);
                out += n;
                z.in += n;
                z.stored_left -= (uint32_t)n;
                if (out == out_end) more = false;
                break;
            }
            case Impl::HUFF: {
                if (out >= out_lim) { more = false; break; }
                int rc;
                if (z.avail() > 64 && z.in <= z.in_end) {
                    rc = z.huff<false>(out, out_lim, z.in_end - 32, hist_begin);
                } else if (!z.file_eof) {
                    continue; // (fill() at the top)
                } else { // the last bytes of the file: a symbol at a time, and a symbol the file does not hold in full is not decoded
                    const uint64_t b0 = z.bb;
                    const unsigned c0 = z.bc;
                    const uint8_t *const i0 = z.in;
                    uint8_t *const o0 = out;
                    rc = z.huff<true>(out, out_lim, z.in_end + 64, hist_begin);
                    if (z.overran()) {
                        z.bb = b0; z.bc = c0; z.in = i0; out = o0;
                        cut_short();
                        break;
                    }
                }
                if (rc < 0) z.bad(z.huff_msg);
                if (rc == 1) z.st = z.final_block ? Impl::TRAILER : Impl::BLOCK_HEADER;
                break;
            }
            case Impl::TRAILER: {
                z.to_byte_boundary();
                if (z.in > z.in_end || z.avail() < 8) { cut_short(); break; }
                account();
                const uint32_t want_crc = (uint32_t)z.in[0] | ((uint32_t)z.in[1] << 8) | ((uint32_t)z.in[2] << 16) | ((uint32_t)z.in[3] << 24);
                const uint32_t want_len = (uint32_t)z.in[4] | ((uint32_t)z.in[5] << 8) | ((uint32_t)z.in[6] << 16) | ((uint32_t)z.in[7] << 24);
                if (want_crc != z.crc) z.bad("incorrect data check");
                if (want_len != (uint32_t)z.member_out) z.bad("incorrect length check");
                z.in += 8;
                z.st = Impl::START;
                break;
            }
            case Impl::END:
                more = false;
                break;
            }
        }
    } catch (const Fatal &f) {
        if (out == dst) throw;
        z.pending = f; // the text in front of the damage first
        z.has_pending = true;
    }
    account();
    const size_t reach = (size_t)(out - hist_begin), keep = reach < kWindow ? reach : kWindow;
    if (z.st == Impl::DIRECT || z.st == Impl::END) z.hist_len = 0;
    else {
        memcpy(z.hist, out - keep, keep);
        z.hist_len = keep;
    }
    return (size_t)(out - dst);
}

} // namespace kidhost
