// kid_inflate.h -- reading a .gz file the way the reference's gzopen / gzread / gzclose do (newkmer_10nx.cpp:673,
// :762-816), with an inflate loop of our own.  zlib 1.2.11 inflates FASTQ text at 0.4-0.5 GB/s on one core and every
// file is one stream: that loop is what the shipped programs wait for.  This one keeps 56+ bits in a
// 64-bit register, looks literals up three at a time in an 11-bit table, copies matches in 8-byte words, and checks
// the CRC-32 with carry-less multiplies: about 2-3 x zlib on the same files.
//
// What a caller of gzread can observe is kept:
//   * concatenated gzip members read as one stream; anything behind a member that is not a gzip header is ignored;
//   * a file that does not start with a gzip header is passed through as it is;
//   * a damaged stream (bad block, bad code, distance too far back, wrong CRC-32 or length) is an error after the
//     data in front of it: read() hands that data out, the next read() throws Fatal{255, "<path>: <zlib's message>"}
//     (the reference: gzread < 0 -> error(gzerror()) -> exit 255, :776);
//   * a file that ends inside a stream is NOT a read error: everything that could be inflated is handed out, read()
//     then says end-of-file, and close() throws Fatal{255, "failed gzclose"} (gzclose returns Z_BUF_ERROR; :815).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <string>

namespace kidhost {

// Where a stream goes on in the middle of a gzip member (kid_pargz.h: the piece of a file that is not inflated in
// parallel): at the block header at `bit_offset` of the file, with the member's text so far as history and in its sums.
struct GzResume {
    uint64_t bit_offset;
    const uint8_t *window; // the last window_len bytes of text in front (window_len <= 32768)
    size_t window_len;
    uint32_t crc;          // CRC-32 of the member's text so far
    uint64_t member_out;   // its length
};

class GzStream {
public:
    static const size_t kWindow = 32768; // bytes in FRONT of a read()'s destination that the stream may write to
    static const size_t kMinRead = 4096; // smallest `cap` read() takes
    // throws Fatal{255} when the file cannot be opened (the reference: gzread(NULL) -> -1 -> exit 255)
    explicit GzStream(const std::string &path);
    GzStream(const std::string &path, const GzResume &from);
    // read() comes back early, in front of the first block header at or behind `byte_offset` of the file that announces
    // a dynamic-code block (the kind a parallel reader can find from the outside); stopped() says so, and where.
    void stop_at_dynamic_block_from(uint64_t byte_offset);
    bool stopped() const;
    uint64_t stopped_at_bit() const;
    uint32_t member_crc() const;      // of the current member's text handed out so far
    uint64_t member_length() const;
    size_t history(uint8_t *dst) const; // the last (up to kWindow) bytes of the current member's text; returns how many
    ~GzStream();
    GzStream(const GzStream &) = delete;
    GzStream &operator=(const GzStream &) = delete;
    // Inflates up to `cap` bytes to `dst`; a short count is normal (the stream stops a few hundred bytes before the
    // end of the room it is given rather than in the middle of a match).  0 = end of the file.  The kWindow bytes in
    // front of `dst` are scratch: the stream puts its history there, matches reach back into it.
    size_t read(uint8_t *dst, size_t cap);
    void close();
    uint64_t bytes_in() const; // compressed bytes consumed so far
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// CRC-32 (the gzip one) of a buffer, continuing from `crc`; carry-less multiply where the CPU has it.
uint32_t crc32_fast(uint32_t crc, const uint8_t *p, size_t n);

} // namespace kidhost
