// kid_pargz.h -- one gzip file inflated by several threads.
//
// A deflate stream is a chain: a match copies text that came up to 32 KiB earlier, so a thread cannot simply start in
// the middle.  What it can do (Kerbiriou & Chikhi, "Parallel decompression of gzip-compressed files and random access to
// DNA sequences", 2019; Knespel & Brunst, "Rapidgzip", 2023) is
//   1. find, from some byte offset of the file on, the first bit position where a dynamic-code block header starts
//      (try every bit; a header that parses into two complete Huffman codes is very unlikely to be chance),
//   2. inflate from there into 16-bit symbols: a literal is itself, a copy from text in front of the starting point
//      becomes "byte i of the unknown 32 KiB window" (256 + i), copies of such symbols carry them along,
//   3. and once the piece of the file in front has been done and its last 32 KiB of text are known, replace the window
//      symbols by table lookup -- a pass that runs at memory speed and in parallel itself.
// Pieces are checked against each other: the piece in front must END exactly where this one was started; if it does not
// (a header found by chance, a file of stored blocks, damage, the end of the file) the stretch is inflated again by the
// sequential reader (kid_inflate.h), which is also what decides every error and every message.  What comes out is the
// text gzread would give, in order, with the same failures at the same places.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <string>

#include "kid_textio.h"

namespace kidhost {

class ParallelGz {
public:
    // chunk_bytes: compressed bytes per piece of the file; piece_bytes: most text handed out at a time;
    // head: bytes of headroom a handed-out buffer has in front of its text (>= 32768)
    // patience: after so many pieces in a row that did not fit, the rest of the file is read sequentially (stored
    // blocks -- data that does not compress -- are read faster than headers are looked for in them)
    ParallelGz(const std::string &path, int threads, size_t chunk_bytes, size_t piece_bytes, size_t head, unsigned patience = 4);
    ~ParallelGz();
    ParallelGz(const ParallelGz &) = delete;
    ParallelGz &operator=(const ParallelGz &) = delete;
    // The next stretch of text, in file order: `buf` gets a buffer with `len` bytes of text at offset head.
    // false = end of the file.  Throws Fatal{255} where GzStream::read would.
    bool next(HostBuf &buf, size_t &len);
    void close();                       // throws Fatal{255, "failed gzclose"} for a file cut off inside a stream
    uint64_t bytes_in_parallel() const; // text that came out of pieces inflated side by side (the rest: sequential)
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

} // namespace kidhost
