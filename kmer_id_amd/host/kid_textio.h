// kid_textio.h -- gzip text input as a pipeline: one thread inflates (kid_inflate.h) and cuts the text into blocks of
// whole lines, the consumers work on the blocks in parallel (probes file: parse workers; FASTQ: the record indexer).
// A gzip stream cannot be split, so the inflate thread is the pace of a file; everything behind it can be spread out.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <string>
#include <vector>

namespace kidhost {

// What the reference does when it gives up (newkmer_10nx.cpp:87-91 and friends)
struct Fatal {
    int exit_code;
    std::string message;
};

// Memory for file text.  Plain malloc until a front-end installs another allocator -- nk10 installs the library's
// page-locked memory (kid_host_alloc), so that a block goes to the GPU by DMA straight from where it was inflated to
// (an upload from pageable memory is first copied by the CPU, at 1.6 GB/s of the consumer thread's time).  Buffers are
// kept for reuse when they are let go (page-locking 8 MiB takes milliseconds).
void set_text_allocator(void *(*alloc)(size_t), void (*release)(void *));
class HostBuf {
public:
    HostBuf() {}
    ~HostBuf() { reset(); }
    HostBuf(const HostBuf &) = delete;
    HostBuf &operator=(const HostBuf &) = delete;
    HostBuf(HostBuf &&o) noexcept : p_(o.p_), cap_(o.cap_) { o.p_ = nullptr; o.cap_ = 0; }
    HostBuf &operator=(HostBuf &&o) noexcept
    {
        if (this != &o) { reset(); p_ = o.p_; cap_ = o.cap_; o.p_ = nullptr; o.cap_ = 0; }
        return *this;
    }
    void resize(size_t n); // at least n bytes (contents are not kept)
    void reset();          // back to the pool
    size_t size() const { return cap_; }
    bool empty() const { return p_ == nullptr; }
    char *data() { return p_; }
    const char *data() const { return p_; }
private:
    char *p_ = nullptr;
    size_t cap_ = 0;
};

// A block of whole lines inside a recycled buffer.  The bytes before `off` are headroom: a consumer that carries an
// unfinished record over from the block before copies it there (prepend) instead of copying the block.
struct TextBlock {
    HostBuf buf;
    size_t off = 0, len = 0;
    const char *data() const { return buf.data() + off; }
    char *data() { return buf.data() + off; }
    // put `n` bytes in front of the block (n <= kHeadroomForRecords)
    void prepend(const char *p, size_t n);
    static const size_t kHeadroomForLine = 0x4000;        // the inflater's own carry: the unfinished last line of a chunk
    static const size_t kHeadroomForRecords = 4 * 0x4000; // a consumer's carry: up to three lines of an unfinished record
};

// A thread that inflates one .gz file (or reads a plain file: like gzread, GzStream passes those through) and hands out its text in
// blocks that end with a '\n'.  The rules of the reference's reader (newkmer_10nx.cpp:762-816) that concern raw text:
// a line of 16384 bytes or more is fatal (exit 255, :773), the unterminated tail of the file is dropped (:812-813).
// Failures surface in stream order: next() throws Fatal{255} where the reference's gzread loop would have called
// error() (:87-91, :772-778), after every block before the failure has been handed out.
// inflate_threads: 0 = what set_inflate_threads() said (1 until a front-end says otherwise); 1 = the one inflate thread
// does it all; more = that many helpers inflate pieces of the file side by side (kid_pargz.h), same text, same failures.
void set_inflate_threads(int n);
class GzLineBlocks {
public:
    explicit GzLineBlocks(const std::string &path, size_t block_bytes = (size_t)8 << 20, size_t depth = 4, int inflate_threads = 0);
    ~GzLineBlocks();
    GzLineBlocks(const GzLineBlocks &) = delete;
    GzLineBlocks &operator=(const GzLineBlocks &) = delete;
    // the next block (swapped into `b`, whose old buffer is recycled); false at the end of the file
    bool next(TextBlock &b);
    void recycle(TextBlock &b);      // give a buffer back without asking for the next block
    void close();                    // gzclose; throws Fatal{255} "failed gzclose"
    double inflate_seconds() const;  // time spent inflating so far
    uint64_t bytes_out() const;
    uint64_t bytes_inflated_in_parallel() const;
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

} // namespace kidhost
