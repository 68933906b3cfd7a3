// kid_host.h -- host-side C++ of the nk10-compatible program: database text loaders, the
// FASTQ.gz reader with the reference's line rules, quality trimming and result writers.
// All classification goes through the C ABI (include/kmer_id_amd.h) to the GPU.
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "kid_inflate.h"
#include "kid_textio.h"
#include "kmer_id_amd.h"

namespace kidhost {


// ---------------------------------------------------------------- database text files
// Tree1 after `linestream >> i >> j; add_edge(i,j)` over every line (newkmer_10nx.cpp:973-983).
// A missing file leaves every node under root, silently, like the reference.
std::vector<int32_t> load_tree(const std::string &path, int ntar);

// strain list (newkmer_10nx.cpp:951-971): only its presence is observable ("narin <name>")
bool strain_list_present(const std::string &path);

// A flat array whose final size is not known while it is being filled by several threads (the entries of a 5 GB
// probes file): address space for the most it can ever hold is reserved once (anonymous, no-reserve mapping: only the
// pages that are written cost memory), so it never moves and is never copied.
template <class T>
class HugeVec {
public:
    HugeVec() {}
    ~HugeVec() { release(); }
    HugeVec(const HugeVec &) = delete;
    HugeVec &operator=(const HugeVec &) = delete;
    HugeVec(HugeVec &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    HugeVec &operator=(HugeVec &&o) noexcept
    {
        if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; }
        return *this;
    }
    bool reserve(size_t cap); // false: the mapping was refused
    void set_size(size_t n) { n_ = n; }
    size_t size() const { return n_; }
    size_t capacity() const { return cap_; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    const T &operator[](size_t i) const { return p_[i]; }
private:
    void release();
    T *p_ = nullptr;
    size_t n_ = 0, cap_ = 0;
};
void *huge_map(size_t nbytes);
void huge_unmap(void *p, size_t nbytes);
template <class T> bool HugeVec<T>::reserve(size_t cap)
{
    release();
    if (cap == 0) return true;
    p_ = static_cast<T *>(huge_map(cap * sizeof(T)));
    cap_ = p_ ? cap : 0;
    return p_ != nullptr;
}
template <class T> void HugeVec<T>::release()
{
    if (p_) huge_unmap(p_, cap_ * sizeof(T));
    p_ = nullptr;
    n_ = cap_ = 0;
}
struct ProbeSet {
    HugeVec<uint64_t> keys;    // forward keys in file order (process_kmer, :619-661)
    HugeVec<uint32_t> targets;
    long long lines_parsed = 0;    // tct, printed as "<n> kmers loaded" (:701,:989)
};
// Wall-clock seconds of the start-up phases, for --timing (nk10 prints them to stderr as one JSON line)
struct StartupTiming {
    double inflate_s = 0;      // inside GzStream::read (the inflate thread; overlaps the parse)
    double parse_wall_s = 0;   // probes text -> keys: inflate + parse workers, wall
    double cache_read_s = 0, cache_write_s = 0;
    double gpu_build_s = 0;    // kid_db_build: upload + table build on the GPU
    int parse_threads = 0;
    uint64_t text_bytes = 0;
};
// process_kmergz (newkmer_10nx.cpp:663-712).  One thread inflates, `threads` workers parse blocks of whole lines, the
// entries come out in file order.  Throws Fatal{255} on gz errors / over-long lines.  threads <= 0: one per core, at most 8
ProbeSet load_probes_gz(const std::string &path, int k, int threads = 0, StartupTiming *timing = nullptr);

// Binary cache of the parsed database: what load_tree + load_probes_gz produce, so that a
// later run skips the text parse (minutes for 108 M lines).  The cache is tied to the size and
// modification time of the two text files and to k / ntar; anything else makes it stale.
//   layout: "KIDX0001", k, ntar, n_entries, lines_parsed, 4 file stamps, parent[ntar], keys[n], targets[n]
bool load_db_cache(const std::string &cache_path, const std::string &tree_path, const std::string &probes_path, int k, int ntar,
                   std::vector<int32_t> &parent, ProbeSet &ps);
bool save_db_cache(const std::string &cache_path, const std::string &tree_path, const std::string &probes_path, int k,
                   const std::vector<int32_t> &parent, const ProbeSet &ps);

// ---------------------------------------------------------------- reads
// process_qual (newkmer_10nx.cpp:714-760).  returns true when process_read would be called.
// Throws Fatal{134} where qual.at() would throw (quality shorter than sequence).
bool trim_read(const std::string &seq, const std::string &qual, int k, int &start, int &stop);

// Lines of a (gz or plain) text file under the rules of process_fqgz (newkmer_10nx.cpp:762-816):
// split at '\n', one trailing '\r' removed, empty lines skipped by the caller, the unterminated
// last line dropped, a line of 16384 bytes or more is fatal (exit 255).
class GzLines {
public:
    explicit GzLines(const std::string &path);
    ~GzLines();
    bool next(const char *&line, size_t &len); // false at end of file
    void close();                               // throws Fatal{255} "failed gzclose"
private:
    std::unique_ptr<GzStream> gz_;
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    bool eof_ = false;
    bool fill();
};

// A block of FASTQ text cut at a record boundary, with the lines of its records found (FastqStream): what the host
// does of process_fqgz.  Trimming and classification of the block happen on the GPU (kid_classify_fastq_async).
struct FastqBlock {
    TextBlock text;
    size_t used = 0;                     // bytes of whole records at the front of `text`
    std::vector<kid_fastq_rec> recs;     // per record: sequence line and quality line (offsets into text.data())
    std::vector<uint32_t> acc_off, acc_len; // ... and its header line (with the leading '@'), for _reads.txt
};

struct ReadBatch {
    std::vector<uint8_t> bases;     // whole sequence lines, concatenated
    std::vector<uint64_t> offsets;  // n+1
    std::vector<int32_t> start, stop; // process_qual's range per read (FASTQ blocks: filled in by the GPU)
    std::vector<std::string> acc;   // header lines (with the leading '@'), for _reads.txt
    std::unique_ptr<FastqBlock> fq; // set: the batch is a FASTQ text block, the vectors above are not used (start / stop receive results)
    size_t size() const { return fq ? fq->recs.size() : start.size(); }
    void clear() { bases.clear(); offsets.assign(1, 0); start.clear(); stop.clear(); acc.clear(); fq.reset(); }
};

// A file of reads delivered as batches.  fill() clears `out`, appends reads until max_reads reads or
// max_bases bases are in it, and returns false when the file is exhausted and nothing was appended.
struct SourceStats { // seconds a file's host stages took (--timing)
    double inflate_s = 0; // inside GzStream::read (its own thread)
    double index_s = 0;   // finding lines / parsing / trimming (the reader thread)
    uint64_t text_bytes = 0;
};
class ReadSource {
public:
    virtual ~ReadSource() {}
    virtual bool fill(ReadBatch &out, size_t max_reads, size_t max_bases) = 0;
    virtual void close() {}
    virtual SourceStats stats() const { return SourceStats(); }
};

// process_fqgz (newkmer_10nx.cpp:762-816): FASTQ(.gz).  One thread inflates (GzLineBlocks), fill() finds the lines of
// a block under the reference's rules -- split at '\n', one trailing '\r' removed, empty lines skipped without
// advancing the 4-line phase (:788), the unterminated last line dropped, a line of 16384 bytes or more fatal -- and
// hands the block on as it is: a batch is one FastqBlock.  (max_reads / max_bases are not used: the block size set at
// construction bounds a batch.)
class FastqStream : public ReadSource {
public:
    FastqStream(const std::string &path, int k, size_t block_bytes = (size_t)8 << 20);
    bool fill(ReadBatch &out, size_t max_reads, size_t max_bases = (size_t)-1) override;
    void close() override { in_.close(); }
    SourceStats stats() const override { SourceStats s; s.inflate_s = in_.inflate_seconds(); s.index_s = index_s_; s.text_bytes = in_.bytes_out(); return s; }
private:
    GzLineBlocks in_;
    int k_;
    std::vector<char> carry_; // the lines of an unfinished record at the end of the block before
    double index_s_ = 0;
};
// process_qual on the host for the records of a block (the --dry-run dump, and tests): start / stop per record
void trim_block_on_host(const FastqBlock &b, int k, std::vector<int32_t> &start, std::vector<int32_t> &stop);

// process_fagz (newkmer_10nx.cpp:818-875, kmer_read_vf6.cpp:803-861): FASTA(.gz), multi-line records
// joined, a record is classified whole if it is longer than k
class FastaGzStream : public ReadSource {
public:
    FastaGzStream(const std::string &path, int k);
    bool fill(ReadBatch &out, size_t max_reads, size_t max_bases = (size_t)-1) override;
    void close() override { lines_.close(); }
private:
    GzLines lines_;
    int k_;
    bool eof_ = false;
    std::string seq_, acc_;
};

// process_fa / process_fastq (kmer_read_vf6.cpp:863-935; process_fq in kmer_read_m3.cpp): plain
// text read with getline + `linestream >> token`: only the first whitespace-delimited token of a
// line counts, and a blank line leaves the previous token in place (it is used again).
class PlainTokenStream : public ReadSource {
public:
    // strip_cr: kmer_read_vf6 / kmer_read_m3 drop one trailing '\r' per line; newkmer_10nx's process_fa (:877-913) does not
    PlainTokenStream(const std::string &path, int k, bool fastq, bool strip_cr = true);
    bool present() const { return open_; }
    bool fill(ReadBatch &out, size_t max_reads, size_t max_bases = (size_t)-1) override;
private:
    struct Impl;
    std::shared_ptr<Impl> impl_;
    int k_;
    bool fastq_, strip_cr_, open_, eof_ = false;
    int mod4_ = 0;
    std::string lseq_, seq_, acc_;
};

// ---------------------------------------------------------------- job lists (kmer_read_vf6)
// <jname>/<jname>.txt: a header line "<job> <n>" followed by n lines naming input files, repeated
// (kmer_read_vf6.cpp:1021-1057).  Read with the reference's extraction rules: lines of at most one character
// are skipped where a header is expected, only the first blank-delimited token of a file line counts, a token
// that is missing leaves the previous one in place, a count that does not parse reads as 0.  The reference
// appends a header for every job line but files a job's inputs under the number of jobs THAT HAD FILES so
// far, so headers and file rows drift apart behind a job without files; runnable jobs are the first
// `runnable` headers with the first `runnable` file rows -- kept as is (its outputs are what we reproduce).
struct JobList {
    std::vector<std::string> header_name;
    std::vector<int> header_count;
    std::vector<std::vector<std::string>> file_rows;
    int runnable = 0;
    // inputs of runnable job j, as the reference walks them (:1116-1164)
    int n_inputs(int j) const
    {
        const int declared = header_count[(size_t)j], have = (int)file_rows[(size_t)j].size();
        return declared < have ? declared : have;
    }
};
// false: the file is missing (the reference prints "narin <file>" and goes on without jobs)
bool load_job_list(const std::string &path, JobList &out);

// ---------------------------------------------------------------- outputs
// <prefix>_result.txt: "i,gcount[i],ucount[i]\n" for every target (newkmer_10nx.cpp:1040-1043)
void write_result(const std::string &path, const std::vector<int64_t> &gcount, const std::vector<int64_t> &ucount);

// _reads.txt saver (newkmer_10nx.cpp:608-612): the first 12 reads of every target > 1, file order.
// kmer_read_vf6 (:612-619) writes that file only when no -target is given, and with -target T all
// reads of T go to a second file.  An empty path disables the respective file.
class ReadSaver {
public:
    ReadSaver(const std::string &first12_path, int ntar, const std::string &target_path = "", uint32_t save_target = 0,
              bool first12_enabled = true);
    ~ReadSaver();
    // returns the number of reads of the batch that the reference hands to process_read (all of them, except in a FASTQ
    // block: there the records process_qual drops are still in the batch, with stop - start < k)
    long long add_batch(const ReadBatch &b, const std::vector<uint32_t> &final_targ, int k) { return add_batch_of(0, b, final_targ, k); }
    // Several files of one sample read at the same time (the two mates, nk10): the batches of file f arrive in file
    // order, but files interleave.  What the reference writes depends on the order "all of file 0, then all of file 1":
    // the reads of a later file that can still be among a target's first 12 (the first 12 of that target within their
    // own file) wait in memory until the files before it are through (file_done).
    long long add_batch_of(size_t file, const ReadBatch &b, const std::vector<uint32_t> &final_targ, int k);
    void file_done(size_t file);
private:
    struct Held { uint32_t t; std::string acc, seq; };
    struct Later { std::vector<Held> held; std::vector<uint16_t> count; bool done = false; };
    void emit(uint32_t t, const char *acc, size_t acc_len, const char *seq, size_t seq_len);
    FILE *f_ = nullptr, *f2_ = nullptr;
    uint32_t save_target_ = 0;
    bool first12_enabled_ = true;
    std::vector<int64_t> seen_; // gcount as the reference sees it at that point of the file
    size_t cur_file_ = 0;       // the file whose reads are written as they come
    std::vector<Later> later_;  // [file]: what waits for its turn
};

} // namespace kidhost
