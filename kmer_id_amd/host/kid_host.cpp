// kid_host.cpp -- see kid_host.h.  Own implementation; the reference lines each piece
// reproduces are cited next to it.
#include "kid_host.h"

#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <fstream>
#include <mutex>
#include <sstream>
#include <thread>

#include "kid_inflate.h"
#include "kid_textio.h"

namespace kidhost {

static const size_t REF_LINE_LIMIT = 0x4000; // BUFLEN, newkmer_10nx.cpp:85

void *huge_map(size_t nbytes)
{
    void *p = mmap(nullptr, nbytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    return p == MAP_FAILED ? nullptr : p;
}
void huge_unmap(void *p, size_t nbytes) { munmap(p, nbytes); }

// ---------------------------------------------------------------- tree / strain list
std::vector<int32_t> load_tree(const std::string &path, int ntar)
{
    std::vector<int32_t> parent((size_t)ntar, 1); // Tree1::Tree1, :101-106
    std::ifstream fin(path);
    if (!fin) return parent; // the reference does not notice a missing tree file (:973-984)
    std::string line;
    int i = 0, j = 0;
    bool j_valid = false;
    while (std::getline(fin, line)) {
        // same extraction the reference performs, so malformed lines behave the same:
        // a failed first number zeroes i and leaves j alone, a failed second zeroes j
        std::stringstream ls(line);
        ls >> i;
        if (ls) { ls >> j; j_valid = true; }
        if (!j_valid) continue; // reference reads an uninitialised j here
        if (j < 0 || j >= ntar) throw Fatal{1, "taxonomy edge names node " + std::to_string(j) + " outside [0," + std::to_string(ntar) + ")"};
        parent[(size_t)j] = i; // Tree1::add_edge, :112-116
    }
    return parent;
}

bool strain_list_present(const std::string &path)
{
    std::ifstream fin(path);
    return (bool)fin;
}

// ---------------------------------------------------------------- gz line reader
GzLines::GzLines(const std::string &path) : gz_(new GzStream(path)) // (throws when the file cannot be opened: exit 255 like the reference's gzread(NULL))
{
    buf_.resize(GzStream::kWindow + (1 << 20));
}

GzLines::~GzLines() {}

void GzLines::close()
{
    if (gz_) {
        std::unique_ptr<GzStream> gz = std::move(gz_);
        gz->close(); // throws Fatal{255, "failed gzclose"}
    }
}

// The text sits behind GzStream::kWindow bytes of headroom: the stream writes its history in front of where it inflates
// to -- which is the line carried over (the same bytes again) and, in front of that, the headroom.
bool GzLines::fill()
{
    if (eof_) return false;
    char *text = buf_.data() + GzStream::kWindow;
    if (pos_ > 0) { // keep the partial line at the front
        memmove(text, text + pos_, end_ - pos_);
        end_ -= pos_;
        pos_ = 0;
    }
    if (buf_.size() - GzStream::kWindow - end_ < GzStream::kMinRead) {
        buf_.resize(buf_.size() * 2);
        text = buf_.data() + GzStream::kWindow;
    }
    const size_t got = gz_->read((uint8_t *)text + end_, buf_.size() - GzStream::kWindow - end_);
    if (got == 0) { eof_ = true; return false; }
    end_ += got;
    return true;
}

bool GzLines::next(const char *&line, size_t &len)
{
    size_t scanned = 0;
    for (;;) {
        const char *base = buf_.data() + GzStream::kWindow + pos_;
        const char *nl = (const char *)memchr(base + scanned, '\n', end_ - pos_ - scanned);
        if (nl) {
            size_t l = (size_t)(nl - base);
            if (l >= REF_LINE_LIMIT) throw Fatal{255, "Buffer to small for input line lengths"};
            if (l > 0 && base[l - 1] == '\r') l--;
            line = base;
            len = l;
            pos_ += (size_t)(nl - base) + 1;
            return true;
        }
        scanned = end_ - pos_;
        if (scanned >= REF_LINE_LIMIT) throw Fatal{255, "Buffer to small for input line lengths"};
        if (!fill()) return false; // unterminated tail is dropped (:812-813)
    }
}

// ---------------------------------------------------------------- probes
static inline bool is_ws(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f' || c == '\r' || c == '\n'; }

// digits only, at most 9 of them: every integer type in the reference's extraction accepts it
static inline bool fast_uint(const char *&p, const char *e, uint32_t &out)
{
    const char *s = p;
    uint32_t v = 0;
    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (uint32_t)(*p - '0'); p++; }
    if (p == s || p - s > 9) return false;
    out = v;
    return true;
}

namespace {
struct ProbeChunk { // what one block of lines parses to: a worker's own buffers, reused from block to block
    std::vector<uint64_t> keys;
    std::vector<uint32_t> targets;
    long long lines_parsed = 0;
};
}

static void roll_probe(const char *seq, size_t len, uint32_t target, int k, ProbeChunk &ps)
{
    // process_kmer, :619-661: forward key, upper-case ACGT only, every full window is inserted
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    int cpos = 0;
    uint64_t keyF = 0;
    for (size_t i = 0; i < len; i++) {
        int c;
        switch (seq[i]) {
        case 'A': c = 0; break;
        case 'C': c = 1; break;
        case 'G': c = 2; break;
        case 'T': c = 3; break;
        default: c = -1; break;
        }
        if (c < 0) { cpos = 0; keyF = 0; continue; }
        keyF = ((keyF << 2) & mask) | (uint64_t)c;
        if (++cpos == k) {
            ps.keys.push_back(keyF);
            ps.targets.push_back(target);
            cpos--;
        }
    }
}

// one block of whole lines -> entries (the per-line work of process_kmergz, :680-705)
static void parse_probe_block(const char *text, size_t nbytes, int k, ProbeChunk &ps)
{
    ps.keys.clear();
    ps.targets.clear();
    ps.lines_parsed = 0;
    std::string tmp, sequence;
    const char *p0 = text, *end = text + nbytes;
    while (p0 < end) {
        const char *nl = (const char *)memchr(p0, '\n', (size_t)(end - p0));
        if (!nl) break; // (blocks end with a newline)
        size_t len = (size_t)(nl - p0);
        if (len >= REF_LINE_LIMIT) throw Fatal{255, "Buffer to small for input line lengths"};
        const char *line = p0;
        p0 = nl + 1;
        if (len > 0 && line[len - 1] == '\r') len--;
        if (len == 0) continue;
        // fast path: SEQ,uint,uint,uint,char,uint with nothing else on the line
        const char *p = line, *e = line + len;
        const char *s0 = p;
        while (p < e && *p != ',' && !is_ws(*p)) p++;
        uint32_t target = 0, org, position, count;
        bool ok = (p > s0 && p < e && *p == ',');
        const char *s1 = p;
        if (ok) { p++; ok = fast_uint(p, e, target) && p < e && *p == ','; }
        if (ok) { p++; ok = fast_uint(p, e, org) && p < e && *p == ','; }
        if (ok) { p++; ok = fast_uint(p, e, position) && p < e && *p == ','; }
        if (ok) { p++; ok = (p < e && *p != ',' && !is_ws(*p)); }
        if (ok) { p++; ok = (p < e && *p == ','); }
        if (ok) { p++; ok = fast_uint(p, e, count) && p == e; }
        if (ok) {
            roll_probe(s0, (size_t)(s1 - s0), target, k, ps);
            ps.lines_parsed++;
            continue;
        }
        // anything else: the reference's own extraction decides (:695-697)
        tmp.assign(line, len);
        for (char &c : tmp) if (c == ',') c = ' ';
        std::istringstream ss(tmp);
        unsigned int t2;
        int o2, p2, c2;
        char strand;
        if (ss >> sequence >> t2 >> o2 >> p2 >> strand >> c2) {
            roll_probe(sequence.data(), sequence.size(), t2, k, ps);
            ps.lines_parsed++;
        }
    }
}

ProbeSet load_probes_gz(const std::string &path, int k, int threads, StartupTiming *timing)
{
    const auto t_begin = std::chrono::steady_clock::now();
    if (threads <= 0) {
        threads = (int)std::thread::hardware_concurrency();
        if (threads > 8) threads = 8;
    }
    if (threads < 1) threads = 1;
    GzLineBlocks in(path, (size_t)8 << 20, (size_t)threads + 2); // (its constructor throws if the file cannot be opened: exit 255 like the reference)
    // The entries go straight into their final place: address space for as many as the library takes (2^32 - 2) is
    // reserved, a worker parses a block into its own small buffers and then, in block order, claims the next
    // `count` places and copies them in.  No growing arrays, no final concatenation of 1.3 GB.
    ProbeSet ps;
    size_t cap = (size_t)1 << 32;
    while (cap >= ((size_t)1 << 20) && !(ps.keys.reserve(cap) && ps.targets.reserve(cap))) cap >>= 1;
    if (cap < ((size_t)1 << 20)) throw Fatal{1, "out of address space for the database entries"};
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::pair<size_t, TextBlock>> work;
    bool no_more = false;
    size_t committed = 0, n_entries = 0; // blocks whose places are claimed; places claimed so far
    size_t fail_block = (size_t)-1;
    long long lines_parsed = 0;
    Fatal failure{0, ""};
    auto worker = [&]() {
        ProbeChunk out;
        for (;;) {
            std::pair<size_t, TextBlock> job;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return !work.empty() || no_more; });
                if (work.empty()) return;
                job = std::move(work.front());
                work.pop_front();
                cv.notify_all();
            }
            bool bad = false;
            Fatal f{0, ""};
            try { parse_probe_block(job.second.data(), job.second.len, k, out); }
            catch (const Fatal &e) { bad = true; f = e; out.keys.clear(); out.targets.clear(); out.lines_parsed = 0; }
            in.recycle(job.second);
            size_t at;
            bool place;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return committed == job.first; });
                if (bad && job.first < fail_block) { fail_block = job.first; failure = f; }
                if (n_entries + out.keys.size() > cap && fail_block == (size_t)-1) {
                    fail_block = job.first;
                    failure = Fatal{1, "out of memory in table "}; // more entries than any table holds (newkmer_10nx.cpp:256-260)
                }
                at = n_entries;
                place = fail_block == (size_t)-1;
                if (place) { n_entries += out.keys.size(); lines_parsed += out.lines_parsed; }
                committed++;
                cv.notify_all();
            }
            if (place && !out.keys.empty()) {
                memcpy(ps.keys.data() + at, out.keys.data(), out.keys.size() * sizeof(uint64_t));
                memcpy(ps.targets.data() + at, out.targets.data(), out.targets.size() * sizeof(uint32_t));
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker);
    size_t n_blocks = 0;
    uint64_t text_bytes = 0;
    try {
        for (;;) {
            TextBlock block;
            if (!in.next(block)) break;
            text_bytes += block.len;
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return work.size() < (size_t)threads; }); // bounded: the inflate thread sets the pace
            work.emplace_back(n_blocks++, std::move(block));
            cv.notify_all();
        }
        in.close();
    } catch (const Fatal &e) { // a gz error / an over-long line found while cutting: behind every block handed out so far
        std::lock_guard<std::mutex> lk(m);
        if (n_blocks < fail_block) { fail_block = n_blocks; failure = e; }
    }
    {
        std::lock_guard<std::mutex> lk(m);
        no_more = true;
        cv.notify_all();
    }
    for (std::thread &t : pool) t.join();
    if (fail_block != (size_t)-1) throw failure; // (the first failure in file order, as the sequential reader would have met it)
    ps.keys.set_size(n_entries);
    ps.targets.set_size(n_entries);
    ps.lines_parsed = lines_parsed;
    if (timing) {
        timing->inflate_s = in.inflate_seconds();
        timing->parse_wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        timing->parse_threads = threads;
        timing->text_bytes = text_bytes;
    }
    return ps;
}

// ---------------------------------------------------------------- binary database cache
namespace {
struct CacheHeader {
    char magic[8];
    int32_t k, ntar;
    uint64_t n_entries;
    int64_t lines_parsed;
    uint64_t stamp[4]; // tree size, tree mtime (ns), probes size, probes mtime (ns); 0 for a missing file
};

void file_stamp(const std::string &path, uint64_t &size, uint64_t &mtime_ns)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) { size = 0; mtime_ns = 0; return; }
    size = (uint64_t)st.st_size;
    mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ull + (uint64_t)st.st_mtim.tv_nsec;
}
}

bool load_db_cache(const std::string &cache_path, const std::string &tree_path, const std::string &probes_path, int k, int ntar,
                   std::vector<int32_t> &parent, ProbeSet &ps)
{
    FILE *f = fopen(cache_path.c_str(), "rb");
    if (!f) return false;
    CacheHeader h;
    uint64_t stamp[4];
    file_stamp(tree_path, stamp[0], stamp[1]);
    file_stamp(probes_path, stamp[2], stamp[3]);
    bool ok = fread(&h, sizeof(h), 1, f) == 1 && memcmp(h.magic, "KIDX0001", 8) == 0 && h.k == k && h.ntar == ntar &&
              memcmp(h.stamp, stamp, sizeof(stamp)) == 0 && stamp[2] != 0;
    if (ok) {
        parent.resize((size_t)ntar);
        ok = ps.keys.reserve(h.n_entries) && ps.targets.reserve(h.n_entries);
        if (ok) { ps.keys.set_size(h.n_entries); ps.targets.set_size(h.n_entries); }
        ps.lines_parsed = h.lines_parsed;
        ok = ok && fread(parent.data(), sizeof(int32_t), (size_t)ntar, f) == (size_t)ntar &&
             fread(ps.keys.data(), sizeof(uint64_t), h.n_entries, f) == h.n_entries &&
             fread(ps.targets.data(), sizeof(uint32_t), h.n_entries, f) == h.n_entries;
    }
    fclose(f);
    if (!ok) { ps = ProbeSet(); parent.clear(); }
    return ok;
}

bool save_db_cache(const std::string &cache_path, const std::string &tree_path, const std::string &probes_path, int k,
                   const std::vector<int32_t> &parent, const ProbeSet &ps)
{
    const std::string tmp = cache_path + ".tmp";
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    CacheHeader h;
    memset(&h, 0, sizeof(h));
    memcpy(h.magic, "KIDX0001", 8);
    h.k = k;
    h.ntar = (int32_t)parent.size();
    h.n_entries = ps.keys.size();
    h.lines_parsed = ps.lines_parsed;
    file_stamp(tree_path, h.stamp[0], h.stamp[1]);
    file_stamp(probes_path, h.stamp[2], h.stamp[3]);
    bool ok = fwrite(&h, sizeof(h), 1, f) == 1 && fwrite(parent.data(), sizeof(int32_t), parent.size(), f) == parent.size() &&
              fwrite(ps.keys.data(), sizeof(uint64_t), ps.keys.size(), f) == ps.keys.size() &&
              fwrite(ps.targets.data(), sizeof(uint32_t), ps.targets.size(), f) == ps.targets.size();
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), cache_path.c_str()) == 0;
    if (!ok) remove(tmp.c_str());
    return ok;
}

// ---------------------------------------------------------------- trimming
bool trim_read(const std::string &seq, const std::string &qual, int k, int &start, int &stop)
{
    // process_qual, :714-760.  std::string::at() yields (signed) char on this ABI.
    const int n = (int)seq.length();
    if ((int)qual.length() < n) throw Fatal{134, "quality line shorter than its sequence (std::out_of_range in the reference)"};
    const signed char *q = (const signed char *)qual.data();
    const signed char cut = 32 + 17;
    const int wcut = 17 * 4;
    stop = n - 1;
    start = 0;
    while (q[start] < cut && start < stop) start++;
    while (q[stop] < cut && stop > start) stop--;
    if (start < stop - 4) {
        int w = (q[start] - 32) + (q[start + 1] - 32) + (q[start + 2] - 32) + (q[start + 3] - 32);
        while (w < wcut && start < stop - 4) { w += q[start + 4] - q[start]; start++; }
    }
    if (start < stop - 4) {
        int w = (q[stop] - 32) + (q[stop - 1] - 32) + (q[stop - 2] - 32) + (q[stop - 3] - 32);
        while (w < wcut && start < stop - 4) { w += q[stop - 4] - q[stop]; stop--; }
    }
    return stop - start >= k;
}

// ---------------------------------------------------------------- FASTQ stream
FastqStream::FastqStream(const std::string &path, int k, size_t block_bytes) : in_(path, block_bytes), k_(k) {}

bool FastqStream::fill(ReadBatch &out, size_t, size_t)
{
    out.clear();
    for (;;) {
        std::unique_ptr<FastqBlock> fb(new FastqBlock());
        if (!in_.next(fb->text)) return false; // (lines of an unfinished record at the end of the file: never a read, :790-802)
        const auto t0 = std::chrono::steady_clock::now();
        if (!carry_.empty()) { fb->text.prepend(carry_.data(), carry_.size()); carry_.clear(); }
        const char *const base = fb->text.data(), *const end = base + fb->text.len;
        if (fb->text.len >= 0xFFFFFFFFull) throw Fatal{255, "FASTQ block of 4 GiB or more"};
        fb->recs.reserve(fb->text.len / 300 + 16);
        fb->acc_off.reserve(fb->text.len / 300 + 16);
        fb->acc_len.reserve(fb->text.len / 300 + 16);
        const char *p = base, *rec_start = base;
        int phase = 0;
        kid_fastq_rec rc{0, 0, 0, 0};
        uint32_t a_off = 0, a_len = 0;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
            if (!nl) break; // (blocks end with a newline)
            size_t l = (size_t)(nl - p);
            if (l >= REF_LINE_LIMIT) throw Fatal{255, "Buffer to small for input line lengths"};
            if (l > 0 && p[l - 1] == '\r') l--;
            if (l > 0) { // blank lines do not advance the record phase (:788)
                switch (phase) {
                case 0: rec_start = p; a_off = (uint32_t)(p - base); a_len = (uint32_t)l; break;
                case 1: rc.seq_off = (uint32_t)(p - base); rc.seq_len = (uint32_t)l; break;
                case 3:
                    rc.qual_off = (uint32_t)(p - base);
                    rc.qual_len = (uint32_t)l;
                    fb->recs.push_back(rc);
                    fb->acc_off.push_back(a_off);
                    fb->acc_len.push_back(a_len);
                    break;
                default: break;
                }
                phase = (phase + 1) & 3;
                if (phase == 0) rec_start = nl + 1;
            }
            p = nl + 1;
        }
        // an unfinished record goes in front of the next block
        fb->used = (size_t)(rec_start - base);
        if (phase != 0) carry_.assign(rec_start, end);
        else fb->used = fb->text.len;
        index_s_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (fb->recs.empty()) { in_.recycle(fb->text); continue; } // (a block of blank lines, or one piece of a huge record)
        out.fq = std::move(fb);
        return true;
    }
}

void trim_block_on_host(const FastqBlock &b, int k, std::vector<int32_t> &start, std::vector<int32_t> &stop)
{
    const size_t n = b.recs.size();
    start.resize(n);
    stop.resize(n);
    const char *base = b.text.data();
    std::string seq, qual;
    for (size_t r = 0; r < n; r++) {
        seq.assign(base + b.recs[r].seq_off, b.recs[r].seq_len);
        qual.assign(base + b.recs[r].qual_off, b.recs[r].qual_len);
        int st, sp;
        trim_read(seq, qual, k, st, sp);
        start[r] = st;
        stop[r] = sp;
    }
}

// ---------------------------------------------------------------- FASTA(.gz)
FastaGzStream::FastaGzStream(const std::string &path, int k) : lines_(path), k_(k) {}

static void push_whole_read(ReadBatch &out, const std::string &seq, const std::string &acc)
{
    out.bases.insert(out.bases.end(), seq.begin(), seq.end());
    out.offsets.push_back(out.bases.size());
    out.start.push_back(0);
    out.stop.push_back((int32_t)seq.size() - 1);
    out.acc.push_back(acc);
}

bool FastaGzStream::fill(ReadBatch &out, size_t max_reads, size_t max_bases)
{
    out.clear();
    if (eof_) return false;
    const char *line;
    size_t len;
    while (out.size() < max_reads && out.bases.size() < max_bases) {
        if (!lines_.next(line, len)) {
            eof_ = true;
            if ((int)seq_.length() > k_) push_whole_read(out, seq_, acc_); // the last record (:867-870)
            seq_.clear();
            break;
        }
        if (len == 0) continue;
        if (line[0] == '>') {
            if ((int)seq_.length() > k_) push_whole_read(out, seq_, acc_); // :849-852
            seq_.clear();
            acc_.assign(line + 1, len - 1);
        } else {
            seq_.append(line, len);
        }
    }
    return out.size() > 0;
}

// ---------------------------------------------------------------- plain FASTA / FASTQ
struct PlainTokenStream::Impl {
    std::ifstream fin;
};

PlainTokenStream::PlainTokenStream(const std::string &path, int k, bool fastq, bool strip_cr)
    : impl_(new Impl()), k_(k), fastq_(fastq), strip_cr_(strip_cr)
{
    impl_->fin.open(path);
    open_ = (bool)impl_->fin;
}

bool PlainTokenStream::fill(ReadBatch &out, size_t max_reads, size_t max_bases)
{
    out.clear();
    if (!open_ || eof_) return false;
    std::string line;
    while (out.size() < max_reads && out.bases.size() < max_bases) {
        if (!std::getline(impl_->fin, line)) {
            eof_ = true;
            if (!fastq_ && (int)seq_.length() > k_) push_whole_read(out, seq_, acc_);
            seq_.clear();
            break;
        }
        if (strip_cr_ && !line.empty() && line.back() == '\r') line.pop_back();
        // the reference's own extraction: a line without a token leaves lseq_ as it was
        std::stringstream ls(line);
        ls >> lseq_;
        if (fastq_) {
            if (lseq_.length() > 0) {
                if (mod4_ == 1) seq_ = lseq_;
                else if (mod4_ == 0) acc_ = lseq_;
                else if (mod4_ == 3) {
                    int st, sp;
                    if (trim_read(seq_, lseq_, k_, st, sp)) {
                        out.bases.insert(out.bases.end(), seq_.begin(), seq_.end());
                        out.offsets.push_back(out.bases.size());
                        out.start.push_back(st);
                        out.stop.push_back(sp);
                        out.acc.push_back(acc_);
                    }
                }
                mod4_ = (mod4_ + 1) % 4;
            }
        } else {
            if (!lseq_.empty() && lseq_[0] == '>') {
                if ((int)seq_.length() > k_) push_whole_read(out, seq_, acc_);
                seq_.clear();
                acc_ = lseq_.substr(1);
            } else {
                seq_ += lseq_;
            }
        }
    }
    return out.size() > 0;
}

// ---------------------------------------------------------------- outputs
void write_result(const std::string &path, const std::vector<int64_t> &gcount, const std::vector<int64_t> &ucount)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return; // an ofstream that failed to open swallows the writes
    for (size_t i = 0; i < gcount.size(); i++) fprintf(f, "%zu,%lld,%lld\n", i, (long long)gcount[i], (long long)ucount[i]);
    fclose(f);
}

ReadSaver::ReadSaver(const std::string &first12_path, int ntar, const std::string &target_path, uint32_t save_target,
                     bool first12_enabled)
    : save_target_(save_target), first12_enabled_(first12_enabled), seen_((size_t)ntar, 0)
{
    if (!first12_path.empty()) f_ = fopen(first12_path.c_str(), "w"); // ofstream::trunc, :1025
    if (!target_path.empty()) f2_ = fopen(target_path.c_str(), "w");
}

ReadSaver::~ReadSaver()
{
    if (f_) fclose(f_);
    if (f2_) fclose(f2_);
}

static void write_saved(FILE *f, uint32_t t, const char *acc, size_t acc_len, const char *seq, size_t seq_len)
{
    fprintf(f, ">%u:", t);
    fwrite(acc, 1, acc_len, f);
    fputc('\n', f);
    fwrite(seq, 1, seq_len, f);
    fputc('\n', f);
}

// one read in the reference's order (:608-613)
void ReadSaver::emit(uint32_t t, const char *acc, size_t acc_len, const char *seq, size_t seq_len)
{
    if (t > 1 && seen_[t] < 12 && f_ && first12_enabled_) write_saved(f_, t, acc, acc_len, seq, seq_len); // SAVENUM, :48,:608
    if (t > 1 && t == save_target_ && f2_) write_saved(f2_, t, acc, acc_len, seq, seq_len);
    seen_[t]++;
}

long long ReadSaver::add_batch_of(size_t file, const ReadBatch &b, const std::vector<uint32_t> &final_targ, int k)
{
    long long handed = 0;
    const bool direct = file == cur_file_;
    if (!direct && later_.size() <= file) later_.resize(file + 1);
    if (!direct && later_[file].count.empty()) later_[file].count.assign(seen_.size(), 0);
    for (size_t r = 0; r < b.size(); r++) {
        if (b.fq && !(b.stop[r] - b.start[r] >= k)) continue; // dropped by process_qual (:757): never reached process_read
        handed++;
        const uint32_t t = final_targ[r];
        const char *acc, *s;
        size_t acc_len;
        if (b.fq) {
            acc = b.fq->text.data() + b.fq->acc_off[r];
            acc_len = b.fq->acc_len[r];
            s = b.fq->text.data() + b.fq->recs[r].seq_off + b.start[r];
        } else {
            acc = b.acc[r].data();
            acc_len = b.acc[r].size();
            s = (const char *)b.bases.data() + b.offsets[r] + b.start[r];
        }
        const size_t seq_len = (size_t)(b.stop[r] - b.start[r] + 1);
        if (direct) { emit(t, acc, acc_len, s, seq_len); continue; }
        // a later file: only its first 12 reads of a target can be among the target's first 12 overall (and every read
        // of the -target file's target is wanted); reads that can be neither only count -- and a count that is past 12
        // decides nothing any more
        if (t <= 1) continue;
        Later &l = later_[file];
        const bool wanted = (l.count[t] < 12 && f_ && first12_enabled_) || (t == save_target_ && f2_);
        if (l.count[t] < 12) l.count[t]++;
        if (wanted) l.held.push_back(Held{t, std::string(acc, acc_len), std::string(s, seq_len)});
    }
    return handed;
}

void ReadSaver::file_done(size_t file)
{
    if (later_.size() <= file) later_.resize(file + 1);
    later_[file].done = true;
    while (cur_file_ < later_.size() && later_[cur_file_].done) { // the next file's turn: what it held back, in its order
        cur_file_++;
        if (cur_file_ < later_.size()) {
            for (const Held &h : later_[cur_file_].held) emit(h.t, h.acc.data(), h.acc.size(), h.seq.data(), h.seq.size());
            later_[cur_file_].held.clear();
            later_[cur_file_].held.shrink_to_fit();
        }
    }
}

// ---------------------------------------------------------------- job lists (kmer_read_vf6.cpp:1021-1057)
namespace {
struct TextLines { // getline with one trailing '\r' removed, like the reference does by hand
    std::ifstream in;
    explicit TextLines(const std::string &path) : in(path) {}
    bool next(std::string &line)
    {
        if (!std::getline(in, line)) return false;
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
};
} // namespace

bool load_job_list(const std::string &path, JobList &out)
{
    out = JobList();
    TextLines text(path);
    if (!text.in) return false;
    std::string line, token; // `token` survives from line to line: an empty file line repeats the previous name
    while (text.next(line)) {
        if (line.length() <= 1) continue;
        int declared = 0;
        {
            std::stringstream fields(line);
            fields >> token >> declared;
        }
        out.header_name.push_back(token);
        out.header_count.push_back(declared);
        out.file_rows.emplace_back();
        std::vector<std::string> &row = out.file_rows[(size_t)out.runnable]; // (see JobList: not necessarily the row just added)
        for (int i = 0; i < declared; i++) {
            line.clear();
            text.next(line); // (past the end of the file: an empty line, the previous name again)
            std::stringstream fields(line);
            fields >> token;
            row.push_back(token);
        }
        if (declared > 0) out.runnable++;
    }
    return true;
}

} // namespace kidhost
