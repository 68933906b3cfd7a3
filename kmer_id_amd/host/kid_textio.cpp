// kid_textio.cpp -- see kid_textio.h
#include "kid_textio.h"

#include "kid_inflate.h"
#include "kid_pargz.h"

#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <stdlib.h>
#include <thread>
#include <utility>

namespace kidhost {

static const size_t REF_LINE_LIMIT = 0x4000; // BUFLEN, newkmer_10nx.cpp:85
static const size_t HEAD = TextBlock::kHeadroomForLine + TextBlock::kHeadroomForRecords;

// ---------------------------------------------------------------- text memory
namespace {
void *plain_alloc(size_t n) { return malloc(n); }
void plain_release(void *p) { free(p); }
struct TextPool {
    std::mutex m;
    void *(*alloc)(size_t) = plain_alloc;
    void (*release)(void *) = plain_release;
    std::vector<std::pair<char *, size_t>> spare; // let go, kept for reuse (at most 64 buffers)
} g_pool;
} // namespace

void set_text_allocator(void *(*alloc)(size_t), void (*release)(void *))
{
    std::lock_guard<std::mutex> lk(g_pool.m);
    for (auto &b : g_pool.spare) g_pool.release(b.first); // (what the old allocator gave goes back to it)
    g_pool.spare.clear();
    g_pool.alloc = alloc ? alloc : plain_alloc;
    g_pool.release = release ? release : plain_release;
}

void HostBuf::resize(size_t n)
{
    if (n <= cap_) return;
    reset();
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        size_t best = g_pool.spare.size(); // the smallest one that is large enough (text blocks and small buffers share the pool)
        for (size_t i = 0; i < g_pool.spare.size(); i++)
            if (g_pool.spare[i].second >= n && (best == g_pool.spare.size() || g_pool.spare[i].second < g_pool.spare[best].second)) best = i;
        if (best < g_pool.spare.size() && g_pool.spare[best].second <= 4 * n + (1 << 16)) {
            p_ = g_pool.spare[best].first;
            cap_ = g_pool.spare[best].second;
            g_pool.spare.erase(g_pool.spare.begin() + (long)best);
            return;
        }
    }
    void *p = g_pool.alloc(n);
    if (!p) throw Fatal{1, "out of memory for a text block"};
    p_ = (char *)p;
    cap_ = n;
}

void HostBuf::reset()
{
    if (!p_) return;
    {
        std::lock_guard<std::mutex> lk(g_pool.m);
        if (g_pool.spare.size() < 64) { g_pool.spare.emplace_back(p_, cap_); p_ = nullptr; cap_ = 0; return; }
    }
    g_pool.release(p_);
    p_ = nullptr;
    cap_ = 0;
}

void TextBlock::prepend(const char *p, size_t n)
{
    if (n > off) throw Fatal{255, "Buffer to small for input line lengths"}; // (a record of lines that long: the reference gave up at the first)
    off -= n;
    len += n;
    memcpy(buf.data() + off, p, n);
}

static std::atomic<int> g_inflate_threads{1};
void set_inflate_threads(int n) { g_inflate_threads = n < 1 ? 1 : n; }

struct GzLineBlocks::Impl {
    std::unique_ptr<GzStream> gz;
    std::unique_ptr<ParallelGz> par;
    uint64_t par_bytes = 0;
    size_t chunk_bytes, depth;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<TextBlock> full;
    std::deque<HostBuf> spare;
    bool done = false, stop = false, failed = false;
    Fatal failure{0, ""};
    std::atomic<uint64_t> ns_inflate{0}, bytes{0};

    void fail(const Fatal &f)
    {
        std::lock_guard<std::mutex> lk(m);
        failed = true;
        failure = f;
        done = true;
        cv.notify_all();
    }

    void run()
    {
        std::vector<char> carry; // the unfinished last line of the chunk before
        for (;;) {
            TextBlock b;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stop || full.size() < depth; });
                if (stop) return;
                if (!spare.empty()) { b.buf = std::move(spare.front()); spare.pop_front(); }
            }
            if (!par && b.buf.size() < HEAD + chunk_bytes) b.buf.resize(HEAD + chunk_bytes);
            const auto t0 = std::chrono::steady_clock::now();
            size_t got = 0;
            try {
                if (par) {
                    if (!par->next(b.buf, got)) got = 0;
                } else {
                    got = gz->read((uint8_t *)b.buf.data() + HEAD, chunk_bytes); // (HEAD >= GzStream::kWindow: the stream's history goes in front)
                }
            } catch (const Fatal &f) {
                fail(f);
                return;
            }
            ns_inflate += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            if (got == 0) { // end of file: the unterminated tail is dropped (:812-813) -- unless the reference's buffer had overflowed on it first
                if (carry.size() >= REF_LINE_LIMIT) { fail(Fatal{255, "Buffer to small for input line lengths"}); return; }
                std::lock_guard<std::mutex> lk(m);
                done = true;
                cv.notify_all();
                return;
            }
            bytes += (uint64_t)got;
            char *base = b.buf.data() + HEAD;
            const char *last = (const char *)memrchr(base, '\n', (size_t)got);
            if (!last) { // no line ends in this chunk
                carry.insert(carry.end(), base, base + got);
                if (carry.size() >= REF_LINE_LIMIT) { fail(Fatal{255, "Buffer to small for input line lengths"}); return; }
                std::lock_guard<std::mutex> lk(m);
                spare.push_back(std::move(b.buf));
                continue;
            }
            const size_t head = (size_t)(last - base) + 1, tail = (size_t)got - head;
            // (carry < 16 KiB here: a longer one ended the run above)
            b.off = HEAD - carry.size();
            b.len = carry.size() + head;
            if (!carry.empty()) memcpy(b.buf.data() + b.off, carry.data(), carry.size());
            carry.assign(base + head, base + head + tail);
            const bool too_long = carry.size() >= REF_LINE_LIMIT;
            {
                std::lock_guard<std::mutex> lk(m);
                full.push_back(std::move(b));
                cv.notify_all();
            }
            if (too_long) { fail(Fatal{255, "Buffer to small for input line lengths"}); return; }
        }
    }
};

GzLineBlocks::GzLineBlocks(const std::string &path, size_t block_bytes, size_t depth, int inflate_threads) : impl_(new Impl())
{
    impl_->chunk_bytes = block_bytes < 2 * REF_LINE_LIMIT ? 2 * REF_LINE_LIMIT : block_bytes;
    const int nt = inflate_threads > 0 ? inflate_threads : g_inflate_threads.load();
    // (both throw when the file cannot be opened: exit 255 like the reference's gzread(NULL))
    if (nt > 1) impl_->par.reset(new ParallelGz(path, nt, (size_t)1 << 20, impl_->chunk_bytes, HEAD)); // (1 MiB pieces: profiles/r03/inflate_threads.txt)
    else impl_->gz.reset(new GzStream(path));
    impl_->depth = depth < 1 ? 1 : depth;
    impl_->th = std::thread([this] { impl_->run(); });
}

GzLineBlocks::~GzLineBlocks()
{
    {
        std::lock_guard<std::mutex> lk(impl_->m);
        impl_->stop = true;
        impl_->cv.notify_all();
    }
    if (impl_->th.joinable()) impl_->th.join();
}

bool GzLineBlocks::next(TextBlock &b)
{
    std::unique_lock<std::mutex> lk(impl_->m);
    impl_->cv.wait(lk, [&] { return !impl_->full.empty() || impl_->done; });
    if (!impl_->full.empty()) {
        if (!b.buf.empty()) impl_->spare.push_back(std::move(b.buf));
        b = std::move(impl_->full.front());
        impl_->full.pop_front();
        impl_->cv.notify_all();
        return true;
    }
    if (impl_->failed) throw impl_->failure;
    return false;
}

void GzLineBlocks::recycle(TextBlock &b)
{
    if (b.buf.empty()) return;
    std::lock_guard<std::mutex> lk(impl_->m);
    impl_->spare.push_back(std::move(b.buf));
    b = TextBlock();
}

void GzLineBlocks::close()
{
    {
        std::lock_guard<std::mutex> lk(impl_->m);
        impl_->stop = true;
        impl_->cv.notify_all();
    }
    if (impl_->th.joinable()) impl_->th.join();
    if (impl_->gz) {
        std::unique_ptr<GzStream> gz = std::move(impl_->gz);
        gz->close(); // throws Fatal{255, "failed gzclose"} for a file that ended inside a stream
    }
    if (impl_->par) {
        std::unique_ptr<ParallelGz> par = std::move(impl_->par);
        impl_->par_bytes = par->bytes_in_parallel();
        par->close();
    }
}

double GzLineBlocks::inflate_seconds() const { return (double)impl_->ns_inflate.load() * 1e-9; }
uint64_t GzLineBlocks::bytes_out() const { return impl_->bytes.load(); }
uint64_t GzLineBlocks::bytes_inflated_in_parallel() const { return impl_->par ? impl_->par->bytes_in_parallel() : impl_->par_bytes; }

} // namespace kidhost
