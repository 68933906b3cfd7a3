#include "kid_driver.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <iostream>
#include <mutex>
#include <thread>

namespace kidhost {

void leave_now(int exit_code)
{
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    _exit(exit_code);
}

void die_kid(int rc)
{
    std::cerr << "kmer_id_amd: " << kid_strerror(rc) << ": " << kid_last_error() << "\n";
    // (a quality line shorter than its sequence, found by the GPU's process_qual: the reference dies in std::string::at,
    //  abort -> 134; "out of memory in table" is exit 1, newkmer_10nx.cpp:256-260)
    exit(rc == KID_ERR_TABLE_FULL ? 1 : rc == KID_ERR_FORMAT ? 134 : 3);
}

static double seconds_since(const std::chrono::steady_clock::time_point &t0)
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

void load_database(const std::string &tree_path, const std::string &probes_path, const std::string &cache_path, int k, int ntar,
                   std::vector<int32_t> &parent, ProbeSet &ps, bool *from_cache, int threads, StartupTiming *timing,
                   std::thread *cache_writer)
{
    if (from_cache) *from_cache = false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!cache_path.empty() && load_db_cache(cache_path, tree_path, probes_path, k, ntar, parent, ps)) {
        if (from_cache) *from_cache = true;
        if (timing) timing->cache_read_s = seconds_since(t0);
        return;
    }
    parent = load_tree(tree_path, ntar);
    ps = load_probes_gz(probes_path, k, threads, timing);
    if (cache_path.empty()) return;
    // the cache is written beside the upload and the table build when the caller can wait for it later (it must: `ps`
    // and `parent` are read until the writer is joined)
    auto write = [cache_path, tree_path, probes_path, k, &parent, &ps, timing]() {
        const auto t1 = std::chrono::steady_clock::now();
        if (!save_db_cache(cache_path, tree_path, probes_path, k, parent, ps))
            std::cerr << "kmer_id_amd: could not write the database cache " << cache_path << "\n";
        if (timing) timing->cache_write_s = seconds_since(t1);
    };
    if (cache_writer) *cache_writer = std::thread(write);
    else write();
}

Engine::~Engine()
{
    for (kid_sample *s : samples) kid_sample_destroy(s);
    if (owns_dbs)
        for (kid_db *d : dbs) kid_db_destroy(d);
}

std::unique_ptr<Engine> engine_worker(const Engine &owner)
{
    std::unique_ptr<Engine> e(new Engine());
    e->owns_dbs = false;
    e->dbs = owner.dbs;
    e->db = owner.db;
    e->ntar = owner.ntar;
    e->k = owner.k;
    e->batch_reads = owner.batch_reads;
    e->batch_bases = owner.batch_bases;
    for (kid_db *d : e->dbs) {
        kid_sample *s = nullptr;
        int rc = kid_sample_begin(d, &s);
        if (rc != KID_OK) die_kid(rc);
        e->samples.push_back(s);
    }
    e->sample = e->samples[0];
    return e;
}

std::vector<int> parse_devices(const std::string &list)
{
    std::vector<int> out;
    size_t pos = 0;
    while (pos <= list.size()) {
        size_t comma = list.find(',', pos);
        if (comma == std::string::npos) comma = list.size();
        if (comma > pos) out.push_back(atoi(list.substr(pos, comma - pos).c_str()));
        pos = comma + 1;
    }
    return out;
}

void engine_reset(Engine &e)
{
    for (kid_sample *s : e.samples) {
        int rc = kid_sample_reset(s);
        if (rc != KID_OK) die_kid(rc);
    }
    e.next_sample = 0;
}

bool engine_open(Engine &e, const ProbeSet &ps, const std::vector<int32_t> &parent, int k, int log2_slots, int max_probes,
                 unsigned flags, int device)
{
    e.ntar = (int)parent.size();
    e.k = k;
    int rc = kid_db_build(ps.keys.data(), ps.targets.data(), ps.keys.size(), parent.data(), e.ntar, k, log2_slots, max_probes,
                          flags, device, &e.db);
    if (rc == KID_ERR_TABLE_FULL) return false;
    if (rc != KID_OK) die_kid(rc);
    rc = kid_sample_begin(e.db, &e.sample);
    if (rc != KID_OK) die_kid(rc);
    e.dbs.assign(1, e.db);
    e.samples.assign(1, e.sample);
    return true;
}

bool engine_open(Engine &e, const ProbeSet &ps, const std::vector<int32_t> &parent, int k, int log2_slots, int max_probes,
                 unsigned flags, const std::vector<int> &devices)
{
    if (devices.empty()) { std::cerr << "kmer_id_amd: no device given\n"; exit(2); }
    if (!engine_open(e, ps, parent, k, log2_slots, max_probes, flags, devices[0])) return false;
    for (size_t i = 1; i < devices.size(); i++) { // the reference, pinned into every GPU's HBM: device-to-device copies of the one built table
        kid_db *r = nullptr;
        int rc = kid_db_replicate(e.db, devices[i], &r);
        if (rc != KID_OK) die_kid(rc);
        e.dbs.push_back(r);
        kid_sample *s = nullptr;
        rc = kid_sample_begin(r, &s);
        if (rc != KID_OK) die_kid(rc);
        e.samples.push_back(s);
    }
    return true;
}

struct Prefetcher::Impl {
    struct Slot {
        std::deque<std::unique_ptr<ReadBatch>> q;
        bool done = false, failed = false;
        Fatal failure{0, ""};
        SourceStats stats;
    };
    double waited_s = 0;
    std::vector<SourceOpener> files;
    std::vector<Slot> slots;
    std::vector<std::thread> pool;
    std::mutex m;
    std::condition_variable cv;
    size_t next_file = 0, batch_reads, batch_bases, depth;
    bool stop = false;

    void reader()
    {
        for (;;) {
            size_t i;
            {
                std::lock_guard<std::mutex> lk(m);
                if (stop || next_file >= files.size()) return;
                i = next_file++;
            }
            try {
                std::unique_ptr<ReadSource> src = files[i]();
                if (src) {
                    for (;;) {
                        std::unique_ptr<ReadBatch> b(new ReadBatch());
                        if (!src->fill(*b, batch_reads, batch_bases)) break;
                        std::unique_lock<std::mutex> lk(m);
                        cv.wait(lk, [&] { return stop || slots[i].q.size() < depth; });
                        if (stop) return;
                        slots[i].q.push_back(std::move(b));
                        cv.notify_all();
                    }
                    src->close();
                    const SourceStats st = src->stats();
                    std::lock_guard<std::mutex> lk(m);
                    slots[i].stats = st;
                }
            } catch (const Fatal &f) {
                std::lock_guard<std::mutex> lk(m);
                slots[i].failed = true;
                slots[i].failure = f;
            }
            std::lock_guard<std::mutex> lk(m);
            slots[i].done = true;
            cv.notify_all();
        }
    }
};

Prefetcher::Prefetcher(std::vector<SourceOpener> files, int threads, size_t batch_reads, size_t batch_bases, size_t depth)
    : impl_(new Impl())
{
    impl_->files = std::move(files);
    impl_->slots = std::vector<Impl::Slot>(impl_->files.size());
    impl_->batch_reads = batch_reads;
    impl_->batch_bases = batch_bases;
    impl_->depth = depth < 1 ? 1 : depth;
    if (threads < 1) threads = 1;
    if ((size_t)threads > impl_->files.size()) threads = (int)impl_->files.size();
    for (int t = 0; t < threads; t++) impl_->pool.emplace_back([this] { impl_->reader(); });
}

Prefetcher::~Prefetcher()
{
    {
        std::lock_guard<std::mutex> lk(impl_->m);
        impl_->stop = true;
        impl_->cv.notify_all();
    }
    for (std::thread &t : impl_->pool) t.join();
}

SourceStats Prefetcher::file_stats(size_t index)
{
    std::lock_guard<std::mutex> lk(impl_->m);
    return impl_->slots[index].stats;
}

double Prefetcher::seconds_waited() const { return impl_->waited_s; }

std::unique_ptr<ReadBatch> Prefetcher::next(size_t index)
{
    Impl::Slot &s = impl_->slots[index];
    std::unique_lock<std::mutex> lk(impl_->m);
    const auto t0 = std::chrono::steady_clock::now();
    impl_->cv.wait(lk, [&] { return !s.q.empty() || s.done; });
    impl_->waited_s += seconds_since(t0);
    if (!s.q.empty()) {
        std::unique_ptr<ReadBatch> b = std::move(s.q.front());
        s.q.pop_front();
        impl_->cv.notify_all();
        return b;
    }
    if (s.failed) throw s.failure; // what was read before the failure has been handed out, as in the reference
    return nullptr;
}

std::unique_ptr<ReadBatch> Prefetcher::next_any(size_t lo, size_t hi, size_t &which)
{
    std::unique_lock<std::mutex> lk(impl_->m);
    for (;;) {
        bool all_done = true;
        for (size_t i = lo; i < hi; i++) {
            Impl::Slot &s = impl_->slots[i];
            if (!s.q.empty()) {
                std::unique_ptr<ReadBatch> b = std::move(s.q.front());
                s.q.pop_front();
                impl_->cv.notify_all();
                which = i;
                return b;
            }
            if (!s.done) all_done = false;
        }
        // a failure surfaces when everything in front of it has been read to its end and handed out
        for (size_t i = lo; i < hi; i++) {
            Impl::Slot &s = impl_->slots[i];
            if (!s.done) break;
            if (s.failed) { which = i; throw s.failure; }
        }
        if (all_done) return nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        impl_->cv.wait(lk);
        impl_->waited_s += seconds_since(t0);
    }
}

// A FASTQ block's record index and its results travel through ONE buffer of text memory (page-locked once a front-end
// has installed the library's allocator): [records | final target | start | stop].  Copies to and from pageable memory
// would make every "asynchronous" call wait for the GPU.
static int submit_fastq_block(kid_sample *sample, ReadBatch &b, HostBuf &io, uint64_t *ticket)
{
    FastqBlock &fb = *b.fq;
    const size_t nr = b.size();
    io.resize(nr * (sizeof(kid_fastq_rec) + 12) + 64);
    kid_fastq_rec *recs = (kid_fastq_rec *)io.data();
    memcpy(recs, fb.recs.data(), nr * sizeof(kid_fastq_rec));
    uint32_t *fin = (uint32_t *)(recs + nr);
    int32_t *start = (int32_t *)(fin + nr), *stop = start + nr;
    return kid_classify_fastq_async(sample, (const uint8_t *)fb.text.data(), fb.used, recs, nr, fin, start, stop, ticket);
}
static void collect_fastq_block(ReadBatch &b, const HostBuf &io, std::vector<uint32_t> &final_targ)
{
    const size_t nr = b.size();
    const uint32_t *fin = (const uint32_t *)((const kid_fastq_rec *)io.data() + nr);
    const int32_t *start = (const int32_t *)(fin + nr), *stop = start + nr;
    final_targ.assign(fin, fin + nr);
    b.start.assign(start, start + nr);
    b.stop.assign(stop, stop + nr);
}

void run_files_together(Engine &e, Prefetcher &pf, size_t first, size_t count, ReadSaver &saver, std::vector<long long> &handed,
                        const std::function<void(size_t)> &done)
{
    struct InFlight {
        std::unique_ptr<ReadBatch> batch;
        std::vector<uint32_t> final_targ;
        HostBuf io;
        uint64_t ticket = 0;
        kid_sample *sample = nullptr;
        size_t file = 0;
        bool last_of_file = false;
    };
    std::deque<InFlight> q;
    handed.assign(count, 0);
    const size_t max_in_flight = 2 * e.samples.size();
    std::vector<char> ended(count, 0);   // no more batches will come from the file
    std::vector<size_t> pending(count, 0); // batches of the file still in flight
    size_t next_done = 0;
    auto announce = [&]() { // files that are through, in file order
        while (next_done < count && ended[next_done] && pending[next_done] == 0) {
            saver.file_done(next_done);
            done(next_done);
            next_done++;
        }
    };
    auto retire = [&]() {
        InFlight &f = q.front();
        const auto t0 = std::chrono::steady_clock::now();
        int rc = kid_classify_wait(f.sample, f.ticket);
        e.gpu_wait_s += seconds_since(t0);
        if (rc != KID_OK) die_kid(rc);
        if (f.batch->fq) collect_fastq_block(*f.batch, f.io, f.final_targ);
        handed[f.file] += saver.add_batch_of(f.file, *f.batch, f.final_targ, e.k);
        pending[f.file]--;
        q.pop_front();
        announce();
    };
    try {
        for (;;) {
            size_t which = 0;
            std::unique_ptr<ReadBatch> b = pf.next_any(first, first + count, which);
            if (!b) break;
            q.emplace_back();
            InFlight &f = q.back();
            f.batch = std::move(b);
            f.file = which - first;
            pending[f.file]++;
            const size_t nr = f.batch->size();
            f.sample = e.samples[e.next_sample]; // batches are dealt round-robin over the devices
            e.next_sample = (e.next_sample + 1) % e.samples.size();
            int rc;
            const auto t_sub = std::chrono::steady_clock::now();
            if (f.batch->fq) {
                rc = submit_fastq_block(f.sample, *f.batch, f.io, &f.ticket);
            } else {
                f.final_targ.resize(nr);
                rc = kid_classify_batch_async(f.sample, f.batch->bases.data(), f.batch->offsets.data(), f.batch->start.data(),
                                              f.batch->stop.data(), nr, f.final_targ.data(), &f.ticket);
            }
            e.submit_s += seconds_since(t_sub);
            if (rc != KID_OK) die_kid(rc);
            while (q.size() > max_in_flight) retire();
        }
    } catch (const Fatal &) {
        // a file failed behind the batches handed out so far: those are the library's until waited for, and the
        // reference had processed them (and every file before the failing one) before it met the failure
        while (!q.empty()) retire();
        throw;
    }
    while (!q.empty()) retire();
    for (size_t f = 0; f < count; f++) ended[f] = 1;
    announce();
}

long long run_file(Engine &e, Prefetcher &pf, size_t index, ReadSaver &saver)
{
    // Two batches in flight per device: while the GPU classifies batch b, batch b + 1 is uploaded and the results of
    // batch b - 1 go through the read saver -- in file order, which is what decides the "first 12 reads of a target"
    // (newkmer_10nx.cpp:608-612).  FASTQ files arrive as text blocks with a line index (FastqStream): those are trimmed
    // and classified on the GPU (kid_classify_fastq_async); everything else as reads with their range.
    struct InFlight {
        std::unique_ptr<ReadBatch> batch;
        std::vector<uint32_t> final_targ;
        HostBuf io;
        uint64_t ticket = 0;
        kid_sample *sample = nullptr;
    };
    std::deque<InFlight> q;
    long long n = 0;
    const size_t max_in_flight = 2 * e.samples.size();
    auto retire = [&]() {
        InFlight &f = q.front();
        const auto t0 = std::chrono::steady_clock::now();
        int rc = kid_classify_wait(f.sample, f.ticket);
        e.gpu_wait_s += seconds_since(t0);
        if (rc != KID_OK) die_kid(rc);
        if (f.batch->fq) collect_fastq_block(*f.batch, f.io, f.final_targ);
        n += saver.add_batch(*f.batch, f.final_targ, e.k);
        q.pop_front();
    };
    try {
        while (std::unique_ptr<ReadBatch> b = pf.next(index)) {
            q.emplace_back();
            InFlight &f = q.back();
            f.batch = std::move(b);
            const size_t nr = f.batch->size();
            f.sample = e.samples[e.next_sample]; // batches are dealt round-robin over the devices
            e.next_sample = (e.next_sample + 1) % e.samples.size();
            int rc;
            const auto t_sub = std::chrono::steady_clock::now();
            if (f.batch->fq) {
                rc = submit_fastq_block(f.sample, *f.batch, f.io, &f.ticket);
            } else {
                f.final_targ.resize(nr);
                rc = kid_classify_batch_async(f.sample, f.batch->bases.data(), f.batch->offsets.data(), f.batch->start.data(),
                                              f.batch->stop.data(), nr, f.final_targ.data(), &f.ticket);
            }
            e.submit_s += seconds_since(t_sub);
            if (rc != KID_OK) die_kid(rc);
            while (q.size() > max_in_flight) retire();
        }
    } catch (const Fatal &) {
        // the file failed behind the batches handed out so far: those are the library's until waited for, and the
        // reference had processed them before it met the failure
        while (!q.empty()) retire();
        throw;
    }
    while (!q.empty()) retire();
    return n;
}

void dry_dump_db(FILE *f, const std::vector<int32_t> &parent, const ProbeSet &ps)
{
    fprintf(f, "PARENT %zu\n", parent.size());
    for (size_t i = 0; i < parent.size(); i++)
        if (parent[i] != 1) fprintf(f, "%zu %d\n", i, parent[i]);
    fprintf(f, "PROBES %zu %lld\n", ps.keys.size(), ps.lines_parsed);
    for (size_t i = 0; i < ps.keys.size(); i++) fprintf(f, "%llu %u\n", (unsigned long long)ps.keys[i], ps.targets[i]);
}

void dry_dump_source(FILE *f, const std::string &label, ReadSource &src, size_t batch_reads, int k)
{
    ReadBatch b;
    fprintf(f, "FILE %s\n", label.c_str());
    while (src.fill(b, batch_reads, (size_t)-1)) {
        if (b.fq) { // a FASTQ block: what the GPU would do with it (process_qual, the >= k test) done here on the host
            trim_block_on_host(*b.fq, k, b.start, b.stop);
            const char *base = b.fq->text.data();
            for (size_t r = 0; r < b.size(); r++) {
                if (!(b.stop[r] - b.start[r] >= k)) continue;
                fwrite(base + b.fq->acc_off[r], 1, b.fq->acc_len[r], f);
                fprintf(f, "\t%d\t%d\t", b.start[r], b.stop[r]);
                fwrite(base + b.fq->recs[r].seq_off, 1, b.fq->recs[r].seq_len, f);
                fputc('\n', f);
            }
            continue;
        }
        for (size_t r = 0; r < b.size(); r++) {
            fprintf(f, "%s\t%d\t%d\t", b.acc[r].c_str(), b.start[r], b.stop[r]);
            fwrite(b.bases.data() + b.offsets[r], 1, (size_t)(b.offsets[r + 1] - b.offsets[r]), f);
            fputc('\n', f);
        }
    }
    src.close();
}

void finish_sample(Engine &e, const std::string &result_path)
{
    std::vector<int64_t> g((size_t)e.ntar), u((size_t)e.ntar);
    int rc = e.samples.size() > 1 ? kid_sample_end_merged(e.samples.data(), (int)e.samples.size(), g.data(), u.data())
                                  : kid_sample_end(e.sample, g.data(), u.data());
    if (rc != KID_OK) die_kid(rc);
    write_result(result_path, g, u);
}

} // namespace kidhost
