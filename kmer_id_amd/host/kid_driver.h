// kid_driver.h -- what the three front-ends (nk10, kmer_read_vf6, kmer_read_m3) share: the database
// on the GPU, the reader-thread / GPU-thread pipeline over one input file, closing a sample.
#pragma once
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "kid_host.h"
#include "kmer_id_amd.h"

namespace kidhost {

struct Engine {
    kid_db *db = nullptr;         // the database on the first device
    kid_sample *sample = nullptr; // ... and its sample
    // one replica of the database + one sample per device (--devices a,b,...): [0] are the two above.  The batches of
    // a file are dealt round-robin over the samples, the per-read results come back in file order, and closing a
    // sample merges the replicas' counters (kid_sample_end_merged).
    std::vector<kid_db *> dbs;
    std::vector<kid_sample *> samples;
    size_t next_sample = 0;
    int ntar = 0, k = 30;
    size_t batch_reads = 1 << 20;
    size_t batch_bases = 256u << 20;
    double gpu_wait_s = 0, submit_s = 0; // the consumer, inside kid_classify_wait / kid_classify_*_async (--timing)
    bool owns_dbs = true; // false: a worker of engine_worker() -- its own samples on the owner's databases
    ~Engine();
};
// Another set of samples (one per device) on the databases of `owner`: for a thread that classifies another input
// sample at the same time (nk10 --samples-in-flight).  The owner must outlive it.
std::unique_ptr<Engine> engine_worker(const Engine &owner);

[[noreturn]] void die_kid(int rc);
// The end of a front-end that has written everything: flush the standard streams and leave without unwinding -- giving
// page-locked buffers, device memory and the HIP runtime back one by one takes 0.15 s that the operating system does at once.
[[noreturn]] void leave_now(int exit_code);

// tree + probes, from the binary cache when `cache_path` names a valid one, else from the text files
// (and the cache is written for the next run).  `from_cache` reports which way it went.
// `threads`: parse workers of the probes text (0: one per core, at most 8).  `cache_writer`: if given, a cache that has
// to be (re)written is written on that thread -- join it before `parent` / `ps` go away.
void load_database(const std::string &tree_path, const std::string &probes_path, const std::string &cache_path, int k, int ntar,
                   std::vector<int32_t> &parent, ProbeSet &ps, bool *from_cache = nullptr, int threads = 0,
                   StartupTiming *timing = nullptr, std::thread *cache_writer = nullptr);

// Hashtable + Tree1 onto the GPU.  Returns false where the reference prints "out of memory in table"
// and exits with 1 (newkmer_10nx.cpp:256-260).
bool engine_open(Engine &e, const ProbeSet &ps, const std::vector<int32_t> &parent, int k, int log2_slots, int max_probes,
                 unsigned flags, int device);
// the same on several devices ("0,1,2,3"; a device may be named twice): the table is built once and replicated
bool engine_open(Engine &e, const ProbeSet &ps, const std::vector<int32_t> &parent, int k, int log2_slots, int max_probes,
                 unsigned flags, const std::vector<int> &devices);
std::vector<int> parse_devices(const std::string &list);
// newkmer_10nx.cpp:1017-1019 on every replica
void engine_reset(Engine &e);

// The input files of a run, parsed (+ trimmed) ahead of their turn on a small pool of reader threads:
// gzip inflate + parsing is the slow half of the program (~1 M reads/s per core) and files are
// independent until their reads reach the counters.  Batches are handed out strictly in file order;
// a reader that runs ahead blocks once its file has `depth` batches queued, so memory stays bounded.
// A failure while reading file i surfaces when the consumer gets to file i (the files before it have
// been processed completely, like in the sequential reference).
using SourceOpener = std::function<std::unique_ptr<ReadSource>()>;
class Prefetcher {
public:
    Prefetcher(std::vector<SourceOpener> files, int threads, size_t batch_reads, size_t batch_bases, size_t depth = 3);
    ~Prefetcher();
    // next batch of file `index` (indices must be visited in increasing order); nullptr at its end
    std::unique_ptr<ReadBatch> next(size_t index);
    // next batch of ANY of the files [lo, hi) -- whichever has one ready; `which` says whose.  nullptr when all of them
    // are at their end.  A file that failed throws its Fatal only once the files before it in the range are through
    // (the reference would have read those completely before it met the failure).
    std::unique_ptr<ReadBatch> next_any(size_t lo, size_t hi, size_t &which);
    SourceStats file_stats(size_t index); // of a file that is through
    double seconds_waited() const;        // the consumer, inside next() / next_any(): the host stages were the slower side
private:
    struct Impl;
    std::unique_ptr<Impl> impl_;
};

// classify every batch of file `index`; returns the number of reads handed to process_read
long long run_file(Engine &e, Prefetcher &pf, size_t index, ReadSaver &saver);
// the same for the files [first, first + count) of ONE sample, read and classified at the same time (the two mates of
// nk10: two inflate threads instead of one after the other); the counters do not care about the order, the read saver
// restores it.  handed[f]: reads of file first + f handed to process_read; done(f) is called when file first + f is
// through, in file order.
void run_files_together(Engine &e, Prefetcher &pf, size_t first, size_t count, ReadSaver &saver, std::vector<long long> &handed,
                        const std::function<void(size_t)> &done);

// --dry-run support (host stages only, no GPU): what WOULD be handed to the GPU, as text
void dry_dump_db(FILE *f, const std::vector<int32_t> &parent, const ProbeSet &ps);
void dry_dump_source(FILE *f, const std::string &label, ReadSource &src, size_t batch_reads, int k);

// gcount / ucount of the sample -> "<i>,<g>,<u>" lines
void finish_sample(Engine &e, const std::string &result_path);

} // namespace kidhost
