// nk10 -- command-line compatible replacement of the reference program newkmer_10nx.cpp:
//     ./nk10 /path-to-fastq-files/
// Same inputs (./bact10/{bData10.txt,btree_10.txt,probes10.txt.gz} relative to the working
// directory, <prefix>_R1_tr.fastq.gz / <prefix>_R2_tr.fastq.gz in the given directory), same
// progress lines on stdout, same <prefix>_result.txt / <prefix>_reads.txt files, same exit codes.
// The per-read work (process_read, newkmer_10nx.cpp:452-617) runs on an MI355X through
// libkmer_id_amd.so; this file is the driver (main, :915-1054).
//
// Options after the directory (all optional, defaults = the reference's compile-time constants):
//   --db-dir DIR (./bact10/)  --ntar N (5982)  --k K (30)  --log2-slots L (30)  --device D (0)
//   --devices A,B,...  several GPUs: the table is built on the first and replicated into the others' HBM, the batches of
//                    a sample are dealt round-robin over them, the counters merged when the sample is closed
//   --batch-reads N (262144)  --threads T (the host's, at most 16: inflating, finding lines, parsing)  --r1 SUFFIX (_R1_tr.fastq.gz)  --r2 SUFFIX (_R2_tr.fastq.gz)
//   --fasta          the reference's compile-time FASTQ=0 mode (:28,:1032-1035): one plain FASTA file
//                    <prefix><r1 suffix> per sample, read by process_fa (:877-913), no R2 file
//   --db-cache FILE  binary cache of the parsed database: read if valid, (re)written otherwise
//   --samples-in-flight N (threads / 2)  samples read and classified at the same time, each with its own counters on the GPU
//   --timing         one JSON line on stderr when the run ends: seconds of the start-up phases (probes inflate / parse,
//                    cache read / write, upload + table build on the GPU, first batch classified) and of the read files
//   --dry-run FILE   host stages only (no GPU): parse the DB text files and the FASTQ files,
//                    write what WOULD be handed to the GPU to FILE (used by the CPU test-suite)
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

#include "kid_driver.h"

using namespace kidhost;

int main(int argc, char **argv)
{
    std::string dname, db_dir = "./bact10/", e1 = "_R1_tr.fastq.gz", e2 = "_R2_tr.fastq.gz";
    int ntar = 5982, k = 30, log2_slots = 30, device = 0;
    int threads = (int)std::thread::hardware_concurrency(); // (at most 16 unless --threads says otherwise)
    threads = threads < 1 ? 1 : threads > 16 ? 16 : threads;
    size_t batch_reads = 1 << 18;
    std::string dry_run, db_cache, device_list;
    bool fasta_mode = false;
    bool parse_only = false; // --parse-only: run the reader pool over the directory without a GPU and report its rate
    bool timing = false;
    int in_flight = 0; // --samples-in-flight N (0: half the reader threads)
    const auto t_start = std::chrono::steady_clock::now();
    auto since_start = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) { std::cerr << "nk10: " << name << " needs a value\n"; exit(2); }
            return argv[++i];
        };
        if (a == "--db-dir") db_dir = val("--db-dir");
        else if (a == "--ntar") ntar = atoi(val("--ntar"));
        else if (a == "--k") k = atoi(val("--k"));
        else if (a == "--log2-slots") log2_slots = atoi(val("--log2-slots"));
        else if (a == "--device") device = atoi(val("--device"));
        else if (a == "--devices") device_list = val("--devices");
        else if (a == "--batch-reads") batch_reads = (size_t)atoll(val("--batch-reads"));
        else if (a == "--threads") threads = atoi(val("--threads"));
        else if (a == "--r1") e1 = val("--r1");
        else if (a == "--r2") e2 = val("--r2");
        else if (a == "--dry-run") dry_run = val("--dry-run");
        else if (a == "--db-cache") db_cache = val("--db-cache");
        else if (a == "--parse-only") parse_only = true;
        else if (a == "--fasta") fasta_mode = true;
        else if (a == "--timing") timing = true;
        else if (a == "--samples-in-flight") in_flight = atoi(val("--samples-in-flight"));
        else if (dname.empty()) dname = a;
        else { std::cerr << "nk10: unexpected argument " << a << "\n"; return 2; }
    }
    if (dname.empty()) {
        std::cerr << "usage: nk10 /path-to-fastq-files/ [--db-dir ./bact10/] [--ntar 5982] [--k 30] [--log2-slots 30] [--device 0]\n";
        return 2;
    }
    if (!db_dir.empty() && db_dir.back() != '/') db_dir += "/";
    if (batch_reads < 1) batch_reads = 1;
    if (threads < 1) threads = 1;
    // the probes file: half of the threads inflate pieces of it side by side, the others parse (fewer than three: one inflates it all)
    set_inflate_threads(threads / 2 >= 3 ? threads / 2 : 1);
    if (in_flight <= 0) in_flight = threads / 2 > 0 ? threads / 2 : 1; // (a sample = two files = two inflate threads)

    try {
        // ---- load the database once (:949-989)
        const std::string iname = db_dir + "bData10.txt", tname = db_dir + "btree_10.txt", pname = db_dir + "probes10.txt.gz";
        if (!strain_list_present(iname)) std::cout << "narin " << iname << std::endl;
        std::string tpath = tname;
        {   // README.md says btree10.txt, the code says btree_10.txt: accept both, the code's name first
            FILE *f = fopen(tpath.c_str(), "r");
            if (f) fclose(f);
            else {
                std::string alt = db_dir + "btree10.txt";
                FILE *g = fopen(alt.c_str(), "r");
                if (g) { fclose(g); tpath = alt; }
            }
        }
        std::vector<int32_t> parent;
        ProbeSet ps;
        StartupTiming tm;
        bool from_cache = false;
        std::thread cache_writer; // (a cache that has to be written is written beside the upload and the table build)
        struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } join_cache_writer{cache_writer};
        // the HIP runtime takes 0.1-0.15 s to come up: while the database is being read, not after
        std::thread gpu_warm;
        Joiner join_gpu_warm{gpu_warm};
        if (dry_run.empty() && !parse_only) {
            const int warm_device = device_list.empty() ? device : atoi(device_list.c_str());
            gpu_warm = std::thread([warm_device] {
                void *p = nullptr;
                if (kid_host_alloc(warm_device, 4096, &p) == KID_OK) kid_host_free(p);
            });
        }
        load_database(tpath, pname, db_cache, k, ntar, parent, ps, &from_cache, threads, &tm, &cache_writer);
        const double t_db_loaded = since_start();
        double t_gpu_ready = -1, t_first_file = -1;
        size_t n_entries = ps.keys.size(), n_devices = 0;
        std::string files_json; // per read file: seconds of its host stages
        std::string samples_json; // per sample: when it began, when its files were through, when its outputs were written
        double host_wait_s = 0, gpu_wait_s = 0, submit_s = 0;
        auto print_timing = [&]() {
            if (!timing) return;
            fprintf(stderr, "{\"nk10_timing\": {\"entries\": %zu, \"from_cache\": %s, \"probes_text_bytes\": %llu, \"probes_inflate_s\": %.3f, "
                            "\"probes_parse_wall_s\": %.3f, \"parse_threads\": %d, \"cache_read_s\": %.3f, \"cache_write_s\": %.3f, "
                            "\"db_loaded_at_s\": %.3f, \"gpu_upload_and_build_s\": %.3f, \"gpu_ready_at_s\": %.3f, "
                            "\"first_file_classified_at_s\": %.3f, \"total_s\": %.3f, \"log2_slots\": %d, \"devices\": %zu, "
                            "\"consumer_waited_for_host_stages_s\": %.3f, \"consumer_waited_for_gpu_s\": %.3f, \"consumer_submit_s\": %.3f, "
                            "\"reader_threads\": %d, \"samples_in_flight\": %d, \"files\": [%s], \"samples\": [%s]}}\n",
                    n_entries, from_cache ? "true" : "false", (unsigned long long)tm.text_bytes, tm.inflate_s, tm.parse_wall_s,
                    tm.parse_threads, tm.cache_read_s, tm.cache_write_s, t_db_loaded, tm.gpu_build_s, t_gpu_ready, t_first_file,
                    since_start(), log2_slots, n_devices, host_wait_s, gpu_wait_s, submit_s, threads, in_flight, files_json.c_str(), samples_json.c_str());
        };
        std::cout << "tree loaded" << std::endl;
        std::cout << ps.lines_parsed << " kmers loaded" << std::endl;

        if (!dry_run.empty()) { // host stages only
            FILE *f = fopen(dry_run.c_str(), "w");
            if (!f) { perror("nk10"); return 2; }
            dry_dump_db(f, parent, ps);
            DIR *dd = opendir(dname.c_str());
            std::vector<std::string> names;
            if (dd) {
                while (struct dirent *ent = readdir(dd)) {
                    std::string n1 = ent->d_name;
                    size_t pos = n1.find(e1);
                    if (pos != std::string::npos) names.push_back(n1.substr(0, pos));
                }
                closedir(dd);
            }
            for (const std::string &prefix : names)
                for (const std::string &suffix : {e1, e2}) {
                    FastqStream fq(dname + prefix + suffix, k);
                    dry_dump_source(f, prefix + suffix, fq, batch_reads, k);
                }
            fclose(f);
            return 0;
        }

        if (parse_only) {
            DIR *dd = opendir(dname.c_str());
            std::vector<SourceOpener> files;
            if (dd) {
                while (struct dirent *ent = readdir(dd)) {
                    std::string n1 = ent->d_name;
                    size_t pos = n1.find(e1);
                    if (pos == std::string::npos) continue;
                    for (const std::string &suffix : {e1, e2}) {
                        const std::string path = dname + n1.substr(0, pos) + suffix;
                        files.push_back([path, k]() { return std::unique_ptr<ReadSource>(new FastqStream(path, k)); });
                    }
                }
                closedir(dd);
            }
            const size_t nf = files.size();
            Prefetcher pf(std::move(files), threads, batch_reads, (size_t)256 << 20);
            long long n = 0, bases = 0;
            for (size_t f = 0; f < nf; f++)
                while (std::unique_ptr<ReadBatch> b = pf.next(f)) { n += (long long)b->size(); bases += (long long)b->bases.size(); }
            std::cout << n << " reads, " << bases << " bases parsed" << std::endl;
            if (cache_writer.joinable()) cache_writer.join();
            print_timing();
            return 0;
        }
        Engine eng;
        eng.batch_reads = batch_reads;
        const std::vector<int> devices = device_list.empty() ? std::vector<int>(1, device) : parse_devices(device_list);
        if (gpu_warm.joinable()) gpu_warm.join();
        if (!engine_open(eng, ps, parent, k, log2_slots, 0, 0, devices)) { // :256-260
            std::cout << "out of memory in table " << std::endl;
            return 1;
        }
        // file text goes into page-locked memory from here on: uploads by DMA, not through a CPU copy
        static int pin_device = devices[0];
        set_text_allocator([](size_t n) -> void * { void *p = nullptr; return kid_host_alloc(pin_device, n, &p) == KID_OK ? p : nullptr; },
                           [](void *p) { kid_host_free(p); });
        t_gpu_ready = since_start();
        tm.gpu_build_s = t_gpu_ready - t_db_loaded;
        n_devices = devices.size();
        if (cache_writer.joinable()) cache_writer.join();
        // The probe arrays (1.7 GB of address space) go back when the first sample is through and another follows: unmapping them takes 0.1 s
        // during which no other thread of the process can map or page-lock memory -- which is what the first blocks of
        // the first sample need.  (A run that ends before that leaves them to the operating system.)
        struct Release { std::mutex m; std::condition_variable cv; bool now = false; };
        static Release &release = *new Release(); // (never destroyed: the waiting thread may outlive main)
        std::thread([old = std::move(ps)]() mutable {
            std::unique_lock<std::mutex> lk(release.m);
            release.cv.wait(lk, [] { return release.now; });
            lk.unlock();
            old = ProbeSet();
        }).detach();
        ps = ProbeSet();

        // ---- find the samples (:992-1014): every directory entry whose name contains the R1 suffix
        std::cout << dname << std::endl;
        std::vector<std::string> fnames;
        DIR *dir = opendir(dname.c_str());
        if (!dir) {
            std::cout << "hosed" << std::endl;
            perror("");
            return EXIT_FAILURE;
        }
        while (struct dirent *ent = readdir(dir)) {
            std::string name1 = ent->d_name;
            size_t pos = name1.find(e1);
            if (pos != std::string::npos) fnames.push_back(name1.substr(0, pos));
        }
        closedir(dir);

        std::vector<SourceOpener> files;
        std::vector<char> missing(fnames.size(), 0);
        for (size_t f = 0; f < fnames.size(); f++) {
            const std::string prefix = fnames[f];
            if (fasta_mode) {
                const std::string path = dname + prefix + e1;
                char *flag = &missing[f];
                files.push_back([path, k, flag]() {
                    std::unique_ptr<PlainTokenStream> p(new PlainTokenStream(path, k, false, /*strip_cr=*/false));
                    *flag = p->present() ? 0 : 1;
                    return std::unique_ptr<ReadSource>(std::move(p));
                });
                continue;
            }
            for (const std::string &suffix : {e1, e2}) {
                const std::string path = dname + prefix + suffix;
                files.push_back([path, k]() { return std::unique_ptr<ReadSource>(new FastqStream(path, k)); });
            }
        }
        {   // the threads are shared out over the files that are read at the same time (before the first file is opened)
            const size_t fps = fasta_mode ? 1 : 2;
            size_t at_once = in_flight < 1 ? 1 : (size_t)in_flight;
            if (at_once > fnames.size()) at_once = fnames.empty() ? 1 : fnames.size();
            const size_t per_file = (size_t)threads / (fps * at_once);
            set_inflate_threads(per_file >= 3 ? (int)per_file : 1);
        }
        Prefetcher pf(std::move(files), threads, eng.batch_reads, eng.batch_bases);
        // ---- the samples (:1015-1045).  Up to `in_flight` of them are read and classified at the same time, each by a
        // thread with its own counters on the GPU(s): the host's work per file (inflate, find the lines) is what a sample
        // waits for, so a directory of samples is as fast as the cores it may use.  What a sample prints is kept and
        // printed in directory order; a sample that fails ends the run where the reference would have ended it (the
        // samples before it complete, what later ones wrote is removed).
        const size_t n_samples = fnames.size(), files_per_sample = fasta_mode ? 1 : 2;
        size_t n_workers = in_flight < 1 ? 1 : (size_t)in_flight;
        if (n_workers > n_samples) n_workers = n_samples ? n_samples : 1;
        struct SampleOut {
            std::string text;
            bool done = false, failed = false, wrote_result = false;
            Fatal failure{0, ""};
        };
        std::vector<SampleOut> outs(n_samples);
        std::mutex om;
        std::condition_variable ocv;
        size_t next_sample = 0;
        bool abort_run = false;
        std::vector<std::unique_ptr<Engine>> worker_engines;
        for (size_t w = 1; w < n_workers; w++) worker_engines.push_back(engine_worker(eng));
        auto process = [&](Engine &e, size_t f, std::string &out) {
            const std::string &prefix = fnames[f];
            const size_t fi0 = f * files_per_sample;
            const double t_begin = since_start();
            double t_file_done[2] = {-1, -1};
            engine_reset(e);
            out += prefix + "\n";
            long long tct = 0;
            {
                ReadSaver saver(dname + prefix + "_reads.txt", ntar);
                if (fasta_mode) {
                    tct += run_file(e, pf, fi0, saver);
                    if (missing[f]) out += "nark " + dname + prefix + e1 + "\n";
                    out += std::to_string(tct) + " reads loaded\n";
                } else {
                    // the two mates are inflated, indexed and classified at the same time; "<tct> reads loaded" (:1030,:1036)
                    // comes when a file is through, R1 first
                    std::vector<long long> handed;
                    run_files_together(e, pf, fi0, 2, saver, handed, [&](size_t mate) {
                        tct += handed[mate];
                        std::lock_guard<std::mutex> lk(om);
                        if (t_first_file < 0) t_first_file = since_start();
                        t_file_done[mate] = since_start();
                        out += std::to_string(tct) + " reads loaded\n";
                    });
                }
            }
            const double t_reads_written = since_start();
            finish_sample(e, dname + prefix + "_result.txt");
            if (timing) {
                char buf[256];
                snprintf(buf, sizeof(buf), "{\"sample\": %zu, \"begin_s\": %.3f, \"r1_through_s\": %.3f, \"r2_through_s\": %.3f, \"reads_txt_written_s\": %.3f, "
                                           "\"result_written_s\": %.3f}", f, t_begin, t_file_done[0], t_file_done[1], t_reads_written, since_start());
                std::lock_guard<std::mutex> lk(om);
                samples_json += (samples_json.empty() ? "" : ", ") + std::string(buf);
            }
        };
        auto worker = [&](Engine *e) {
            for (;;) {
                size_t f;
                {
                    std::lock_guard<std::mutex> lk(om);
                    if (abort_run || next_sample >= n_samples) return;
                    f = next_sample++;
                }
                std::string text;
                bool failed = false;
                Fatal failure{0, ""};
                try { process(*e, f, text); }
                catch (const Fatal &x) { failed = true; failure = x; }
                std::lock_guard<std::mutex> lk(om);
                outs[f].text = std::move(text);
                outs[f].failed = failed;
                outs[f].failure = failure;
                outs[f].wrote_result = !failed;
                outs[f].done = true;
                if (failed) abort_run = true;
                ocv.notify_all();
                if (next_sample < n_samples) { // (more samples to come: now is the time; a run that is about to end leaves it to the exit)
                    std::lock_guard<std::mutex> rl(release.m);
                    release.now = true;
                    release.cv.notify_all();
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t w = 1; w < n_workers; w++) pool.emplace_back(worker, worker_engines[w - 1].get());
        std::thread first(worker, &eng);
        int exit_code = 0;
        std::string fail_message;
        for (size_t f = 0; f < n_samples; f++) { // print in directory order
            std::unique_lock<std::mutex> lk(om);
            ocv.wait(lk, [&] { return outs[f].done || (abort_run && f >= next_sample); });
            if (!outs[f].done) break; // never started: a sample before it failed
            std::cout << outs[f].text << std::flush;
            if (outs[f].failed) { exit_code = outs[f].failure.exit_code; fail_message = outs[f].failure.message; break; }
        }
        first.join();
        for (std::thread &t : pool) t.join();
        if (exit_code) {
            // samples behind the failed one that were already through: the reference never got to them
            bool behind = false;
            for (size_t f = 0; f < n_samples; f++) {
                if (behind && outs[f].done) {
                    remove((dname + fnames[f] + "_result.txt").c_str());
                    remove((dname + fnames[f] + "_reads.txt").c_str());
                }
                if (outs[f].failed) behind = true;
            }
            throw Fatal{exit_code, fail_message};
        }
        const size_t fi = n_samples * files_per_sample;
        for (const std::unique_ptr<Engine> &we : worker_engines) { eng.gpu_wait_s += we->gpu_wait_s; eng.submit_s += we->submit_s; }
        if (timing) {
            for (size_t i = 0; i < fi; i++) {
                const SourceStats st = pf.file_stats(i);
                char buf[256];
                snprintf(buf, sizeof(buf), "%s{\"file\": %zu, \"text_bytes\": %llu, \"inflate_s\": %.3f, \"index_s\": %.3f}", i ? ", " : "", i,
                         (unsigned long long)st.text_bytes, st.inflate_s, st.index_s);
                files_json += buf;
            }
            host_wait_s = pf.seconds_waited();
            gpu_wait_s = eng.gpu_wait_s;
            submit_s = eng.submit_s;
        }
        print_timing();
        leave_now(0);
    } catch (const Fatal &f) {
        std::cerr << f.message << "\n";
        return f.exit_code;
    }
    return 0;
}
