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
//   --batch-reads N (1048576)  --r1 SUFFIX (_R1_tr.fastq.gz)  --r2 SUFFIX (_R2_tr.fastq.gz)
//   --dry-run FILE   host stages only (no GPU): parse the DB text files and the FASTQ files,
//                    write what WOULD be handed to the GPU to FILE (used by the CPU test-suite)
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <iostream>
#include <memory>
#include <mutex>
#include <thread>

#include "kid_host.h"
#include "kmer_id_amd.h"

using namespace kidhost;

static void die_kid(int rc)
{
    std::cerr << "nk10: " << kid_strerror(rc) << ": " << kid_last_error() << "\n";
    exit(rc == KID_ERR_TABLE_FULL ? 1 : 3);
}

namespace {
struct Pipe { // bounded hand-off of parsed batches from the reader thread to the GPU thread
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::unique_ptr<ReadBatch>> q;
    bool done = false;
    bool failed = false;
    Fatal failure{0, ""};
};
}

// one FASTQ file: parse + trim on a reader thread, classify batch by batch on the caller's thread
static long long run_file(const std::string &path, int k, size_t batch_reads, kid_sample *sample, ReadSaver &saver)
{
    Pipe pipe;
    std::thread reader([&]() {
        try {
            FastqStream fq(path, k);
            for (;;) {
                std::unique_ptr<ReadBatch> b(new ReadBatch());
                bool more = fq.fill(*b, batch_reads);
                if (!more) break;
                std::unique_lock<std::mutex> lk(pipe.m);
                pipe.cv.wait(lk, [&] { return pipe.q.size() < 3; });
                pipe.q.push_back(std::move(b));
                pipe.cv.notify_all();
            }
            fq.close();
        } catch (const Fatal &f) {
            std::lock_guard<std::mutex> lk(pipe.m);
            pipe.failed = true;
            pipe.failure = f;
        }
        std::lock_guard<std::mutex> lk(pipe.m);
        pipe.done = true;
        pipe.cv.notify_all();
    });
    long long n = 0;
    std::vector<uint32_t> final_targ;
    int rc = KID_OK;
    for (;;) {
        std::unique_ptr<ReadBatch> b;
        {
            std::unique_lock<std::mutex> lk(pipe.m);
            pipe.cv.wait(lk, [&] { return !pipe.q.empty() || pipe.done; });
            if (pipe.q.empty()) break;
            b = std::move(pipe.q.front());
            pipe.q.pop_front();
            pipe.cv.notify_all();
        }
        if (rc != KID_OK) continue; // drain
        final_targ.resize(b->size());
        rc = kid_classify_batch(sample, b->bases.data(), b->offsets.data(), b->start.data(), b->stop.data(), b->size(),
                                final_targ.data());
        if (rc == KID_OK) {
            saver.add_batch(*b, final_targ);
            n += (long long)b->size();
        }
    }
    reader.join();
    if (rc != KID_OK) die_kid(rc);
    if (pipe.failed) throw pipe.failure; // reads parsed before the failure were processed, as in the reference
    return n;
}

int main(int argc, char **argv)
{
    std::string dname, db_dir = "./bact10/", e1 = "_R1_tr.fastq.gz", e2 = "_R2_tr.fastq.gz";
    int ntar = 5982, k = 30, log2_slots = 30, device = 0;
    size_t batch_reads = 1 << 20;
    std::string dry_run;
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
        else if (a == "--batch-reads") batch_reads = (size_t)atoll(val("--batch-reads"));
        else if (a == "--r1") e1 = val("--r1");
        else if (a == "--r2") e2 = val("--r2");
        else if (a == "--dry-run") dry_run = val("--dry-run");
        else if (dname.empty()) dname = a;
        else { std::cerr << "nk10: unexpected argument " << a << "\n"; return 2; }
    }
    if (dname.empty()) {
        std::cerr << "usage: nk10 /path-to-fastq-files/ [--db-dir ./bact10/] [--ntar 5982] [--k 30] [--log2-slots 30] [--device 0]\n";
        return 2;
    }
    if (!db_dir.empty() && db_dir.back() != '/') db_dir += "/";
    if (batch_reads < 1) batch_reads = 1;

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
        std::vector<int32_t> parent = load_tree(tpath, ntar);
        std::cout << "tree loaded" << std::endl;
        ProbeSet ps = load_probes_gz(pname, k);
        std::cout << ps.lines_parsed << " kmers loaded" << std::endl;

        if (!dry_run.empty()) { // host stages only
            FILE *f = fopen(dry_run.c_str(), "w");
            if (!f) { perror("nk10"); return 2; }
            fprintf(f, "PARENT %d\n", ntar);
            for (int i = 0; i < ntar; i++) if (parent[(size_t)i] != 1) fprintf(f, "%d %d\n", i, parent[(size_t)i]);
            fprintf(f, "PROBES %zu %lld\n", ps.keys.size(), ps.lines_parsed);
            for (size_t i = 0; i < ps.keys.size(); i++) fprintf(f, "%llu %u\n", (unsigned long long)ps.keys[i], ps.targets[i]);
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
                    ReadBatch b;
                    fprintf(f, "FILE %s%s\n", prefix.c_str(), suffix.c_str());
                    while (fq.fill(b, batch_reads))
                        for (size_t r = 0; r < b.size(); r++)
                            fprintf(f, "%s\t%d\t%d\t%llu\n", b.acc[r].c_str(), b.start[r], b.stop[r],
                                    (unsigned long long)(b.offsets[r + 1] - b.offsets[r]));
                    fq.close();
                }
            fclose(f);
            return 0;
        }

        kid_db *db = nullptr;
        int rc = kid_db_build(ps.keys.data(), ps.targets.data(), ps.keys.size(), parent.data(), ntar, k, log2_slots, 0, 0, device, &db);
        if (rc == KID_ERR_TABLE_FULL) { std::cout << "out of memory in table " << std::endl; return 1; } // :256-260
        if (rc != KID_OK) die_kid(rc);
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

        kid_sample *sample = nullptr;
        rc = kid_sample_begin(db, &sample);
        if (rc != KID_OK) die_kid(rc);
        std::vector<int64_t> gcount((size_t)ntar), ucount((size_t)ntar);
        for (const std::string &prefix : fnames) { // :1015-1045
            rc = kid_sample_reset(sample);
            if (rc != KID_OK) die_kid(rc);
            std::cout << prefix << std::endl;
            long long tct = 0;
            {
                ReadSaver saver(dname + prefix + "_reads.txt", ntar);
                tct += run_file(dname + prefix + e1, k, batch_reads, sample, saver);
                std::cout << tct << " reads loaded" << std::endl;
                tct += run_file(dname + prefix + e2, k, batch_reads, sample, saver);
                std::cout << tct << " reads loaded" << std::endl;
            }
            rc = kid_sample_end(sample, gcount.data(), ucount.data());
            if (rc != KID_OK) die_kid(rc);
            write_result(dname + prefix + "_result.txt", gcount, ucount);
        }
        kid_sample_destroy(sample);
        kid_db_destroy(db);
    } catch (const Fatal &f) {
        std::cerr << f.message << "\n";
        return f.exit_code;
    }
    return 0;
}
