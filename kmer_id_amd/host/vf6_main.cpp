// kmer_read_vf6 -- command-line compatible replacement of the reference's generic reader
// (kmer_read_vf6.cpp, main at :968-1170):
//     kmer_read_vf6 -name DB -jname JOBS [-target T] [-fadir DIR]
// reads ./DB/DB_data.txt, ./DB/DB_tree.txt, ./DB/DB_probes.txt.gz and the job list
// ./JOBS/JOBS.txt ("<job> <nfiles>" followed by nfiles paths), classifies every file of a job
// (.fastq.gz / .fasta.gz / .fasta / .fastq) on the GPU and writes ./JOBS/<job>_result.txt,
// ./JOBS/<job>_reads.txt and, with -target, ./JOBS/<job>_target_reads.txt.
// Differences to nk10 that reach the kernel: U/u count as T (:496-500,521-525) and the number of
// targets comes from the data file (:1073-1084).
// Extra options: --k K (30) --log2-slots L (30) --device D (0) --batch-reads N
#include <stdlib.h>

#include <fstream>
#include <iostream>
#include <sstream>

#include "kid_driver.h"

using namespace kidhost;

static bool ends_with(const std::string &s, const std::string &suffix)
{
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

// suffix dispatch of the reference (:1133-1152): .fastq.gz, .fasta.gz, .fasta, .fastq; anything else is skipped
static std::unique_ptr<ReadSource> open_by_suffix(const std::string &name, int k, bool *missing_plain_fasta)
{
    if (ends_with(name, ".fastq.gz")) return std::unique_ptr<ReadSource>(new FastqStream(name, k));
    if (ends_with(name, ".fasta.gz")) return std::unique_ptr<ReadSource>(new FastaGzStream(name, k));
    if (ends_with(name, ".fasta")) {
        std::unique_ptr<PlainTokenStream> p(new PlainTokenStream(name, k, false));
        if (!p->present() && missing_plain_fasta) *missing_plain_fasta = true;
        return std::unique_ptr<ReadSource>(std::move(p));
    }
    if (ends_with(name, ".fastq")) return std::unique_ptr<ReadSource>(new PlainTokenStream(name, k, true));
    return nullptr;
}

int main(int argc, char **argv)
{
    std::string dname, wdir, jname, jdir, fdir;
    int save_target = 0, k = 30, log2_slots = 30, device = 0, threads = 4;
    std::string device_list;
    size_t batch_reads = 1 << 18;
    std::string dry_run; // --dry-run FILE: host stages only (no GPU), for the CPU test-suite
    std::string db_cache; // --db-cache FILE: binary cache of the parsed database
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : "";
        if (a == "-name") { dname = v; wdir = "./" + dname + "/"; }
        if (a == "-fadir") fdir = v; // only the dead alignment branch reads it
        if (a == "-jname") { jname = v; jdir = "./" + jname + "/"; }
        if (a == "-target") save_target = atoi(v);
        if (a == "--k") k = atoi(v);
        if (a == "--log2-slots") log2_slots = atoi(v);
        if (a == "--device") device = atoi(v);
        if (a == "--devices") device_list = v; // several GPUs: replicas of the table, batches dealt round-robin, counters merged
        if (a == "--batch-reads") batch_reads = (size_t)atoll(v);
        if (a == "--dry-run") dry_run = v;
        if (a == "--threads") threads = atoi(v);
        if (a == "--db-cache") db_cache = v;
    }
    const std::string iname = wdir + dname + "_data.txt", tname = wdir + dname + "_tree.txt",
                      pname = wdir + dname + "_probes.txt.gz", jfile = jdir + jname + ".txt";
    try {
        // ---- job list (:1021-1057)
        JobList jobs;
        if (load_job_list(jfile, jobs)) std::cout << jobs.runnable << " jobs" << std::endl;
        else std::cout << "narin " << jfile << std::endl;
        // ---- strain list: number of targets = largest target id + 1 (:1059-1089)
        int num_targ = 0, num_orgs = 0;
        {
            std::ifstream fin(iname);
            if (!fin) {
                std::cout << "narin " << iname << std::endl;
                std::cerr << "kmer_read_vf6: no strain list, the number of targets is unknown\n";
                return 3; // the reference goes on with a zero-sized tree and crashes
            }
            std::string line, acc;
            int targi = 0;
            while (std::getline(fin, line)) {
                if (!line.empty() && line.back() == '\r') line.pop_back();
                if (line.length() > 1) {
                    std::stringstream ls(line);
                    ls >> targi >> acc;
                    if (targi > num_targ) num_targ = targi;
                    num_orgs++;
                }
            }
            std::cout << num_orgs << " strains" << std::endl;
            std::cout << num_targ << " targs" << std::endl;
            num_targ++;
        }
        std::vector<int32_t> parent;
        ProbeSet ps;
        set_inflate_threads(threads >= 6 ? threads / 2 : 1); // (gzip inputs: pieces inflated side by side when there are threads for it)
        load_database(tname, pname, db_cache, k, num_targ, parent, ps);
        std::cout << "tree loaded" << std::endl;
        std::cout << ps.lines_parsed << " kmers loaded" << std::endl;

        if (!dry_run.empty()) {
            FILE *f = fopen(dry_run.c_str(), "w");
            if (!f) { perror("kmer_read_vf6"); return 2; }
            dry_dump_db(f, parent, ps);
            for (int j = 0; j < jobs.runnable; j++)
                for (int i = 0; i < jobs.n_inputs(j); i++) {
                    const std::string &input = jobs.file_rows[(size_t)j][(size_t)i];
                    std::unique_ptr<ReadSource> src = open_by_suffix(input, k, nullptr);
                    if (src) dry_dump_source(f, jobs.header_name[(size_t)j] + " " + input, *src, batch_reads, k);
                }
            fclose(f);
            return 0;
        }
        Engine eng;
        eng.batch_reads = batch_reads;
        const std::vector<int> devices = device_list.empty() ? std::vector<int>(1, device) : parse_devices(device_list);
        if (!engine_open(eng, ps, parent, k, log2_slots, 0, KID_FLAG_U_IS_T, devices)) {
            std::cout << "out of memory in table " << std::endl;
            return 1;
        }
        ps = ProbeSet();

        std::vector<SourceOpener> files;
        std::vector<std::string> names;
        for (int j = 0; j < jobs.runnable; j++)
            for (int i = 0; i < jobs.n_inputs(j); i++) names.push_back(jobs.file_rows[(size_t)j][(size_t)i]);
        std::vector<char> missing(names.size(), 0);
        for (size_t f = 0; f < names.size(); f++) {
            const std::string name = names[f];
            char *flag = &missing[f];
            files.push_back([name, k, flag]() {
                bool m = false;
                std::unique_ptr<ReadSource> src = open_by_suffix(name, k, &m);
                *flag = m ? 1 : 0;
                return src;
            });
        }
        Prefetcher pf(std::move(files), threads, eng.batch_reads, eng.batch_bases);
        size_t fi = 0;
        for (int j = 0; j < jobs.runnable; j++) { // :1116-1164
            const std::string jstr = jobs.header_name[(size_t)j];
            engine_reset(eng);
            const std::string base = "./" + jname + "/" + jstr;
            long long tct = 0;
            {
                ReadSaver saver(base + "_reads.txt", num_targ, save_target > 0 ? base + "_target_reads.txt" : "",
                                (uint32_t)(save_target > 0 ? save_target : 0), save_target == 0);
                for (int i = 0; i < jobs.n_inputs(j); i++, fi++) {
                    std::cout << names[fi] << std::endl;
                    tct += run_file(eng, pf, fi, saver);
                    if (missing[fi]) std::cout << "nark " << names[fi] << std::endl;
                }
            }
            std::cout << tct << " reads loaded" << std::endl;
            finish_sample(eng, base + "_result.txt");
        }
        leave_now(0);
    } catch (const Fatal &f) {
        std::cerr << f.message << "\n";
        return f.exit_code;
    }
    return 0;
}
