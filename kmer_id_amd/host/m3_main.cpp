// kmer_read_m3 -- command-line compatible replacement of the reference's mitochondria reader, the
// back end of the Galaxy tool (kmer_read_m3.cpp, main at :973-1131):
//     kmer_read_m3 -wdir DIR/ -f1 reads1 [-f2 reads2|none]
// reads DIR/mitochondria_{data.txt,tree.txt,probes.txt.gz}, classifies -f1 and -f2
// (.fastq.gz / .fasta / .fastq / .fasta.gz) and writes DIR/result.txt.
// What reaches the kernel: lookups give up after MAXREPROBE = 16 probes (:42,:232), so the table is
// built sequentially on the host with the reference's exact cell geometry (results depend on it).
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

// suffix order of the reference (:1083-1100): .fastq.gz, .fasta, .fastq, .fasta.gz
static std::unique_ptr<ReadSource> open_by_suffix(const std::string &name, int k, bool *missing_plain_fasta)
{
    if (ends_with(name, ".fastq.gz")) return std::unique_ptr<ReadSource>(new FastqStream(name, k));
    if (ends_with(name, ".fasta")) {
        std::unique_ptr<PlainTokenStream> p(new PlainTokenStream(name, k, false));
        if (!p->present() && missing_plain_fasta) *missing_plain_fasta = true;
        return std::unique_ptr<ReadSource>(std::move(p));
    }
    if (ends_with(name, ".fastq")) return std::unique_ptr<ReadSource>(new PlainTokenStream(name, k, true));
    if (ends_with(name, ".fasta.gz")) return std::unique_ptr<ReadSource>(new FastaGzStream(name, k));
    return nullptr;
}

int main(int argc, char **argv)
{
    std::string wdir, r1name, r2name;
    int k = 30, log2_slots = 30, device = 0, threads = 2;
    std::string device_list;
    size_t batch_reads = 1 << 18;
    std::string dry_run; // --dry-run FILE: host stages only (no GPU), for the CPU test-suite
    std::string db_cache; // --db-cache FILE: binary cache of the parsed database
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : "";
        if (a == "-wdir") wdir = v;
        if (a == "-f1") r1name = v;
        if (a == "-f2") r2name = v;
        if (a == "--k") k = atoi(v);
        if (a == "--log2-slots") log2_slots = atoi(v);
        if (a == "--device") device = atoi(v);
        if (a == "--devices") device_list = v; // several GPUs: replicas of the table, batches dealt round-robin, counters merged
        if (a == "--batch-reads") batch_reads = (size_t)atoll(v);
        if (a == "--dry-run") dry_run = v;
        if (a == "--threads") threads = atoi(v);
        if (a == "--db-cache") db_cache = v;
    }
    const std::string iname = wdir + "mitochondria_data.txt", tname = wdir + "mitochondria_tree.txt",
                      pname = wdir + "mitochondria_probes.txt.gz";
    try {
        std::cout << "r1 " << r1name << std::endl;
        std::cout << "r2 " << r2name << std::endl;
        std::cout << "wd " << wdir << std::endl;
        int num_targ = 0, num_orgs = 0; // (num_targ is uninitialised in the reference, :981; 0 in practice)
        {
            std::ifstream fin(iname);
            if (!fin) {
                std::cerr << "kmer_read_m3: cannot read " << iname << "\n";
                return 3; // the reference builds a zero-sized tree here and crashes later
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
            num_targ++;
        }
        {
            std::ifstream fin(tname);
            if (!fin) return 1; // `else exit(1)`, :1060
        }
        std::vector<int32_t> parent;
        ProbeSet ps;
        set_inflate_threads(threads >= 6 ? threads / 2 : 1); // (gzip inputs: pieces inflated side by side when there are threads for it)
        load_database(tname, pname, db_cache, k, num_targ, parent, ps);
        std::cout << "tree loaded" << std::endl;
        std::cout << ps.lines_parsed << " kmers loaded" << std::endl;
        if (ps.lines_parsed < 2) return 1; // :1067

        if (!dry_run.empty()) {
            FILE *f = fopen(dry_run.c_str(), "w");
            if (!f) { perror("kmer_read_m3"); return 2; }
            dry_dump_db(f, parent, ps);
            for (const std::string &name : {r1name, r2name}) {
                if (name.empty() || name == "none") continue;
                std::unique_ptr<ReadSource> src = open_by_suffix(name, k, nullptr);
                if (src) dry_dump_source(f, name, *src, batch_reads, k);
            }
            fclose(f);
            return 0;
        }
        Engine eng;
        eng.batch_reads = batch_reads;
        const std::vector<int> devices = device_list.empty() ? std::vector<int>(1, device) : parse_devices(device_list);
        if (!engine_open(eng, ps, parent, k, log2_slots, /*MAXREPROBE*/ 16, 0, devices)) {
            std::cout << "out of memory in table " << std::endl;
            return 1;
        }
        ps = ProbeSet();

        if (r1name.empty()) throw Fatal{134, "no -f1 given (std::out_of_range in the reference, :1080)"};
        std::cout << r1name.length() << " : " << r1name[r1name.length() - 1] << std::endl;
        const bool have2 = r2name.length() > 1 && r2name != "none";
        std::vector<std::string> names{r1name};
        if (have2) names.push_back(r2name);
        std::vector<char> missing(names.size(), 0);
        std::vector<SourceOpener> files;
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
        ReadSaver saver("", num_targ); // the reads file is commented out in this program (:612-621)
        auto is_fagz = [](const std::string &n) { return ends_with(n, ".fasta.gz"); };
        if (is_fagz(r1name)) std::cout << "true" << std::endl; // process_fagz, :789
        long long tct = run_file(eng, pf, 0, saver);
        if (missing[0]) std::cout << "nark " << r1name << std::endl;
        std::cout << tct << " reads loaded" << std::endl;
        if (have2) {
            if (is_fagz(r2name)) std::cout << "true" << std::endl;
            tct += run_file(eng, pf, 1, saver);
            if (missing[1]) std::cout << "nark " << r2name << std::endl;
            if (ends_with(r2name, ".fastq.gz")) std::cout << tct << " reads loaded" << std::endl; // printed inside that branch too (:1107)
            std::cout << tct << " reads loaded" << std::endl;
        }
        finish_sample(eng, wdir + "result.txt");
        leave_now(0);
    } catch (const Fatal &f) {
        std::cerr << f.message << "\n";
        return f.exit_code;
    }
    return 0;
}
