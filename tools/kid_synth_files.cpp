// kid_synth_files -- writes the synthetic workload (DESIGN.md "synthetic workload") as the FILES the reference's
// programs read: a probes file in the format of kmer_build_vf6.cpp:625 ("SEQ,target,org,position,strand,count", gzip)
// and paired FASTQ.gz files named like newkmer_10nx.cpp:29-30 wants them.  The real probes10.txt.gz is not
// distributed with the reference (README.md:12); this is how a full-scale stand-in (108 585 519 lines, ~5.4 GB of
// text) is made in seconds: the generators are the library's own (kid_common.h: kid_synth_db_key, kid_synth_read),
// blocks of lines are deflated on all cores and written the way pigz does it: ONE gzip member, every block's deflate
// data ended on a byte boundary (Z_SYNC_FLUSH) and appended, the last one finished, the CRC-32s combined -- the kind of
// file `gzip` itself writes, which zlib's gzread (what the reference uses, newkmer_10nx.cpp:673) reads as one stream.
// (--members: every block a gzip member of its own, the way bgzip-like writers do it.)
//
//   kid_synth_files probes --counts FILE --out FILE.gz [--scale S] [--k 30] [--seed N] [--threads T] [--level L]
//   kid_synth_files fastq  --counts FILE --tree FILE --out-dir DIR/ --samples S --pairs P [--read-len 150]
//                          [--qual mixed|high] [--r0 N] [--prefix S] [--threads T] [--level L] [--scale S] [--k 30]
//   both: [--members]
// FILE formats: --counts: one line per target "target,kmers" (what b10/refkey10.txt holds in columns 1 and 3);
//               --tree: "parent child" per line (b10/btree_10.txt).
// FASTQ records have a fixed width ("@r%09llu/%d"), so that a test can map a file back to arrays without parsing.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../kmer_id_amd/csrc/kid_common.h"

static const uint64_t DB_SEED = 0xB10, READ_SEED = 0x5EED, QUAL_SEED = 0x9A1; // kmer_id_amd/synth.py

[[noreturn]] static void die(const std::string &msg)
{
    fprintf(stderr, "kid_synth_files: %s\n", msg.c_str());
    exit(2);
}

// ---------------------------------------------------------------- ordered, parallel gzip writer
// make(c, text) fills the text of chunk c; workers deflate it and append the pieces to the file in chunk order.
static bool g_members = false;
static void write_gz_chunks(const std::string &path, uint64_t n_chunks, int threads, int level,
                            const std::function<void(uint64_t, std::string &)> &make)
{
    if (n_chunks == 0) n_chunks = 1; // (an empty text still makes a valid gzip member)
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) die("cannot create " + path);
    static const unsigned char header[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3};
    if (!g_members && fwrite(header, 1, 10, f) != 10) die("write error on " + path);
    std::atomic<uint64_t> next{0};
    std::mutex m;
    std::condition_variable cv;
    uint64_t next_write = 0, total_len = 0;
    uLong total_crc = crc32(0, nullptr, 0);
    bool failed = false;
    auto worker = [&]() {
        std::string text;
        std::vector<unsigned char> out;
        for (;;) {
            const uint64_t c = next.fetch_add(1);
            if (c >= n_chunks) return;
            text.clear();
            make(c, text);
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, level, Z_DEFLATED, g_members ? 15 + 16 : -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) die("deflateInit2 failed");
            out.resize(deflateBound(&zs, (uLong)text.size()) + 64);
            zs.next_in = (Bytef *)text.data();
            zs.avail_in = (uInt)text.size();
            zs.next_out = out.data();
            zs.avail_out = (uInt)out.size();
            const bool last = g_members || c + 1 == n_chunks;
            const int rc = deflate(&zs, last ? Z_FINISH : Z_SYNC_FLUSH);
            if ((last && rc != Z_STREAM_END) || (!last && (rc != Z_OK || zs.avail_in != 0 || zs.avail_out == 0))) die("deflate failed");
            const size_t n_out = out.size() - zs.avail_out;
            deflateEnd(&zs);
            const uLong piece_crc = g_members ? 0 : crc32(crc32(0, nullptr, 0), (const Bytef *)text.data(), (uInt)text.size());
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return next_write == c; });
            if (fwrite(out.data(), 1, n_out, f) != n_out) failed = true;
            if (!g_members) {
                total_crc = crc32_combine(total_crc, piece_crc, (z_off_t)text.size());
                total_len += text.size();
            }
            next_write++;
            cv.notify_all();
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++) pool.emplace_back(worker);
    for (std::thread &t : pool) t.join();
    if (!g_members) {
        unsigned char tr[8];
        for (int i = 0; i < 4; i++) {
            tr[i] = (unsigned char)(total_crc >> (8 * i));
            tr[4 + i] = (unsigned char)(total_len >> (8 * i));
        }
        if (fwrite(tr, 1, 8, f) != 8) failed = true;
    }
    if (fclose(f) != 0 || failed) die("write error on " + path);
}

// ---------------------------------------------------------------- inputs
static std::vector<uint64_t> load_counts(const std::string &path, double scale)
{
    FILE *f = fopen(path.c_str(), "r");
    if (!f) die("cannot open " + path);
    std::vector<uint64_t> cnt;
    char line[256];
    while (fgets(line, sizeof(line), f)) {
        unsigned long long t = 0, c = 0;
        if (sscanf(line, "%llu,%llu", &t, &c) != 2) continue;
        if (t >= cnt.size()) cnt.resize(t + 1, 0);
        cnt[t] = c;
    }
    fclose(f);
    if (scale < 1.0) // kmer_id_amd/synth.py scaled_counts: every target that has k-mers keeps at least one
        for (uint64_t &c : cnt) {
            const uint64_t s = (uint64_t)((double)c * scale);
            c = (c > 0 && s == 0) ? 1 : s;
        }
    return cnt;
}

static std::vector<int32_t> load_parent(const std::string &path, size_t ntar)
{
    std::vector<int32_t> parent(ntar, 1);
    FILE *f = fopen(path.c_str(), "r");
    if (!f) die("cannot open " + path);
    long long x, y;
    while (fscanf(f, "%lld %lld", &x, &y) == 2)
        if (y >= 0 && (size_t)y < ntar) parent[(size_t)y] = (int32_t)x;
    fclose(f);
    return parent;
}

static std::vector<uint64_t> cumulative(const std::vector<uint64_t> &cnt)
{
    std::vector<uint64_t> cum(cnt.size() + 1, 0);
    for (size_t i = 0; i < cnt.size(); i++) cum[i + 1] = cum[i] + cnt[i];
    return cum;
}

// ---------------------------------------------------------------- qualities (the model of kmer_id_amd/synth.py qualities())
// mostly 'I'; ~30 % of the reads have a decaying tail, ~10 % a poor head, ~5 % are noisy throughout, ~2 % are bad
// everywhere (process_qual drops those)
static void make_quals(uint64_t r, uint32_t len, bool mixed, char *q)
{
    memset(q, 'I', len);
    if (!mixed) return;
    const uint64_t d = kid_splitmix64(QUAL_SEED ^ (r * 0xA24BAED4963EE407ULL));
    const uint32_t kind = (uint32_t)(d & 0xFF);
    const uint32_t half = len / 2 > 0 ? len / 2 : 1, quarter = len / 4 > 0 ? len / 4 : 1;
    const uint32_t tail = (uint32_t)((d >> 8) % half), head = (uint32_t)((d >> 24) % quarter);
    auto noise = [&](uint32_t p) { return kid_splitmix64(d + p); };
    if (kind < 77) { for (uint32_t p = len - tail; p < len; p++) q[p] = (char)('#' + noise(p) % 16); }
    else if (kind < 103) { for (uint32_t p = 0; p < head; p++) q[p] = (char)('#' + noise(p) % 16); }
    else if (kind < 116) { for (uint32_t p = 0; p < len; p++) q[p] = (char)('+' + noise(p) % 30); }
    else if (kind < 121) { for (uint32_t p = 0; p < len; p++) q[p] = (char)('#' + noise(p) % 16); }
}

int main(int argc, char **argv)
{
    if (argc < 2) die("usage: kid_synth_files probes|fastq ... (see the head of tools/kid_synth_files.cpp)");
    const std::string mode = argv[1];
    std::string counts_path, tree_path, out, out_dir, qual = "mixed", prefix = "S";
    double scale = 1.0;
    int k = 30, threads = (int)std::thread::hardware_concurrency(), level = 1, samples = 1;
    uint64_t seed = DB_SEED, pairs = 0, r0 = 0;
    uint32_t read_len = 150;
    if (threads < 1) threads = 1;
    for (int i = 2; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { if (i + 1 >= argc) die(a + " needs a value"); return argv[++i]; };
        if (a == "--counts") counts_path = val();
        else if (a == "--tree") tree_path = val();
        else if (a == "--out") out = val();
        else if (a == "--out-dir") out_dir = val();
        else if (a == "--scale") scale = atof(val());
        else if (a == "--k") k = atoi(val());
        else if (a == "--seed") seed = strtoull(val(), nullptr, 0);
        else if (a == "--threads") threads = atoi(val());
        else if (a == "--level") level = atoi(val());
        else if (a == "--members") g_members = true;
        else if (a == "--samples") samples = atoi(val());
        else if (a == "--pairs") pairs = strtoull(val(), nullptr, 0);
        else if (a == "--read-len") read_len = (uint32_t)atoi(val());
        else if (a == "--qual") qual = val();
        else if (a == "--r0") r0 = strtoull(val(), nullptr, 0);
        else if (a == "--prefix") prefix = val();
        else die("unknown argument " + a);
    }
    if (threads < 1) threads = 1;
    if (k < 1 || k > 31) die("--k outside [1,31]");
    if (counts_path.empty()) die("--counts is required");
    const std::vector<uint64_t> cnt = load_counts(counts_path, scale);
    const std::vector<uint64_t> cum = cumulative(cnt);
    const int32_t ntar = (int32_t)cnt.size();
    const uint64_t n_keys = cum.back();

    if (mode == "probes") {
        if (out.empty()) die("--out is required");
        const uint64_t per_chunk = 1u << 20;
        const uint64_t n_chunks = (n_keys + per_chunk - 1) / per_chunk;
        write_gz_chunks(out, n_chunks, threads, level, [&](uint64_t c, std::string &text) {
            const uint64_t j0 = c * per_chunk, j1 = j0 + per_chunk < n_keys ? j0 + per_chunk : n_keys;
            text.reserve((size_t)(j1 - j0) * 56);
            uint32_t t = kid_synth_target_of(cum.data(), ntar, j0);
            char line[96];
            for (uint64_t j = j0; j < j1; j++) {
                while (j >= cum[t + 1]) t++; // (targets own consecutive ranges of ordinals)
                const uint64_t key = kid_synth_db_key(seed, k, j);
                for (int b = 0; b < k; b++) line[b] = "ACGT"[(key >> (2 * (k - 1 - b))) & 3];
                const int n = snprintf(line + k, sizeof(line) - (size_t)k, ",%u,0,%llu,F,1\n", t, (unsigned long long)j);
                text.append(line, (size_t)(k + n));
            }
        });
        fprintf(stderr, "kid_synth_files: %llu probe lines -> %s\n", (unsigned long long)n_keys, out.c_str());
        return 0;
    }
    if (mode == "fastq") {
        if (out_dir.empty() || tree_path.empty() || pairs == 0) die("--out-dir, --tree and --pairs are required");
        if (out_dir.back() != '/') out_dir += "/";
        if (pairs > 999999999ull) die("--pairs above 999 999 999 (the fixed-width read names hold 9 digits)");
        const std::vector<int32_t> parent = load_parent(tree_path, (size_t)ntar);
        const bool mixed = qual != "high";
        const uint64_t per_chunk = 1u << 16;
        // sample s, mate m: reads r0 + (2 s + m - 1) * pairs ... of the synthetic stream (the layout of bench.py's CLI leg)
        for (int s = 0; s < samples; s++)
            for (int mate = 1; mate <= 2; mate++) {
                const uint64_t first = r0 + (uint64_t)(2 * s + mate - 1) * pairs;
                const std::string path = out_dir + prefix + std::to_string(s) + "_R" + std::to_string(mate) + "_tr.fastq.gz";
                write_gz_chunks(path, (pairs + per_chunk - 1) / per_chunk, threads, level, [&](uint64_t c, std::string &text) {
                    const uint64_t i0 = c * per_chunk, i1 = i0 + per_chunk < pairs ? i0 + per_chunk : pairs;
                    const size_t rec = 14 + (size_t)read_len + 1 + 2 + (size_t)read_len + 1;
                    text.resize((size_t)(i1 - i0) * rec);
                    char *p = &text[0];
                    for (uint64_t i = i0; i < i1; i++) {
                        p += snprintf(p, 16, "@r%09llu/%d\n", (unsigned long long)i, mate); // 14 bytes
                        kid_synth_read(DB_SEED, READ_SEED, k, cum.data(), parent.data(), ntar, first + i, read_len, (uint8_t *)p);
                        p += read_len;
                        *p++ = '\n'; *p++ = '+'; *p++ = '\n';
                        make_quals(first + i, read_len, mixed, p);
                        p += read_len;
                        *p++ = '\n';
                    }
                });
            }
        fprintf(stderr, "kid_synth_files: %d sample(s) x %llu pairs of %u bp -> %s\n", samples, (unsigned long long)pairs, read_len, out_dir.c_str());
        return 0;
    }
    die("unknown mode " + mode);
}
