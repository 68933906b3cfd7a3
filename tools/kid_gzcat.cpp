// kid_gzcat -- test tool: the text of a .gz file on stdout, read either by the host code's own inflate loop
// (kmer_id_amd/host/kid_inflate.cpp) or, with --zlib, by zlib's gzread the way the reference reads its files
// (newkmer_10nx.cpp:762-816: 16 KiB calls).  tests/test_host_inflate.py compares the two on good, odd and damaged files.
//   kid_gzcat [--zlib] [--room BYTES] [--time] FILE      (--time: no text; the best of three passes, as MB/s of text)
//   kid_gzcat --threads N [--chunk BYTES] [--patience P] [--room BYTES] [--time] FILE     pieces of the file inflated side by side
//                                                         (kmer_id_amd/host/kid_pargz.cpp); "parallel: N bytes" on stderr
// exit 0 = read to the end and closed; 3 = a read failed (message on stderr); 4 = the close failed ("failed gzclose").
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <chrono>
#include <string>
#include <vector>

#include "../kmer_id_amd/host/kid_inflate.h"
#include "../kmer_id_amd/host/kid_pargz.h"
#include "../kmer_id_amd/host/kid_textio.h"

int main(int argc, char **argv)
{
    bool use_zlib = false, time_it = false;
    size_t room = (size_t)1 << 20, chunk = (size_t)2 << 20;
    int threads = 0;
    unsigned patience = 4;
    const char *path = nullptr;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--zlib")) use_zlib = true;
        else if (!strcmp(argv[i], "--time")) time_it = true;
        else if (!strcmp(argv[i], "--threads") && i + 1 < argc) threads = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--chunk") && i + 1 < argc) chunk = (size_t)atol(argv[++i]);
        else if (!strcmp(argv[i], "--patience") && i + 1 < argc) patience = (unsigned)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--room") && i + 1 < argc) room = (size_t)atol(argv[++i]);
        else path = argv[i];
    }
    if (!path) return 2;
    if (time_it) {
        double best = 1e30;
        size_t total = 0;
        for (int pass = 0; pass < 3; pass++) {
            const auto t0 = std::chrono::steady_clock::now();
            total = 0;
            if (use_zlib) {
                gzFile g = gzopen(path, "rb");
                if (!g) return 3;
                gzbuffer(g, 1 << 20);
                std::vector<char> buf(room);
                int n;
                while ((n = gzread(g, buf.data(), (unsigned)buf.size())) > 0) total += (size_t)n;
                gzclose(g);
            } else if (threads > 0) {
                try {
                    kidhost::ParallelGz z(path, threads, chunk, room, kidhost::GzStream::kWindow, patience);
                    kidhost::HostBuf buf;
                    size_t n;
                    while (z.next(buf, n)) total += n;
                } catch (const kidhost::Fatal &f) {
                    fprintf(stderr, "%s\n", f.message.c_str());
                    return 3;
                }
            } else {
                try {
                    kidhost::GzStream z(path);
                    std::vector<uint8_t> buf(kidhost::GzStream::kWindow + room);
                    size_t n;
                    while ((n = z.read(buf.data() + kidhost::GzStream::kWindow, room)) > 0) total += n;
                } catch (const kidhost::Fatal &f) {
                    fprintf(stderr, "%s\n", f.message.c_str());
                    return 3;
                }
            }
            const double t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (t < best) best = t;
        }
        printf("%s %s: %zu bytes of text, %.3f s, %.1f MB/s\n", use_zlib ? "zlib" : "kid ", path, total, best, (double)total / best / 1e6);
        return 0;
    }
    if (use_zlib) {
        gzFile g = gzopen(path, "rb");
        if (!g) { fprintf(stderr, "cannot open\n"); return 3; }
        std::vector<char> buf(0x4000);
        for (;;) {
            const int n = gzread(g, buf.data(), (unsigned)buf.size());
            if (n == 0) break;
            if (n < 0) {
                int err = 0;
                fprintf(stderr, "%s\n", gzerror(g, &err));
                return 3;
            }
            fwrite(buf.data(), 1, (size_t)n, stdout);
        }
        if (gzclose(g) != Z_OK) { fprintf(stderr, "failed gzclose\n"); return 4; }
        return 0;
    }
    if (threads > 0) {
        try {
            kidhost::ParallelGz z(path, threads, chunk, room, kidhost::GzStream::kWindow, patience);
            kidhost::HostBuf buf;
            try {
                size_t n;
                while (z.next(buf, n)) fwrite(buf.data() + kidhost::GzStream::kWindow, 1, n, stdout);
            } catch (const kidhost::Fatal &f) {
                fprintf(stderr, "%s\n", f.message.c_str());
                return 3;
            }
            fflush(stdout);
            fprintf(stderr, "parallel: %llu bytes\n", (unsigned long long)z.bytes_in_parallel());
            z.close();
        } catch (const kidhost::Fatal &f) {
            fprintf(stderr, "%s\n", f.message.c_str());
            return 4;
        }
        return 0;
    }
    try {
        kidhost::GzStream z(path);
        std::vector<uint8_t> buf(kidhost::GzStream::kWindow + room);
        try {
            for (;;) {
                const size_t n = z.read(buf.data() + kidhost::GzStream::kWindow, room);
                if (n == 0) break;
                fwrite(buf.data() + kidhost::GzStream::kWindow, 1, n, stdout);
            }
        } catch (const kidhost::Fatal &f) {
            fprintf(stderr, "%s\n", f.message.c_str());
            return 3;
        }
        z.close();
    } catch (const kidhost::Fatal &f) {
        fprintf(stderr, "%s\n", f.message.c_str());
        return 4;
    }
    return 0;
}
