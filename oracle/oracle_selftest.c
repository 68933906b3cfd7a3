/* oracle_selftest.c -- TEST INFRASTRUCTURE: drives oracle/kmer_oracle.c through every entry point on
 * seeded inputs so that `make -C oracle sanitize` can run the restatement under AddressSanitizer and
 * UBSan.  Values are checked against the golden vectors by the Python tests;
 * here the point is memory safety and defined behaviour on the edge cases -- plus one functional
 * check the Python tests lean on: the multi-threaded driver must give the single-threaded counts.
 *
 *   usage: oracle_selftest_asan <tests/golden directory>
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct ko_db ko_db;
typedef struct ko_sample ko_sample;
uint64_t ko_fmix64(uint64_t k);
ko_db *ko_db_new(int ntar, int k, int log2_slots, int max_probes, uint32_t flags);
void ko_db_free(ko_db *db);
int ko_db_add_edge(ko_db *db, int x, int y);
int ko_msca(ko_db *db, int x, int y);
int ko_db_add_kmer(ko_db *db, uint64_t key, uint32_t target);
uint32_t ko_db_get(const ko_db *db, uint64_t key, uint32_t *probes_out);
int ko_db_process_kmer(ko_db *db, const char *seq, size_t len, uint32_t target);
int ko_db_probe_line(ko_db *db, const char *line, size_t len);
ko_sample *ko_sample_new(ko_db *db);
void ko_sample_free(ko_sample *s);
void ko_sample_reset(ko_sample *s);
const int64_t *ko_sample_gcount(const ko_sample *s);
const int64_t *ko_sample_ucount(const ko_sample *s);
int ko_process_read(ko_sample *s, const char *seq, int start, int stop, int *save_out);
void ko_classify_batch(ko_sample *s, const uint8_t *bases, const uint64_t *offsets, const int32_t *start, const int32_t *stop,
                       uint64_t n, uint32_t *final_out);
double ko_classify_batch_mt(ko_db *db, const uint8_t *bases, const uint64_t *offsets, const int32_t *start, const int32_t *stop,
                            uint64_t n, int nthreads, int64_t *gcount_out, int64_t *ucount_out, uint64_t *stats3);
int ko_process_qual(const char *qual, int seqlen, int quallen, int k, int *start_out, int *stop_out);
int ko_process_fqgz(ko_sample *s, const char *path, FILE *reads_out);

static uint64_t rng_state = 0x1234567ULL;
static uint64_t rnd(void)
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "selftest: %s failed at line %d\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int ntar = 600, k = 30;
    ko_db *db = ko_db_new(ntar, k, 16, 0, 0);
    CHECK(db != NULL);
    for (int y = 2; y < ntar; y++) { /* a random forest under root 1, depth <= 6; some nodes stay without an edge */
        if (y % 11 == 0) continue;
        int x = y < 40 ? 1 : 2 + (int)(rnd() % (uint64_t)(y / 3));
        CHECK(ko_db_add_edge(db, x, y) == 0);
    }
    CHECK(ko_db_add_edge(db, 3, ntar + 5) != 0); /* child outside [0,ntar): must be refused, not written */
    for (int i = 0; i < 20000; i++) {
        int x = (int)(rnd() % (uint64_t)ntar), y = (int)(rnd() % (uint64_t)ntar);
        int m = ko_msca(db, x ? x : 1, y ? y : 1);
        CHECK(m >= 1 && m < ntar);
    }
    /* entries: random canonical-or-not keys, duplicates with another target, target 0 */
    enum { NKEYS = 20000 };
    static uint64_t keys[NKEYS];
    for (int i = 0; i < NKEYS; i++) {
        keys[i] = rnd() & ((1ULL << 60) - 1);
        CHECK(ko_db_add_kmer(db, keys[i], (uint32_t)(i % 17 == 0 ? 0 : 2 + rnd() % (uint64_t)(ntar - 2))) == 0);
    }
    for (int i = 0; i < 500; i++) CHECK(ko_db_add_kmer(db, keys[i * 7], (uint32_t)(2 + i % 50)) == 0);
    uint32_t probes = 0;
    for (int i = 0; i < NKEYS; i++) (void)ko_db_get(db, keys[i], &probes);
    (void)ko_db_get(db, 0, &probes);
    /* probes-file lines: good, short, malformed, over-long sequence, lower case */
    const char *lines[] = {"ACGTACGTACGTACGTACGTACGTACGTAC,5,1,2,F,1", "ACGT,5,1,2,F,1", "garbage", ",,,,,",
                           "ACGTACGTACGTACGTACGTACGTACGTACGTACGTAC,7,1,2,R,3", "acgtacgtacgtacgtacgtacgtacgtac,5,1,2,F,1", ""};
    for (size_t i = 0; i < sizeof(lines) / sizeof(lines[0]); i++) (void)ko_db_probe_line(db, lines[i], strlen(lines[i]));
    CHECK(ko_db_process_kmer(db, "ACGTNACGTACGTACGTACGTACGTACGTACGTACGTACGT", 41, 9) >= 0);

    /* reads: lengths around k, N, lower case, U, implanted DB k-mers on either strand */
    enum { NREADS = 6000, MAXLEN = 260 };
    uint8_t *bases = (uint8_t *)malloc((size_t)NREADS * MAXLEN);
    uint64_t *off = (uint64_t *)malloc(sizeof(uint64_t) * (NREADS + 1));
    int32_t *st = (int32_t *)malloc(sizeof(int32_t) * NREADS), *sp = (int32_t *)malloc(sizeof(int32_t) * NREADS);
    CHECK(bases && off && st && sp);
    uint64_t pos = 0;
    for (int r = 0; r < NREADS; r++) {
        const int lens[] = {0, 1, 29, 30, 31, 59, 60, 61, 100, 150, 250};
        int len = lens[rnd() % 11];
        off[r] = pos;
        const char *al = (rnd() % 50 == 0) ? "acgt" : "ACGT";
        for (int i = 0; i < len; i++) bases[pos + (uint64_t)i] = (uint8_t)al[rnd() & 3];
        if (len >= 60 && rnd() % 2) { /* implant 1-3 DB keys */
            int n = 1 + (int)(rnd() % 3);
            for (int j = 0; j < n && (j + 1) * 30 <= len; j++) {
                uint64_t v = keys[rnd() % NKEYS];
                int rc = (int)(rnd() & 1);
                for (int i = 0; i < 30; i++) {
                    int c = rc ? 3 - (int)((v >> (2 * i)) & 3) : (int)((v >> (2 * (29 - i))) & 3);
                    bases[pos + (uint64_t)(j * 30 + i)] = (uint8_t)"ACGT"[c];
                }
            }
        }
        if (len > 0 && rnd() % 20 == 0) bases[pos + rnd() % (uint64_t)len] = (uint8_t)"NUu-"[rnd() & 3];
        st[r] = 0; sp[r] = len - 1;
        if (len > 40 && rnd() % 4 == 0) { st[r] = (int32_t)(rnd() % 10); sp[r] = len - 1 - (int32_t)(rnd() % 10); }
        pos += (uint64_t)len;
    }
    off[NREADS] = pos;
    ko_sample *s = ko_sample_new(db);
    CHECK(s != NULL);
    uint32_t *fin = (uint32_t *)malloc(sizeof(uint32_t) * NREADS);
    CHECK(fin != NULL);
    ko_classify_batch(s, bases, off, st, sp, NREADS, fin);
    int64_t *g = (int64_t *)malloc(sizeof(int64_t) * ntar), *u = (int64_t *)malloc(sizeof(int64_t) * ntar);
    CHECK(g && u);
    for (int nt = 1; nt <= 5; nt += 2) { /* 1, 3, 5 threads: the merged counters must equal the sequential run */
        uint64_t stats[3];
        CHECK(ko_classify_batch_mt(db, bases, off, st, sp, NREADS, nt, g, u, stats) >= 0.0);
        CHECK(memcmp(g, ko_sample_gcount(s), sizeof(int64_t) * ntar) == 0);
        CHECK(memcmp(u, ko_sample_ucount(s), sizeof(int64_t) * ntar) == 0);
    }
    int save = 0;
    (void)ko_process_read(s, "ACGT", 0, 3, &save);
    (void)ko_process_read(s, "", 0, -1, &save);

    /* process_qual: random strings, all-low, short, quality shorter than the sequence */
    for (int i = 0; i < 20000; i++) {
        char q[300];
        int len = (int)(rnd() % 260), a, b;
        for (int j = 0; j < len; j++) q[j] = (char)(33 + rnd() % (i % 3 == 0 ? 8 : 42));
        (void)ko_process_qual(q, len, len, k, &a, &b);
        if (len > 2) (void)ko_process_qual(q, len, len - 2, k, &a, &b);
    }

    /* the FASTQ.gz reader on the golden files (CRLF, blank lines, missing final newline, ragged lengths) */
    if (argc > 1) {
        const char *files[] = {"e2e_small/S1_R1_tr.fastq.gz", "e2e_small/S1_R2_tr.fastq.gz", "e2e_small/S2_R1_tr.fastq.gz",
                               "e2e_small/S2_R2_tr.fastq.gz"};
        FILE *sink = fopen("/dev/null", "w");
        CHECK(sink != NULL);
        ko_sample_reset(s);
        for (size_t i = 0; i < 4; i++) {
            char path[4096];
            snprintf(path, sizeof(path), "%s/%s", argv[1], files[i]);
            CHECK(ko_process_fqgz(s, path, sink) == 0);
        }
        CHECK(ko_process_fqgz(s, "/nonexistent/file.fastq.gz", sink) != 0);
        fclose(sink);
    }
    free(g); free(u); free(fin); free(bases); free(off); free(st); free(sp);
    ko_sample_free(s);
    ko_db_free(db);
    printf("oracle selftest ok\n");
    return 0;
}
