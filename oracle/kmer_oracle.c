/*
 * kmer_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the read-classification hot
 * path of the reference program newkmer_10nx.cpp (and the two switches that
 * turn it into kmer_read_m3.cpp / kmer_read_vf6.cpp).  It exists so that the
 * HIP path can be checked bit-for-bit on a machine where the reference source
 * is absent.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load this file; the product library (libkmer_id_amd.so) never
 * links, loads or calls it.
 *
 * Parity pin: every function here is checked in tests/test_oracle_golden.py
 * against golden vectors produced by the compiled reference itself
 * (oracle/Makefile -> oracle/_ref/, oracle/make_golden.py -> tests/golden/).
 *
 * Every function cites the reference lines (file:line, CRLF-stripped numbering
 * is identical) it restates.  Nothing here is copied: data structures are
 * re-designed (mark array instead of std::set for the ancestor walk, an
 * open-addressed u64 set instead of std::set<ktype> for kmer_seen), the
 * observable results are the same.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <zlib.h>

#define KO_FLAG_U_IS_T 1u /* kmer_read_vf6.cpp:496-500,521-525: U/u counts as T */

/* newkmer_10nx.cpp:164-171 -- same 24-byte footprint as the reference cell so
 * that the CPU baseline sees the same cache behaviour. */
typedef struct {
    uint64_t key;
    uint32_t value;
    int32_t org;
    int32_t position;
    uint8_t fstrand;
} ko_cell;

typedef struct ko_db {
    int ntar;       /* MAXTAR                      newkmer_10nx.cpp:45  */
    int k;          /* KSIZE                       newkmer_10nx.cpp:43  */
    int log2_slots; /* log2(MAXHASH)               newkmer_10nx.cpp:49  */
    int max_probes; /* 0 = unbounded; 16 = MAXREPROBE of kmer_read_m3.cpp:42,232 */
    uint32_t flags;
    uint64_t nslots, size;
    uint64_t mask, hi_c, hi_g, hi_t; /* newkmer_10nx.cpp:76-79 */
    ko_cell *cells;
    uint64_t cells_bytes;
    int cells_mapped;
    int32_t *parent; /* Tree1::parent              newkmer_10nx.cpp:98  */
    uint32_t *mark;  /* scratch for msca (replaces the per-call std::set) */
    uint32_t mark_gen;
} ko_db;

typedef struct ko_sample {
    ko_db *db;
    int64_t *gcount, *ucount; /* newkmer_10nx.cpp:61-62 (int there; i64 here, printed in decimal) */
    int64_t tct;              /* newkmer_10nx.cpp:61 */
    /* kmer_seen (newkmer_10nx.cpp:64) as an open-addressed set; ~0 = empty */
    uint64_t *seen;
    uint64_t seen_cap, seen_n;
    /* statistics that are not part of the reference's outputs */
    uint64_t n_lookups, n_probes, n_hits;
} ko_sample;

/* ---- Hashtable::integerHash, newkmer_10nx.cpp:189-197 (MurmurHash3 fmix64) ---- */
uint64_t ko_fmix64(uint64_t k)
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

/* ---- Tree1::Tree1 + Hashtable::Hashtable, newkmer_10nx.cpp:101-106,173-180,199-202 ---- */
ko_db *ko_db_new(int ntar, int k, int log2_slots, int max_probes, uint32_t flags)
{
    if (ntar < 2 || k < 1 || k > 31 || log2_slots < 6 || log2_slots > 34) return NULL;
    ko_db *db = (ko_db *)calloc(1, sizeof(ko_db));
    if (!db) return NULL;
    db->ntar = ntar;
    db->k = k;
    db->log2_slots = log2_slots;
    db->max_probes = max_probes;
    db->flags = flags;
    db->nslots = 1ULL << log2_slots;
    db->mask = (1ULL << (2 * k)) - 1;
    db->hi_c = 1ULL << ((k - 1) * 2);
    db->hi_g = 2ULL << ((k - 1) * 2);
    db->hi_t = 3ULL << ((k - 1) * 2);
    /* zero-filled like HashClear() (:199-202); big tables come from mmap with
     * transparent huge pages so that the 24 GiB zero-fill takes seconds, not minutes */
    db->cells_bytes = db->nslots * sizeof(ko_cell);
    if (db->cells_bytes >= (64u << 20)) {
        void *m = mmap(NULL, db->cells_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m != MAP_FAILED) {
            madvise(m, db->cells_bytes, MADV_HUGEPAGE);
            db->cells = (ko_cell *)m;
            db->cells_mapped = 1;
        }
    }
    if (!db->cells) db->cells = (ko_cell *)calloc(db->nslots, sizeof(ko_cell));
    db->parent = (int32_t *)malloc(sizeof(int32_t) * (size_t)ntar);
    db->mark = (uint32_t *)calloc((size_t)ntar, sizeof(uint32_t));
    if (!db->cells || !db->parent || !db->mark) {
        if (db->cells_mapped) munmap(db->cells, db->cells_bytes); else free(db->cells);
        free(db->parent); free(db->mark); free(db);
        return NULL;
    }
    for (int i = 0; i < ntar; i++) db->parent[i] = 1; /* every node hangs off root=1 by default */
    return db;
}

void ko_db_free(ko_db *db)
{
    if (!db) return;
    if (db->cells_mapped) munmap(db->cells, db->cells_bytes); else free(db->cells);
    free(db->parent); free(db->mark); free(db);
}

uint64_t ko_db_size(const ko_db *db) { return db->size; }
const int32_t *ko_db_parent(const ko_db *db) { return db->parent; }

/* ---- Tree1::add_edge, newkmer_10nx.cpp:112-116 (children[] is never read on the path) ---- */
int ko_db_add_edge(ko_db *db, int x, int y)
{
    if (y < 0 || y >= db->ntar) return -1;
    db->parent[y] = x;
    return 0;
}

/* ---- Tree1::get_parent, newkmer_10nx.cpp:146-152 ---- */
static inline int ko_get_parent(const ko_db *db, int x)
{
    return (x != 1 && x > 0) ? db->parent[x] : 1;
}

/* ---- Tree1::msca, newkmer_10nx.cpp:118-144 ----
 * ancestors = {root} U path(x -> root); y in ancestors -> x; otherwise climb
 * from y: reaching x -> y, reaching any other ancestor -> that ancestor. */
int ko_msca(ko_db *db, int x, int y)
{
    uint32_t g = ++db->mark_gen;
    if (g == 0) { memset(db->mark, 0, sizeof(uint32_t) * (size_t)db->ntar); g = db->mark_gen = 1; }
    db->mark[1] = g;
    for (int z = x; z != 1; z = ko_get_parent(db, z)) db->mark[z] = g;
    if (db->mark[y] == g) return x;
    int z = y;
    while (db->mark[z] != g) {
        z = ko_get_parent(db, z);
        if (z == x) return y;
    }
    return z;
}

/* checksum of msca over all ordered pairs of nodes 1..ntar-1, weighted with a
 * splitmix64 finaliser exactly like oracle/ref_kat_driver.cpp does with the
 * reference's own msca (golden value: tests/golden/kat_10nx.npz msca_all_sum) */
uint64_t ko_msca_checksum(ko_db *db)
{
    uint64_t sum = 0;
    for (int x = 1; x < db->ntar; x++)
        for (int y = 1; y < db->ntar; y++) {
            uint64_t k = (uint64_t)x * (uint64_t)db->ntar + (uint64_t)y;
            k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ULL;
            k ^= k >> 27; k *= 0x94d049bb133111ebULL;
            k ^= k >> 31;
            sum += k * (uint64_t)ko_msca(db, x, y);
        }
    return sum;
}

/* ---- Hashtable::add_kmer, newkmer_10nx.cpp:235-263 ----
 * first cell on the triangular probe path whose value is 0 takes the entry; no
 * key comparison, so a duplicate key lands further down its own path. A target
 * of 0 writes the key but leaves the cell "empty". returns -1 where the
 * reference exits with "out of memory in table". */
int ko_db_add_kmer(ko_db *db, uint64_t key, uint32_t target)
{
    uint64_t hash = ko_fmix64(key), reprobe = 0, i = 0;
    for (;;) {
        uint64_t index = (hash + reprobe) & (db->nslots - 1);
        reprobe += ++i;
        if (db->cells[index].value == 0) {
            db->cells[index].key = key;
            db->cells[index].value = target;
            if (++db->size > db->nslots - 32) return -1;
            return 0;
        }
    }
}

int ko_db_add_batch(ko_db *db, const uint64_t *keys, const uint32_t *targets, uint64_t n)
{
    for (uint64_t i = 0; i < n; i++)
        if (ko_db_add_kmer(db, keys[i], targets[i]) != 0) return -1;
    return 0;
}

/* ---- Hashtable::getHash, newkmer_10nx.cpp:204-233; probe cap of kmer_read_m3.cpp:232 ---- */
uint32_t ko_db_get(const ko_db *db, uint64_t key, uint32_t *probes_out)
{
    uint64_t hash = ko_fmix64(key), reprobe = 0, i = 0;
    uint32_t res = 0;
    do {
        uint64_t index = (hash + reprobe) & (db->nslots - 1);
        reprobe += ++i;
        const ko_cell *c = &db->cells[index];
        if (c->value == 0) break;
        if (c->key == key) { res = c->value; break; }
    } while (reprobe < db->nslots && (db->max_probes == 0 || i < (uint64_t)db->max_probes));
    if (probes_out) *probes_out = (uint32_t)i;
    return res;
}

/* batch form used by the unit-parity tests */
void ko_db_get_batch(const ko_db *db, const uint64_t *keys, uint64_t n, uint32_t *targets, uint32_t *probes)
{
    for (uint64_t j = 0; j < n; j++) {
        uint32_t p;
        targets[j] = ko_db_get(db, keys[j], &p);
        if (probes) probes[j] = p;
    }
}

/* ---- process_kmer, newkmer_10nx.cpp:619-661: forward key only, upper-case ACGT only ---- */
int ko_db_process_kmer(ko_db *db, const char *seq, size_t len, uint32_t target)
{
    int cpos = 0;
    uint64_t keyF = 0;
    for (size_t p = 0; p < len; p++) {
        int code;
        switch (seq[p]) {
        case 'A': code = 0; break;
        case 'C': code = 1; break;
        case 'G': code = 2; break;
        case 'T': code = 3; break;
        default: code = -1; break;
        }
        if (code < 0) { cpos = 0; keyF = 0; }
        else { keyF = ((keyF << 2) & db->mask) | (uint64_t)code; cpos++; }
        if (cpos == db->k) {
            if (ko_db_add_kmer(db, keyF, target) != 0) return -1;
            cpos--;
        }
    }
    return 0;
}

/* whitespace-delimited token scanner equivalent to `istringstream >> x` for the
 * six fields of a probes line (newkmer_10nx.cpp:695-697) */
static const char *ko_skip_ws(const char *p, const char *e)
{
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\v' || *p == '\f' || *p == '\r')) p++;
    return p;
}

/* parse a decimal integer the way operator>>(int/unsigned) does for the inputs
 * the probes format can contain: optional sign, at least one digit. */
static int ko_scan_int(const char **pp, const char *e, long long *out)
{
    const char *p = ko_skip_ws(*pp, e);
    int neg = 0;
    if (p < e && (*p == '+' || *p == '-')) { neg = (*p == '-'); p++; }
    if (p >= e || *p < '0' || *p > '9') return 0;
    long long v = 0;
    while (p < e && *p >= '0' && *p <= '9') { v = v * 10 + (*p - '0'); if (v > (1LL << 40)) return 0; p++; }
    *out = neg ? -v : v;
    *pp = p;
    return 1;
}

/* one line of the probes file: SEQ,target,org,position,strand,count
 * (format written by kmer_build_vf6.cpp:625).  returns 1 if the line parsed and
 * was inserted, 0 if skipped, -1 on table overflow. */
int ko_db_probe_line(ko_db *db, const char *line, size_t len)
{
    char tmp[0x4000];
    if (len == 0 || len >= sizeof(tmp)) return 0;
    for (size_t i = 0; i < len; i++) tmp[i] = (line[i] == ',') ? ' ' : line[i];
    const char *p = tmp, *e = tmp + len;
    p = ko_skip_ws(p, e);
    const char *s0 = p;
    while (p < e && !(*p == ' ' || *p == '\t' || *p == '\v' || *p == '\f' || *p == '\r')) p++;
    size_t slen = (size_t)(p - s0);
    if (slen == 0) return 0;
    long long target, org, position, count;
    /* vtype target is unsigned: a negative literal fails extraction in practice
     * (wraps + failbit is implementation detail); well-formed files never have one */
    if (!ko_scan_int(&p, e, &target) || target < 0 || target > 0xffffffffLL) return 0;
    if (!ko_scan_int(&p, e, &org)) return 0;
    if (!ko_scan_int(&p, e, &position)) return 0;
    p = ko_skip_ws(p, e);
    if (p >= e) return 0;
    p++; /* strand char */
    if (!ko_scan_int(&p, e, &count)) return 0;
    if (ko_db_process_kmer(db, s0, slen, (uint32_t)target) != 0) return -1;
    return 1;
}

/* ---- process_kmergz, newkmer_10nx.cpp:663-712 ----
 * 16 KiB rolling buffer over gzread; lines split at '\n', one trailing '\r'
 * removed, empty lines skipped, unterminated tail dropped.  returns the number
 * of parsed lines (tct, printed as "<n> kmers loaded"), -1 on I/O error,
 * -2 on a line that fills the buffer, -3 on table overflow. */
long long ko_db_load_probes_gz(ko_db *db, const char *path)
{
    gzFile in = gzopen(path, "rb");
    if (!in) return -1;
    char buf[0x4000];
    size_t pending = 0;
    long long tct = 0;
    for (;;) {
        int room = (int)(sizeof(buf) - pending);
        if (room == 0) { gzclose(in); return -2; }
        int got = gzread(in, buf + pending, (unsigned)room);
        if (got == 0) break;
        if (got < 0) { gzclose(in); return -1; }
        char *cur = buf, *end = buf + pending + got;
        for (;;) {
            char *eol = (char *)memchr(cur, '\n', (size_t)(end - cur));
            if (!eol) break;
            size_t len = (size_t)(eol - cur);
            if (len > 0 && cur[len - 1] == '\r') len--;
            if (len > 0) {
                int r = ko_db_probe_line(db, cur, len);
                if (r < 0) { gzclose(in); return -3; }
                tct += r;
            }
            cur = eol + 1;
        }
        pending = (size_t)(end - cur);
        memmove(buf, cur, pending);
    }
    if (gzclose(in) != Z_OK) return -1;
    return tct;
}

/* ------------------------------------------------------------------ sample */

ko_sample *ko_sample_new(ko_db *db)
{
    ko_sample *s = (ko_sample *)calloc(1, sizeof(ko_sample));
    if (!s) return NULL;
    s->db = db;
    s->gcount = (int64_t *)calloc((size_t)db->ntar, sizeof(int64_t));
    s->ucount = (int64_t *)calloc((size_t)db->ntar, sizeof(int64_t));
    s->seen_cap = 1024;
    s->seen = (uint64_t *)malloc(sizeof(uint64_t) * s->seen_cap);
    if (!s->gcount || !s->ucount || !s->seen) { free(s->gcount); free(s->ucount); free(s->seen); free(s); return NULL; }
    memset(s->seen, 0xff, sizeof(uint64_t) * s->seen_cap);
    return s;
}

void ko_sample_free(ko_sample *s)
{
    if (!s) return;
    free(s->gcount); free(s->ucount); free(s->seen); free(s);
}

/* per-sample reset, newkmer_10nx.cpp:1017-1019,1023 */
void ko_sample_reset(ko_sample *s)
{
    memset(s->gcount, 0, sizeof(int64_t) * (size_t)s->db->ntar);
    memset(s->ucount, 0, sizeof(int64_t) * (size_t)s->db->ntar);
    memset(s->seen, 0xff, sizeof(uint64_t) * s->seen_cap);
    s->seen_n = 0;
    s->tct = 0;
    s->n_lookups = s->n_probes = s->n_hits = 0;
}

const int64_t *ko_sample_gcount(const ko_sample *s) { return s->gcount; }
const int64_t *ko_sample_ucount(const ko_sample *s) { return s->ucount; }
int64_t ko_sample_tct(const ko_sample *s) { return s->tct; }
void ko_sample_stats(const ko_sample *s, uint64_t *out3)
{
    out3[0] = s->n_lookups; out3[1] = s->n_probes; out3[2] = s->n_hits;
}

/* kmer_seen.find + insert (newkmer_10nx.cpp:598-602); returns 1 if newly added */
static int ko_seen_add(ko_sample *s, uint64_t key)
{
    if ((s->seen_n + 1) * 2 > s->seen_cap) {
        uint64_t ncap = s->seen_cap * 2;
        uint64_t *nt = (uint64_t *)malloc(sizeof(uint64_t) * ncap);
        if (!nt) abort();
        memset(nt, 0xff, sizeof(uint64_t) * ncap);
        for (uint64_t i = 0; i < s->seen_cap; i++) {
            uint64_t v = s->seen[i];
            if (v == ~0ULL) continue;
            uint64_t h = ko_fmix64(v) & (ncap - 1);
            while (nt[h] != ~0ULL) h = (h + 1) & (ncap - 1);
            nt[h] = v;
        }
        free(s->seen);
        s->seen = nt;
        s->seen_cap = ncap;
    }
    uint64_t h = ko_fmix64(key) & (s->seen_cap - 1);
    while (s->seen[h] != ~0ULL) {
        if (s->seen[h] == key) return 0;
        h = (h + 1) & (s->seen_cap - 1);
    }
    s->seen[h] = key;
    s->seen_n++;
    return 1;
}

/* ---- process_read, newkmer_10nx.cpp:452-617 (alignment branch :530-587 is dead at minalign=0) ----
 * seq[start..stop] inclusive.  Returns final_targ; bumps gcount/ucount/tct.
 * save_out (nullable) receives 1 when the reference would append the read to
 * _reads.txt (:608-612; test precedes the gcount increment). */
int ko_process_read(ko_sample *s, const char *seq, int start, int stop, int *save_out)
{
    ko_db *db = s->db;
    int cpos = 0;
    uint64_t keyF = 0, keyR = 0;
    int final_targ = 0;
    const int rshift = 2;
    for (int it = start; it <= stop; ++it) {
        int code;
        switch (seq[it]) {
        case 'A': case 'a': code = 0; break;
        case 'C': case 'c': code = 1; break;
        case 'G': case 'g': code = 2; break;
        case 'T': case 't': code = 3; break;
        case 'U': case 'u': code = (db->flags & KO_FLAG_U_IS_T) ? 3 : -1; break;
        default: code = -1; break;
        }
        if (code < 0) {
            cpos = 0; keyF = 0; keyR = 0; /* :520-524 */
        } else {
            /* :480-519: forward key shifts the new base in at the bottom, the
             * reverse-complement key shifts the complement in at the top */
            keyF = ((keyF << 2) & db->mask) | (uint64_t)code;
            uint64_t comp_hi = code == 0 ? db->hi_t : code == 1 ? db->hi_g : code == 2 ? db->hi_c : 0;
            keyR = (keyR >> rshift) | comp_hi;
            cpos++;
        }
        if (cpos == db->k) { /* :526 */
            uint64_t key = keyF < keyR ? keyF : keyR;
            uint32_t probes;
            int target = (int)ko_db_get(db, key, &probes);
            s->n_lookups++;
            s->n_probes += probes;
            if (target > 0) s->n_hits++;
            if (final_targ > 0 && target > 0)      /* :588-591 */
                final_targ = ko_msca(db, target, final_targ);
            else if (target > 0)                   /* :592-595 */
                final_targ = target;
            if (target > 1 && ko_seen_add(s, key)) /* :596-603 */
                s->ucount[target]++;
            cpos--;                                /* :604 */
        }
    }
    int save = (final_targ > 1 && s->gcount[final_targ] < 12); /* SAVENUM :48,:608 */
    if (save_out) *save_out = save;
    s->gcount[final_targ]++; /* :613 */
    s->tct++;                /* :614 */
    return final_targ;
}

/* batch form over concatenated reads: bases + offsets[n+1] + start/stop relative to each read */
void ko_classify_batch(ko_sample *s, const uint8_t *bases, const uint64_t *offsets,
                       const int32_t *start, const int32_t *stop, uint64_t n, uint32_t *final_out)
{
    for (uint64_t r = 0; r < n; r++) {
        int f = ko_process_read(s, (const char *)bases + offsets[r], start[r], stop[r], NULL);
        if (final_out) final_out[r] = (uint32_t)f;
    }
}

/* wall-clock timed form for the cpu_baseline leg of bench.py: seconds for n reads */
double ko_classify_batch_timed(ko_sample *s, const uint8_t *bases, const uint64_t *offsets,
                               const int32_t *start, const int32_t *stop, uint64_t n)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    ko_classify_batch(s, bases, offsets, start, stop, n, NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- the same loop on several host threads (bench.py's all-cores cpu_baseline leg) ----
 * The reference is single-threaded; this is what "the reference's CPU path on all host cores" can
 * mean without changing its results: reads are independent given the table, so thread t runs
 * ko_process_read over a contiguous range of the reads into counters of its own (the table and the
 * tree are shared read-only; msca's scratch array is per thread), and the counters are merged the
 * way the reference's globals would have ended up: gcount adds, kmer_seen is the union of the
 * threads' sets, ucount counts the union once per key.  Returns wall seconds, merge included. */
#include <pthread.h>
typedef struct ko_mt_arg {
    ko_db dbc; /* shallow copy: own msca scratch */
    ko_sample *s;
    const uint8_t *bases;
    const uint64_t *offsets;
    const int32_t *start, *stop;
    uint64_t r0, r1;
} ko_mt_arg;

static void *ko_mt_worker(void *p)
{
    ko_mt_arg *a = (ko_mt_arg *)p;
    for (uint64_t r = a->r0; r < a->r1; r++)
        ko_process_read(a->s, (const char *)a->bases + a->offsets[r], a->start[r], a->stop[r], NULL);
    return NULL;
}

double ko_classify_batch_mt(ko_db *db, const uint8_t *bases, const uint64_t *offsets, const int32_t *start,
                            const int32_t *stop, uint64_t n, int nthreads, int64_t *gcount_out, int64_t *ucount_out,
                            uint64_t *stats3)
{
    if (nthreads < 1) nthreads = 1;
    if ((uint64_t)nthreads > n && n > 0) nthreads = (int)n;
    ko_mt_arg *args = (ko_mt_arg *)calloc((size_t)nthreads, sizeof(ko_mt_arg));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    if (!args || !th) { free(args); free(th); return -1.0; }
    for (int t = 0; t < nthreads; t++) {
        args[t].dbc = *db;
        args[t].dbc.mark = (uint32_t *)calloc((size_t)db->ntar, sizeof(uint32_t));
        args[t].dbc.mark_gen = 0;
        args[t].s = ko_sample_new(&args[t].dbc);
        if (!args[t].dbc.mark || !args[t].s) {
            for (int u = 0; u <= t; u++) { ko_sample_free(args[u].s); free(args[u].dbc.mark); }
            free(args); free(th);
            return -1.0;
        }
        args[t].bases = bases; args[t].offsets = offsets; args[t].start = start; args[t].stop = stop;
        args[t].r0 = n * (uint64_t)t / (uint64_t)nthreads;
        args[t].r1 = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, ko_mt_worker, &args[t]);
    ko_mt_worker(&args[0]);
    for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
    ko_sample *m = args[0].s;
    for (int t = 1; t < nthreads; t++) {
        const ko_sample *o = args[t].s;
        for (int i = 0; i < db->ntar; i++) m->gcount[i] += o->gcount[i];
        m->tct += o->tct;
        m->n_lookups += o->n_lookups; m->n_probes += o->n_probes; m->n_hits += o->n_hits;
        for (uint64_t i = 0; i < o->seen_cap; i++) {
            const uint64_t key = o->seen[i];
            if (key == ~0ULL || !ko_seen_add(m, key)) continue;
            uint32_t probes;
            m->ucount[ko_db_get(db, key, &probes)]++; /* the target the key was credited to (:600) */
        }
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    memcpy(gcount_out, m->gcount, sizeof(int64_t) * (size_t)db->ntar);
    memcpy(ucount_out, m->ucount, sizeof(int64_t) * (size_t)db->ntar);
    if (stats3) { stats3[0] = m->n_lookups; stats3[1] = m->n_probes; stats3[2] = m->n_hits; }
    for (int t = 0; t < nthreads; t++) { ko_sample_free(args[t].s); free(args[t].dbc.mark); }
    free(args); free(th);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- process_qual, newkmer_10nx.cpp:714-760 ----
 * qual bytes are compared as (signed) char exactly like std::string::at()
 * returns on x86.  Returns 1 if process_read would be called (stop-start >= k),
 * 0 if the read is dropped, -1 where qual.at() would throw (qual shorter than seq). */
int ko_process_qual(const char *qual, int seqlen, int quallen, int k, int *start_out, int *stop_out)
{
    const int cutoff_qual = 17, window_size = 4;
    const signed char cutoff_char = 32 + cutoff_qual;
    const int window_cut = cutoff_qual * window_size;
    const signed char *q = (const signed char *)qual;
    int stop = seqlen - 1, start = 0, w;
    if (seqlen <= 0 || quallen < seqlen) return -1;
    while (q[start] < cutoff_char && start < stop) start++;
    while (q[stop] < cutoff_char && stop > start) stop--;
    if (start < stop - window_size) {
        w = 0;
        for (int i = 0; i < window_size; i++) w += q[start + i] - 32;
        while (w < window_cut && start < stop - window_size) {
            w += q[start + window_size] - q[start];
            start++;
        }
    }
    if (start < stop - window_size) {
        w = 0;
        for (int i = 0; i < window_size; i++) w += q[stop - i] - 32;
        while (w < window_cut && start < stop - window_size) {
            w += q[stop - window_size] - q[stop];
            stop--;
        }
    }
    *start_out = start;
    *stop_out = stop;
    return (stop - start >= k) ? 1 : 0;
}

/* ---- process_fqgz, newkmer_10nx.cpp:762-816 ----
 * reads_out (nullable FILE*) receives the _reads.txt records.  returns 0, or
 * -1 gz error (reference: exit 255), -2 line fills the 16 KiB buffer (exit 255),
 * -4 qual shorter than seq (reference: uncaught std::out_of_range). */
int ko_process_fqgz(ko_sample *s, const char *path, FILE *reads_out)
{
    gzFile in = gzopen(path, "rb");
    if (!in) return -1; /* gzread(NULL) -> len<0 -> error() */
    char buf[0x4000];
    size_t pending = 0;
    int mod4 = 0, rc = 0;
    char *seq = (char *)malloc(0x4000), *acc = (char *)malloc(0x4000);
    size_t seqlen = 0, acclen = 0;
    for (;;) {
        int room = (int)(sizeof(buf) - pending);
        if (room == 0) { rc = -2; break; }
        int got = gzread(in, buf + pending, (unsigned)room);
        if (got == 0) break;
        if (got < 0) { rc = -1; break; }
        char *cur = buf, *end = buf + pending + got;
        for (;;) {
            char *eol = (char *)memchr(cur, '\n', (size_t)(end - cur));
            if (!eol) break;
            size_t len = (size_t)(eol - cur);
            if (len > 0 && cur[len - 1] == '\r') len--;
            if (len > 0) { /* empty lines do not advance mod4 (:788) */
                if (mod4 == 1) { memcpy(seq, cur, len); seqlen = len; }
                else if (mod4 == 0) { memcpy(acc, cur, len); acclen = len; }
                else if (mod4 == 3) {
                    int st, sp;
                    int r = ko_process_qual(cur, (int)seqlen, (int)len, s->db->k, &st, &sp);
                    if (r < 0) { rc = -4; goto done; }
                    if (r == 1) {
                        int save;
                        int f = ko_process_read(s, seq, st, sp, &save);
                        if (save && reads_out) { /* :611 */
                            fprintf(reads_out, ">%d:%.*s\n%.*s\n", f, (int)acclen, acc, sp - st + 1, seq + st);
                        }
                    }
                }
                mod4 = (mod4 + 1) % 4;
            }
            cur = eol + 1;
        }
        pending = (size_t)(end - cur);
        memmove(buf, cur, pending);
    }
done:
    free(seq); free(acc);
    if (rc != 0) { gzclose(in); return rc; }
    if (gzclose(in) != Z_OK) return -1;
    return 0;
}

/* ---- per-sample body of main, newkmer_10nx.cpp:1017-1043 ----
 * dir must end in '/', exactly like argv[1] of the reference. */
int ko_run_sample(ko_sample *s, const char *dir, const char *prefix, const char *e1, const char *e2)
{
    char path[4096];
    ko_sample_reset(s);
    snprintf(path, sizeof(path), "%s%s_reads.txt", dir, prefix);
    FILE *reads = fopen(path, "w");
    if (!reads) return -5;
    snprintf(path, sizeof(path), "%s%s%s", dir, prefix, e1);
    int rc = ko_process_fqgz(s, path, reads);
    if (rc == 0) {
        snprintf(path, sizeof(path), "%s%s%s", dir, prefix, e2);
        rc = ko_process_fqgz(s, path, reads);
    }
    fclose(reads);
    if (rc != 0) return rc;
    snprintf(path, sizeof(path), "%s%s_result.txt", dir, prefix);
    FILE *out = fopen(path, "w");
    if (!out) return -5;
    for (int i = 0; i < s->db->ntar; i++)
        fprintf(out, "%d,%lld,%lld\n", i, (long long)s->gcount[i], (long long)s->ucount[i]);
    fclose(out);
    return 0;
}

/* tree loader, newkmer_10nx.cpp:973-983: `linestream >> i >> j; add_edge(i,j)`
 * per line.  C++11 extraction semantics are kept: if the first number fails, i
 * becomes 0 and j keeps its previous value; if only the second fails, j becomes
 * 0.  (The shipped tree files have no such lines.)  returns -1 if the file is
 * missing (the reference silently continues with an all-root tree), -2 if an
 * edge names a child outside [0,ntar) (out-of-bounds write in the reference). */
int ko_db_load_tree(ko_db *db, const char *path)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    char line[4096];
    long long i = 0, j = 0;
    int have_j = 0, rc = 0;
    while (fgets(line, sizeof(line), f)) {
        const char *p = line, *e = line + strlen(line);
        long long v;
        if (ko_scan_int(&p, e, &v)) {
            i = v;
            if (ko_scan_int(&p, e, &v)) j = v; else j = 0;
            have_j = 1;
        } else {
            i = 0;
            if (!have_j) continue;
        }
        if (ko_db_add_edge(db, (int)i, (int)j) != 0) rc = -2;
    }
    fclose(f);
    return rc;
}
