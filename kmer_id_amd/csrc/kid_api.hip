// kid_api.hip -- C ABI (include/kmer_id_amd.h) over the gfx950 kernels.
// Host side: handle bookkeeping, tree preparation, the reference-order host
// table builder, launches.  No classification work is done on the CPU.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <sched.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kmer_id_amd.h"
#include "kid_kernels.hip.h"

static thread_local std::string g_last_error;

static int kid_fail(int status, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return status;
}

#define KID_HIP(call)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (call);                                                                                    \
        if (e_ != hipSuccess)                                                                                      \
            return kid_fail(e_ == hipErrorOutOfMemory ? KID_ERR_NOMEM : KID_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                            hipGetErrorString(e_), __FILE__, __LINE__);                                            \
    } while (0)

// temporaries of the unit entry points: freed on every exit path, KID_HIP's early returns included
struct KidDevBuf {
    void *p = nullptr;
    KidDevBuf() {}
    KidDevBuf(const KidDevBuf &) = delete;
    KidDevBuf &operator=(const KidDevBuf &) = delete;
    ~KidDevBuf() { if (p) hipFree(p); }
    hipError_t alloc(size_t nbytes) { return hipMalloc(&p, nbytes ? nbytes : 16); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};
struct KidEvent {
    hipEvent_t e = nullptr;
    KidEvent() {}
    KidEvent(const KidEvent &) = delete;
    KidEvent &operator=(const KidEvent &) = delete;
    ~KidEvent() { if (e) hipEventDestroy(e); }
    hipError_t create() { return hipEventCreate(&e); }
    hipEvent_t release() { hipEvent_t r = e; e = nullptr; return r; }
};

struct kid_db {
    int device = 0;
    int num_cu = 0;
    KidDevDb d{};
    uint4 *table = nullptr;
    uint4 *rows = nullptr;
    int32_t *parent = nullptr;
    int32_t *depth = nullptr;
    uint32_t *ord_target = nullptr; // target of entry o as handed to the builder, padded with zeros to a multiple of 128
    uint64_t seen_bits = 0;         // entries rounded up to whole 16-byte groups of the seen-bitmap
    kid_db_info info{};
};

struct kid_sample {
    kid_db *db = nullptr;
    unsigned long long *gcount = nullptr; // [ntar]
    unsigned long long *ucount = nullptr; // [ntar]
    unsigned long long *stats = nullptr;  // [8]
    uint32_t *seen = nullptr;
    uint64_t seen_words = 0;
    hipStream_t stream = nullptr;
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timed; // around each classify launch, while timing is on
    uint64_t timed_batches = 0;
    uint32_t batch_seq = 0;
    // Per-batch scratch of the device pipeline (prepare -> classify), grown on demand.  Three sets taken in turn, so
    // that the prepare kernel of batch b + 1 can run (on another stream) while the classify kernels of batch b still
    // read theirs: the descriptors and the device argument block the kernels find them in.  (The read text is not
    // copied: the classify kernels read the caller's ASCII and pack in registers.  Only a batch with very long records
    // gets a packed image, for the long-record kernels.)
    struct Scratch {
        KidReadDesc *desc = nullptr;
        uint64_t desc_cap = 0;
        KidLongList *long_list = nullptr; // very long records of the batch: flagged by the prepare kernel ...
        KidLongPlan *long_plan = nullptr; // ... placed by kid_long_plan_kernel (allocated with the first batch that can hold one)
        KidRareArgs *rare = nullptr;  // device copy, written in kid_sample_begin (per batch: batch_max, desc, out_final)
        hipEvent_t ev_prep = nullptr; // the prepare kernel (+ pack) of the batch using the set is done
        hipEvent_t ev_used = nullptr; // ... its classify kernels are done: the set may be overwritten (recorded when a pack on another stream asks)
        hipStream_t used_stream = nullptr; // the stream those classify kernels were queued on
        bool used_recorded = false;        // ev_used was recorded right behind them
        bool used = false;
    };
    static const int NSET = 3;
    Scratch sets[NSET];
    uint32_t next_set = 0;
    hipStream_t prep_stream = nullptr; // the prepare kernel of kid_classify_batch_device when the caller promised KID_OPT_INPUTS_READY
    bool inputs_ready = false;
    int64_t long_kmers = 65536; // records of more k-mers than this take the long-record kernels (KID_OPT_LONG_RECORD_KMERS)
    // fixed-layout batches have no descriptors and no prepare kernel: one argument block of their own, rewritten (a
    // one-thread kernel in stream order) only when a launch differs from what the block holds
    KidRareArgs *rare_fixed = nullptr;
    struct { uint32_t *out_final = nullptr; uint64_t read0 = 0; uint32_t fixed_len = 0; int32_t fixed_nk = 0; bool valid = false; } fixed_held;
    uint64_t dev_clock_batches = 0; // batches since kid_sample_kernel_time_device was last asked
    // the hit log (kid_seenlog_* in kid_kernels.hip.h): where the resolver leaves the entry ordinals of its hits, and the
    // scratch of the pass that turns them into bits of `seen`
    uint32_t *seen_log = nullptr, *seen_log_tail = nullptr, *seen_sorted = nullptr, *log_counts = nullptr, *log_bin_total = nullptr;
    uint32_t seen_log_cap = 0, log_nbins = 0;
    unsigned long long *log_host_total = nullptr; // mapped host memory: [0] log places per 1024 reads as of the last pass, [+8] "log off", written by the device
    bool log_dirty = false;            // something may have been logged since the last pass
    bool log_off = false;              // a pass has found this sample's reads to hit so often that atomics from the resolver are cheaper
    uint32_t passes_done = 0;
    uint32_t launches_since_apply = 0;
    uint64_t reads_since_apply = 0;
    double log_entries_per_read = 4.0; // pace of the passes: a guess until the first pass has reported
    uint64_t reads_of_last_pass = 0;
    // very long records: one word per k-mer position for the hits
    uint32_t *long_hits = nullptr;
    uint64_t long_hits_cap = 0;
    uint8_t *long_tiles = nullptr; // "this tile of 256 positions holds a hit"
    uint64_t long_tiles_cap = 0;
    uint64_t reads_submitted = 0; // since the last reset: checked against the device's count when results are read
    // the scratch below is one set per sample: batches on different streams are ordered behind each other
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    hipEvent_t order_ev = nullptr;
    // Staging for the host-buffer entry points: a ring of slots so that the upload of batch b + 1 (copy stream) and
    // the download of batch b - 1's results (result stream) run beside the kernels of batch b (the sample's stream).
    struct Slot {
        uint8_t *bases = nullptr;
        uint64_t bases_cap = 0;
        uint64_t *offsets = nullptr;
        int32_t *start = nullptr, *stop = nullptr;
        uint32_t *out = nullptr;
        uint64_t reads_cap = 0;
        KidFastqRec *recs = nullptr; // kid_classify_fastq_async: where the host found the lines of the block's records
        uint64_t recs_cap = 0;
        hipEvent_t ev_h2d = nullptr, ev_done = nullptr, ev_out = nullptr;
        std::vector<uint64_t> rel; // offsets rebased to the slot (alive until the copy has been issued AND done)
        uint64_t ticket = 0;
        bool busy = false;
    };
    static const int NSLOT = 3;
    Slot slots[NSLOT];
    hipStream_t copy_stream = nullptr, out_stream = nullptr;
    uint64_t next_ticket = 1;
};

extern "C" const char *kid_strerror(int status)
{
    switch (status) {
    case KID_OK: return "ok";
    case KID_ERR_ARG: return "bad argument";
    case KID_ERR_NOMEM: return "out of memory";
    case KID_ERR_HIP: return "HIP runtime error";
    case KID_ERR_TABLE_FULL: return "out of memory in table";
    case KID_ERR_TREE: return "taxonomy parent[] is out of range or cyclic";
    case KID_ERR_NO_DEVICE: return "no HIP device";
    case KID_ERR_TARGET: return "target id outside [0, ntar)";
    case KID_ERR_IO: return "I/O error";
    case KID_ERR_FORMAT: return "malformed input";
    case KID_ERR_STATE: return "call sequence error";
    default: return "unknown status";
    }
}

extern "C" const char *kid_last_error(void) { return g_last_error.c_str(); }

extern "C" int kid_device_count(int *count)
{
    if (!count) return kid_fail(KID_ERR_ARG, "count is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return kid_fail(KID_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return KID_OK;
}

static int kid_use_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return kid_fail(KID_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                        e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return kid_fail(KID_ERR_ARG, "device %d out of range [0,%d)", device, n);
    KID_HIP(hipSetDevice(device));
    return KID_OK;
}

static inline int kid_grid_for(uint64_t n, int block, int cap_blocks)
{
    uint64_t g = (n + (uint64_t)block - 1) / (uint64_t)block;
    if (g < 1) g = 1;
    if (g > (uint64_t)cap_blocks) g = (uint64_t)cap_blocks;
    return (int)g;
}

// ---------------------------------------------------------------- taxonomy preparation
// effective parent = Tree1::get_parent (newkmer_10nx.cpp:146-152): nodes 0 and 1 answer root.
static int kid_prepare_tree(const int32_t *parent, int32_t ntar, std::vector<int32_t> &par, std::vector<int32_t> &depth,
                            int &max_depth)
{
    par.assign((size_t)ntar, 1);
    depth.assign((size_t)ntar, -1);
    for (int32_t i = 0; i < ntar; i++) {
        int32_t p = (i != 1 && i > 0) ? parent[i] : 1;
        if (p < 0 || p >= ntar) return kid_fail(KID_ERR_TREE, "parent[%d] = %d is outside [0,%d)", i, p, ntar);
        par[(size_t)i] = p;
    }
    depth[1] = 0;
    max_depth = 0;
    std::vector<int32_t> stack;
    for (int32_t i = 0; i < ntar; i++) {
        if (depth[(size_t)i] >= 0) continue;
        stack.clear();
        int32_t z = i;
        while (depth[(size_t)z] < 0) {
            if ((int32_t)stack.size() > ntar) return kid_fail(KID_ERR_TREE, "cycle in parent[] reachable from node %d", i);
            depth[(size_t)z] = -2; // on stack
            stack.push_back(z);
            z = par[(size_t)z];
            if (depth[(size_t)z] == -2) return kid_fail(KID_ERR_TREE, "cycle in parent[] reachable from node %d", i);
        }
        int32_t d = depth[(size_t)z];
        for (size_t j = stack.size(); j-- > 0;) depth[(size_t)stack[j]] = ++d;
    }
    for (int32_t i = 0; i < ntar; i++) max_depth = depth[(size_t)i] > max_depth ? depth[(size_t)i] : max_depth;
    return KID_OK;
}

static void kid_make_rows(const std::vector<int32_t> &par, const std::vector<int32_t> &depth, std::vector<uint4> &rows)
{
    const size_t ntar = par.size();
    rows.assign(ntar, make_uint4(0, 0, 0, 0));
    for (size_t i = 0; i < ntar; i++) {
        uint16_t e[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int32_t d = depth[i];
        e[0] = (uint16_t)d;
        int32_t z = (int32_t)i;
        while (d >= 1) {
            if (d <= 7) e[d] = (uint16_t)z;
            z = par[(size_t)z];
            d--;
        }
        rows[i] = make_uint4((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                             (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16));
    }
}

// ---------------------------------------------------------------- host table builder
// Hashtable::add_kmer replayed in file order into 16-byte cells: the exact cell
// geometry of the reference, needed when lookups are probe-capped (kmer_read_m3).
static int kid_host_build(const uint64_t *keys, const uint32_t *targets, uint64_t n, int log2_slots, uint4 *cells,
                          uint64_t *n_occupied)
{
    const uint64_t nslots = 1ULL << log2_slots, mask = nslots - 1;
    uint64_t size = 0, occ = 0;
    for (uint64_t e = 0; e < n; e++) {
        const uint64_t key = keys[e], hash = kid_fmix64(key);
        uint64_t reprobe = 0, i = 0;
        for (;;) {
            const uint64_t idx = (hash + reprobe) & mask;
            reprobe += ++i;
            if (cells[idx].z == 0) {
                cells[idx].x = (uint32_t)key;
                cells[idx].y = (uint32_t)(key >> 32);
                cells[idx].z = targets[e];
                cells[idx].w = (uint32_t)e + 1u;
                if (targets[e] != 0) occ++;
                if (++size > nslots - 32) return kid_fail(KID_ERR_TABLE_FULL, "out of memory in table");
                break;
            }
        }
    }
    *n_occupied = occ;
    return KID_OK;
}

static int kid_db_build_common(const uint64_t *h_keys, const uint32_t *h_targets, const void *d_keys_in,
                               const void *d_targets_in, uint64_t n, const int32_t *parent, int32_t ntar, int k,
                               int log2_slots, int max_probes, uint32_t flags, int device, kid_db **out)
{
    if (!out) return kid_fail(KID_ERR_ARG, "out is null");
    *out = nullptr;
    if (!parent || ntar < 2) return kid_fail(KID_ERR_ARG, "parent is null or ntar < 2");
    if (k < 1 || k > 31) return kid_fail(KID_ERR_ARG, "k = %d outside [1,31]", k);
    if (log2_slots < 6 || log2_slots > 32) return kid_fail(KID_ERR_ARG, "log2_slots = %d outside [6,32]", log2_slots);
    if (max_probes < 0) return kid_fail(KID_ERR_ARG, "max_probes < 0");
    if (n > 0 && !((h_keys && h_targets) || (d_keys_in && d_targets_in))) return kid_fail(KID_ERR_ARG, "keys/targets null");
    if (n >= 0xFFFFFFFFull) return kid_fail(KID_ERR_ARG, "more than 2^32-2 entries");
    const uint64_t nslots = 1ULL << log2_slots;
    if (n > nslots - 32) return kid_fail(KID_ERR_TABLE_FULL, "out of memory in table");
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;

    std::vector<int32_t> par, depth;
    int max_depth = 0;
    rc = kid_prepare_tree(parent, ntar, par, depth, max_depth);
    if (rc != KID_OK) return rc;
    if (h_targets)
        for (uint64_t i = 0; i < n; i++)
            if (h_targets[i] >= (uint32_t)ntar) return kid_fail(KID_ERR_TARGET, "targets[%llu] = %u >= ntar", (unsigned long long)i, h_targets[i]);

    kid_db *db = new kid_db();
    db->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) db->num_cu = prop.multiProcessorCount;
    if (db->num_cu <= 0) db->num_cu = 256;

#define KID_DB_HIP(call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            kid_db_destroy(db);                                                                  \
            return kid_fail(e_ == hipErrorOutOfMemory ? KID_ERR_NOMEM : KID_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
        }                                                                                        \
    } while (0)

    const uint64_t table_bytes = nslots * sizeof(uint4);
    KID_DB_HIP(hipMalloc(&db->table, table_bytes));
    // the per-sample seen-bitmap has one bit per ENTRY (its insertion ordinal, cell word 3), not per cell: a key's bit
    // is then the same in every table built from the same entries, whatever the cell placement -- what lets samples of
    // different GPUs (each with its own replica of the table) be OR-ed.  ord_target maps a bit back to its target.
    db->seen_bits = ((n + 127) / 128) * 128;
    if (db->seen_bits == 0) db->seen_bits = 128;
    KID_DB_HIP(hipMalloc(&db->ord_target, db->seen_bits * 4));
    KID_DB_HIP(hipMemset(db->ord_target, 0, db->seen_bits * 4));
    if (n > 0) {
        if (h_targets) KID_DB_HIP(hipMemcpy(db->ord_target, h_targets, n * 4, hipMemcpyHostToDevice));
        else KID_DB_HIP(hipMemcpy(db->ord_target, d_targets_in, n * 4, hipMemcpyDeviceToDevice));
    }
    KID_DB_HIP(hipMalloc(&db->parent, sizeof(int32_t) * (size_t)ntar));
    KID_DB_HIP(hipMalloc(&db->depth, sizeof(int32_t) * (size_t)ntar));
    KID_DB_HIP(hipMemcpy(db->parent, par.data(), sizeof(int32_t) * (size_t)ntar, hipMemcpyHostToDevice));
    KID_DB_HIP(hipMemcpy(db->depth, depth.data(), sizeof(int32_t) * (size_t)ntar, hipMemcpyHostToDevice));
    const bool rows_ok = (max_depth <= 8 && ntar <= 65536);
    if (rows_ok) {
        std::vector<uint4> rows;
        kid_make_rows(par, depth, rows);
        KID_DB_HIP(hipMalloc(&db->rows, sizeof(uint4) * (size_t)ntar));
        KID_DB_HIP(hipMemcpy(db->rows, rows.data(), sizeof(uint4) * (size_t)ntar, hipMemcpyHostToDevice));
    }

    uint64_t n_occupied = 0;
    const bool host_build = (max_probes > 0) || (flags & KID_FLAG_HOST_BUILD);
    // minimizer-localised placement needs an unbounded probe loop (results must not depend on the
    // cell geometry) and k >= 24 (minimizers of k - 14 >= 10 bases)
    // (7 of 8 cells hold entries, and chains need free lines: at most 80 % of the cells may be taken)
    const uint32_t minloc = (!host_build && !(flags & KID_FLAG_REF_GEOMETRY) && k >= 24 && n <= (nslots / 10) * 8) ? 1u : 0u;
    const uint32_t line_bits = (uint32_t)log2_slots - 3u;
    const uint32_t line_shift = 32u - line_bits, line_mask = (uint32_t)((nslots >> 3) - 1);
    if (host_build) {
        std::vector<uint64_t> hk;
        std::vector<uint32_t> ht;
        if (!h_keys && n > 0) { // entries live on the device: fetch them
            hk.resize(n); ht.resize(n);
            KID_DB_HIP(hipMemcpy(hk.data(), d_keys_in, n * 8, hipMemcpyDeviceToHost));
            KID_DB_HIP(hipMemcpy(ht.data(), d_targets_in, n * 4, hipMemcpyDeviceToHost));
            h_keys = hk.data(); h_targets = ht.data();
            for (uint64_t i = 0; i < n; i++)
                if (h_targets[i] >= (uint32_t)ntar) { kid_db_destroy(db); return kid_fail(KID_ERR_TARGET, "targets[%llu] >= ntar", (unsigned long long)i); }
        }
        uint4 *cells = (uint4 *)calloc(nslots, sizeof(uint4));
        if (!cells) { kid_db_destroy(db); return kid_fail(KID_ERR_NOMEM, "host table of %llu bytes", (unsigned long long)table_bytes); }
        rc = kid_host_build(h_keys, h_targets, n, log2_slots, cells, &n_occupied);
        if (rc != KID_OK) { free(cells); kid_db_destroy(db); return rc; }
        hipError_t e = hipMemcpy(db->table, cells, table_bytes, hipMemcpyHostToDevice);
        free(cells);
        KID_DB_HIP(e);
    } else {
        KID_DB_HIP(hipMemset(db->table, 0, table_bytes));
        if (n > 0) {
            uint64_t *dk = nullptr;
            const uint64_t *dkc = (const uint64_t *)d_keys_in;
            const uint32_t *dtc = db->ord_target;
            if (!dkc) {
                KID_DB_HIP(hipMalloc(&dk, n * 8));
                hipError_t e = hipMemcpy(dk, h_keys, n * 8, hipMemcpyHostToDevice);
                if (e != hipSuccess) { hipFree(dk); KID_DB_HIP(e); }
                dkc = dk;
            }
            unsigned long long *d_occ = nullptr;
            {
                hipError_t e = hipMalloc(&d_occ, 16);
                if (e == hipSuccess) e = hipMemset(d_occ, 0, 16);
                if (e != hipSuccess) { if (d_occ) hipFree(d_occ); if (dk) hipFree(dk); KID_DB_HIP(e); }
            }
            const int grid = kid_grid_for(n, 256, db->num_cu * 16);
            hipLaunchKernelGGL(kid_build_insert_kernel, dim3(grid), dim3(256), 0, 0, db->table, (uint32_t)(nslots - 1), dkc,
                               dtc, n, (uint32_t)ntar, d_occ, k, minloc, line_shift, line_mask);
            hipLaunchKernelGGL(kid_build_firstwins_kernel, dim3(grid), dim3(256), 0, 0, db->table, (uint32_t)(nslots - 1),
                               dkc, dtc, n, k, minloc, line_shift, line_mask);
            hipError_t e = hipDeviceSynchronize();
            unsigned long long occ[2] = {0, 0};
            if (e == hipSuccess) e = hipMemcpy(occ, d_occ, 16, hipMemcpyDeviceToHost);
            hipFree(d_occ);
            if (dk) hipFree(dk);
            KID_DB_HIP(e);
            if (occ[1] != 0) { kid_db_destroy(db); return kid_fail(KID_ERR_TARGET, "%llu targets >= ntar", occ[1]); }
            n_occupied = occ[0];
        }
    }
#undef KID_DB_HIP

    db->d.table = db->table;
    db->d.nslots = nslots;
    db->d.slot_mask = (uint32_t)(nslots - 1);
    db->d.max_probes = (uint32_t)max_probes;
    db->d.k = k;
    db->d.u_is_t = (flags & KID_FLAG_U_IS_T) ? 1u : 0u;
    db->d.minloc = minloc;
    db->d.line_shift = line_shift;
    db->d.line_mask = line_mask;
    db->info.geometry = (int32_t)minloc;
    db->d.rows = db->rows;
    db->d.parent = db->parent;
    db->d.depth = db->depth;
    db->d.ntar = ntar;
    db->info.ntar = ntar;
    db->info.k = k;
    db->info.log2_slots = log2_slots;
    db->info.max_probes = max_probes;
    db->info.flags = flags;
    db->info.device = device;
    db->info.tree_depth = max_depth;
    db->info.host_built = host_build ? 1 : 0;
    db->info.n_entries = n;
    db->info.n_occupied = n_occupied;
    db->info.table_bytes = table_bytes;
    *out = db;
    return KID_OK;
}

extern "C" int kid_db_build(const uint64_t *keys, const uint32_t *targets, uint64_t n, const int32_t *parent, int32_t ntar,
                            int k, int log2_slots, int max_probes, uint32_t flags, int device, kid_db **out)
{
    return kid_db_build_common(keys, targets, nullptr, nullptr, n, parent, ntar, k, log2_slots, max_probes, flags, device, out);
}

extern "C" int kid_db_build_device(const void *d_keys, const void *d_targets, uint64_t n, const int32_t *parent,
                                   int32_t ntar, int k, int log2_slots, int max_probes, uint32_t flags, int device,
                                   kid_db **out)
{
    return kid_db_build_common(nullptr, nullptr, d_keys, d_targets, n, parent, ntar, k, log2_slots, max_probes, flags, device, out);
}

// A replica of a database on another GPU (or on the same one): device-to-device copies of the table (16 GiB at bact10
// scale: over xGMI between peers), the taxonomy arrays and the entry -> target map.  Entry ordinals are part of the
// cells, so the replicas' samples share one seen-bitmap numbering (kid_sample_end_merged).
extern "C" int kid_db_replicate(const kid_db *src, int device, kid_db **out)
{
    if (!src || !out) return kid_fail(KID_ERR_ARG, "null argument");
    *out = nullptr;
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    kid_db *db = new kid_db();
    db->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) db->num_cu = prop.multiProcessorCount;
    if (db->num_cu <= 0) db->num_cu = 256;
    const size_t nt = (size_t)src->info.ntar;
#define KID_R_HIP(call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            kid_db_destroy(db);                                                                  \
            return kid_fail(e_ == hipErrorOutOfMemory ? KID_ERR_NOMEM : KID_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
        }                                                                                        \
    } while (0)
    auto copy = [&](void *dst, const void *from, size_t nbytes) -> hipError_t {
        if (device == src->device) return hipMemcpy(dst, from, nbytes, hipMemcpyDeviceToDevice);
        return hipMemcpyPeer(dst, device, from, src->device, nbytes);
    };
    KID_R_HIP(hipMalloc(&db->table, src->info.table_bytes));
    KID_R_HIP(copy(db->table, src->table, src->info.table_bytes));
    KID_R_HIP(hipMalloc(&db->parent, sizeof(int32_t) * nt));
    KID_R_HIP(copy(db->parent, src->parent, sizeof(int32_t) * nt));
    KID_R_HIP(hipMalloc(&db->depth, sizeof(int32_t) * nt));
    KID_R_HIP(copy(db->depth, src->depth, sizeof(int32_t) * nt));
    if (src->rows) {
        KID_R_HIP(hipMalloc(&db->rows, sizeof(uint4) * nt));
        KID_R_HIP(copy(db->rows, src->rows, sizeof(uint4) * nt));
    }
    db->seen_bits = src->seen_bits;
    KID_R_HIP(hipMalloc(&db->ord_target, db->seen_bits * 4));
    KID_R_HIP(copy(db->ord_target, src->ord_target, db->seen_bits * 4));
    KID_R_HIP(hipDeviceSynchronize());
#undef KID_R_HIP
    db->d = src->d;
    db->d.table = db->table;
    db->d.rows = db->rows;
    db->d.parent = db->parent;
    db->d.depth = db->depth;
    db->info = src->info;
    db->info.device = device;
    *out = db;
    return KID_OK;
}

extern "C" int kid_db_get_info(const kid_db *db, kid_db_info *out)
{
    if (!db || !out) return kid_fail(KID_ERR_ARG, "null argument");
    *out = db->info;
    return KID_OK;
}

extern "C" void kid_db_destroy(kid_db *db)
{
    if (!db) return;
    hipSetDevice(db->device);
    if (db->table) hipFree(db->table);
    if (db->rows) hipFree(db->rows);
    if (db->parent) hipFree(db->parent);
    if (db->depth) hipFree(db->depth);
    if (db->ord_target) hipFree(db->ord_target);
    delete db;
}

extern "C" int kid_db_lookup(kid_db *db, const uint64_t *keys, uint64_t n, uint32_t *targets, uint32_t *probes)
{
    if (!db || (n && (!keys || !targets))) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(db->device);
    if (rc != KID_OK) return rc;
    if (n == 0) return KID_OK;
    KidDevBuf dk, dt, dp;
    KID_HIP(dk.alloc(n * 8));
    KID_HIP(dt.alloc(n * 4));
    if (probes) KID_HIP(dp.alloc(n * 4));
    KID_HIP(hipMemcpy(dk.p, keys, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kid_lookup_kernel, dim3(kid_grid_for(n, 256, db->num_cu * 16)), dim3(256), 0, 0, db->d, dk.as<uint64_t>(), n,
                       dt.as<uint32_t>(), dp.as<uint32_t>());
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(targets, dt.p, n * 4, hipMemcpyDeviceToHost));
    if (probes) KID_HIP(hipMemcpy(probes, dp.p, n * 4, hipMemcpyDeviceToHost));
    return KID_OK;
}

extern "C" int kid_hash_keys(int device, const uint64_t *keys, uint64_t n, uint64_t *out)
{
    if (n && (!keys || !out)) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    if (n == 0) return KID_OK;
    KidDevBuf dk, dout;
    KID_HIP(dk.alloc(n * 8));
    KID_HIP(dout.alloc(n * 8));
    KID_HIP(hipMemcpy(dk.p, keys, n * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kid_fmix_kernel, dim3(kid_grid_for(n, 256, 4096)), dim3(256), 0, 0, dk.as<uint64_t>(), n, dout.as<uint64_t>());
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(out, dout.p, n * 8, hipMemcpyDeviceToHost));
    return KID_OK;
}

extern "C" int kid_db_msca(kid_db *db, const int32_t *x, const int32_t *y, uint64_t n, int32_t *out)
{
    if (!db || (n && (!x || !y || !out))) return kid_fail(KID_ERR_ARG, "null argument");
    for (uint64_t i = 0; i < n; i++)
        if (x[i] < 0 || x[i] >= db->info.ntar || y[i] < 0 || y[i] >= db->info.ntar)
            return kid_fail(KID_ERR_TARGET, "pair %llu outside [0,ntar)", (unsigned long long)i);
    int rc = kid_use_device(db->device);
    if (rc != KID_OK) return rc;
    if (n == 0) return KID_OK;
    KidDevBuf dx, dy, dout;
    KID_HIP(dx.alloc(n * 4));
    KID_HIP(dy.alloc(n * 4));
    KID_HIP(dout.alloc(n * 4));
    KID_HIP(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice));
    KID_HIP(hipMemcpy(dy.p, y, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kid_msca_kernel, dim3(kid_grid_for(n, 256, db->num_cu * 16)), dim3(256), 0, 0, db->d, dx.as<int32_t>(),
                       dy.as<int32_t>(), n, dout.as<int32_t>());
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(out, dout.p, n * 4, hipMemcpyDeviceToHost));
    return KID_OK;
}

// ---------------------------------------------------------------- sample
extern "C" void kid_sample_destroy(kid_sample *s)
{
    if (!s) return;
    if (s->db) hipSetDevice(s->db->device);
    if (s->gcount) hipFree(s->gcount);
    if (s->ucount) hipFree(s->ucount);
    if (s->stats) hipFree(s->stats);
    if (s->seen) hipFree(s->seen);
    for (auto &ev : s->timed) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); }
    for (kid_sample::Scratch &sc : s->sets) {
        if (sc.rare) hipFree(sc.rare);
        if (sc.desc) hipFree(sc.desc);
        if (sc.long_list) hipFree(sc.long_list);
        if (sc.long_plan) hipFree(sc.long_plan);
        if (sc.ev_prep) hipEventDestroy(sc.ev_prep);
        if (sc.ev_used) hipEventDestroy(sc.ev_used);
    }
    if (s->prep_stream) hipStreamDestroy(s->prep_stream);
    if (s->long_hits) hipFree(s->long_hits);
    if (s->long_tiles) hipFree(s->long_tiles);
    if (s->rare_fixed) hipFree(s->rare_fixed);
    if (s->seen_log) hipFree(s->seen_log);
    if (s->seen_log_tail) hipFree(s->seen_log_tail);
    if (s->seen_sorted) hipFree(s->seen_sorted);
    if (s->log_counts) hipFree(s->log_counts);
    if (s->log_bin_total) hipFree(s->log_bin_total);
    if (s->log_host_total) hipHostFree(s->log_host_total);
    if (s->order_ev) hipEventDestroy(s->order_ev);
    hipDeviceSynchronize();
    for (kid_sample::Slot &sl : s->slots) {
        if (sl.bases) hipFree(sl.bases);
        if (sl.offsets) hipFree(sl.offsets);
        if (sl.start) hipFree(sl.start);
        if (sl.stop) hipFree(sl.stop);
        if (sl.out) hipFree(sl.out);
        if (sl.recs) hipFree(sl.recs);
        if (sl.ev_h2d) hipEventDestroy(sl.ev_h2d);
        if (sl.ev_done) hipEventDestroy(sl.ev_done);
        if (sl.ev_out) hipEventDestroy(sl.ev_out);
    }
    if (s->copy_stream) hipStreamDestroy(s->copy_stream);
    if (s->out_stream) hipStreamDestroy(s->out_stream);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

static int kid_seenlog_point(kid_sample *s, uint32_t *log, hipStream_t stream);
extern "C" int kid_sample_reset(kid_sample *s)
{
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    const size_t nt = (size_t)s->db->info.ntar;
    KID_HIP(hipMemset(s->gcount, 0, nt * 8));
    KID_HIP(hipMemset(s->ucount, 0, nt * 8));
    KID_HIP(hipMemset(s->stats, 0, 256));
    {   // device-clock stamps of a launch: [30] first workgroup start (min), [31] last end (max); see kid_classify_kernel
        const unsigned long long never = ~0ull;
        KID_HIP(hipMemcpy(s->stats + 30, &never, 8, hipMemcpyHostToDevice));
    }
    s->dev_clock_batches = 0;
    s->reads_submitted = 0;
    if (s->seen_log_tail) KID_HIP(hipMemset(s->seen_log_tail, 0, KID_LOG_SHARDS * 64));
    if (s->seen_log) { // (a pass may have taken the log out of the argument blocks: KidLogArgs)
        int rc = kid_seenlog_point(s, s->seen_log, nullptr);
        if (rc != KID_OK) return rc;
        *(volatile unsigned int *)((char *)s->log_host_total + 8) = 0;
        *(volatile unsigned long long *)s->log_host_total = 0;
        s->log_off = false;
        s->passes_done = 0;
        s->log_entries_per_read = 4.0;
    }
    s->log_dirty = false;
    s->launches_since_apply = 0;
    s->reads_since_apply = 0;
    KID_HIP(hipMemset(s->seen, 0, s->seen_words * 4));
    KID_HIP(hipDeviceSynchronize());
    return KID_OK;
}

extern "C" int kid_sample_begin(kid_db *db, kid_sample **out)
{
    if (!db || !out) return kid_fail(KID_ERR_ARG, "null argument");
    *out = nullptr;
    int rc = kid_use_device(db->device);
    if (rc != KID_OK) return rc;
    kid_sample *s = new kid_sample();
    s->db = db;
    const size_t nt = (size_t)db->info.ntar;
    s->seen_words = db->seen_bits / 32; // one bit per DB entry, whole 16-byte groups
#define KID_S_HIP(call)                                                                                           \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            kid_sample_destroy(s);                                                                                \
            return kid_fail(e_ == hipErrorOutOfMemory ? KID_ERR_NOMEM : KID_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
        }                                                                                                         \
    } while (0)
    KID_S_HIP(hipMalloc(&s->gcount, nt * 8));
    KID_S_HIP(hipMalloc(&s->ucount, nt * 8));
    KID_S_HIP(hipMalloc(&s->stats, 256)); // [0..7] counters, [8..31] KID_PROFILE phase cycles
    KID_S_HIP(hipMalloc(&s->seen, s->seen_words * 4));
    KID_S_HIP(hipStreamCreate(&s->stream));
    {
        // the hit log: for the minimizer-localised table (its resolver is the one that logs), bitmaps of up to 1024 pieces
        const uint64_t nbins = (db->seen_bits + (1ull << KID_LOG_BIN_BITS) - 1) >> KID_LOG_BIN_BITS;
        if (db->d.minloc && nbins <= 1024) {
            uint64_t total = db->info.n_entries * 2ull;              // the whole log holds 2 x the entries of the database ...
            if (total < (32ull << 20)) total = 32ull << 20;          // ... at least 32 M (a launch of 1 M pairs with 16 hits per read) ...
            if (total > (128ull << 20)) total = 128ull << 20;        // ... at most 128 M hits = 512 MiB (+ as much to sort them)
            uint64_t cap = (total / KID_LOG_SHARDS) & ~63ull;        // per region
            s->seen_log_cap = (uint32_t)cap;
            s->log_nbins = (uint32_t)nbins;
            KID_S_HIP(hipMalloc(&s->seen_log, cap * KID_LOG_SHARDS * 4));
            KID_S_HIP(hipMalloc(&s->seen_sorted, cap * KID_LOG_SHARDS * 4));
            KID_S_HIP(hipMalloc(&s->seen_log_tail, KID_LOG_SHARDS * 64));
            KID_S_HIP(hipMalloc(&s->log_counts, nbins * KID_LOG_WGS * 4));
            KID_S_HIP(hipMalloc(&s->log_bin_total, nbins * 4));
            KID_S_HIP(hipHostMalloc((void **)&s->log_host_total, 64, hipHostMallocMapped));
            memset((void *)s->log_host_total, 0, 64);
        }
        const KidRareArgs ra{s->gcount, s->stats, db->d.line_mask, 0u, 0ull, 0ull, 0, 0u, db->rows, s->seen, nullptr, nullptr,
                             s->seen_log, s->seen_log_tail, s->seen_log_cap, 0u};
        for (kid_sample::Scratch &sc : s->sets) {
            KID_S_HIP(hipMalloc(&sc.rare, sizeof(ra)));
            KID_S_HIP(hipMemcpy(sc.rare, &ra, sizeof(ra), hipMemcpyHostToDevice));
            KID_S_HIP(hipEventCreateWithFlags(&sc.ev_prep, hipEventDisableTiming));
            KID_S_HIP(hipEventCreateWithFlags(&sc.ev_used, hipEventDisableTiming));
        }
        KID_S_HIP(hipMalloc(&s->rare_fixed, sizeof(ra)));
        KID_S_HIP(hipMemcpy(s->rare_fixed, &ra, sizeof(ra), hipMemcpyHostToDevice));
    }
#undef KID_S_HIP
    rc = kid_sample_reset(s);
    if (rc != KID_OK) { kid_sample_destroy(s); return rc; }
    *out = s;
    return KID_OK;
}

// One batch on `stream`: kid_prepare_kernel (read descriptors, range checks, longest read -- not for fixed-layout
// batches, whose reads need no descriptors) and the instantiation(s) of kid_classify_kernel (512-thread workgroups =
// 8 waves; pair loop / duo loop / general loops -- the ones the batch is not for return at once; the gcount histogram
// lives in LDS when 4 workgroups per CU still fit).  The kernels read the caller's ASCII text directly.
// max_kmers: the largest n_kmers of the batch when the host knows it (then only the kernel the batch is for is
// launched), -1 when only the device does
// long_records: the batch may hold records of more than s->long_kmers k-mers (FASTA contigs): those take the
// long-record kernels; the host does not need to know which they are
// prep_stream: where the prepare kernel runs.  The same as `stream` unless the read text is known to be ready earlier
// than stream order says (host path: the copy stream behind the upload; kid_classify_batch_device under
// KID_OPT_INPUTS_READY: an internal stream) -- then it overlaps with the classify kernels of the batch before.
// The hit log -> bits of `seen` (kid_seenlog_* kernels), on `stream`, behind everything queued there.
static int kid_seenlog_apply(kid_sample *s, hipStream_t stream)
{
    if (!s->seen_log || !s->log_dirty) return KID_OK;
    void *dev_total = nullptr;
    KID_HIP(hipHostGetDevicePointer(&dev_total, s->log_host_total, 0));
    const KidLogArgs a{s->seen_log, s->seen_log_tail, s->seen_log_cap, s->log_nbins, s->log_counts, s->log_bin_total, s->seen_sorted,
                       s->seen, s->seen_words, (unsigned long long *)dev_total, s->reads_since_apply,
                       {s->sets[0].rare, s->sets[1].rare, s->sets[2].rare, s->rare_fixed}, (unsigned int *)((char *)dev_total + 8)};
    static_assert(kid_sample::NSET == 3, "KidLogArgs::blocks");
    const uint32_t nb = s->log_nbins;
    hipLaunchKernelGGL(kid_seenlog_count_kernel, dim3(KID_LOG_WGS), dim3(256), nb * 4, stream, a);
    hipLaunchKernelGGL(kid_seenlog_scan_kernel, dim3(nb), dim3(KID_LOG_WGS), 0, stream, a);
    hipLaunchKernelGGL(kid_seenlog_scatter_kernel, dim3(KID_LOG_WGS), dim3(256), (4 * nb + 1 + KID_LOG_TILE) * 4, stream, a);
    hipLaunchKernelGGL(kid_seenlog_apply_kernel, dim3(nb), dim3(1024), ((1u << (KID_LOG_BIN_BITS - 5)) + nb + 1) * 4, stream, a);
    KID_HIP(hipMemsetAsync(s->seen_log_tail, 0, KID_LOG_SHARDS * 64, stream));
    KID_HIP(hipGetLastError());
    s->reads_of_last_pass = s->reads_since_apply;
    s->log_dirty = false;
    s->launches_since_apply = 0;
    s->reads_since_apply = 0;
    s->passes_done++;
    return KID_OK;
}
// ... when somebody wants to read the bitmap: behind everything the sample has queued anywhere (the caller synchronises after it)
static int kid_seenlog_flush(kid_sample *s)
{
    if (!s->seen_log || !s->log_dirty) return KID_OK;
    KID_HIP(hipDeviceSynchronize());
    return kid_seenlog_apply(s, s->stream);
}
// the log pointer of every argument block of the sample (null: the resolvers set the bits with atomics), in stream order
static int kid_seenlog_point(kid_sample *s, uint32_t *log, hipStream_t stream)
{
    KidRareArgs *blocks[kid_sample::NSET + 1];
    int nb = 0;
    for (kid_sample::Scratch &sc : s->sets) blocks[nb++] = sc.rare;
    blocks[nb++] = s->rare_fixed;
    for (int i = 0; i < nb; i++)
        if (blocks[i]) {
            uint32_t **where = &blocks[i]->seen_log;
            if (log) KID_HIP(hipMemcpyAsync(where, &s->seen_log, sizeof(uint32_t *), hipMemcpyHostToDevice, stream));
            else KID_HIP(hipMemsetAsync(where, 0, sizeof(uint32_t *), stream));
        }
    return KID_OK;
}
// before a launch of n_reads reads: run the pass if the log might not hold what the launch adds.  The device reports
// the entries of every pass (mapped host memory, read without waiting: whatever pass has finished by now), from which
// the hits per read of this sample are known; a region that does fill up falls back to atomics, so the pace only
// matters for speed.
static int kid_seenlog_pace(kid_sample *s, uint64_t n_reads, hipStream_t stream)
{
    if (!s->seen_log || s->log_off) return KID_OK;
    if (*(volatile unsigned int *)((char *)s->log_host_total + 8)) { // a pass found more than 8 hits per read and took the log away (KidLogArgs)
        s->log_off = true;
        return KID_OK;
    }
    const unsigned long long rate = *(volatile unsigned long long *)s->log_host_total; // places per 1024 reads | 1 << 63, from the latest pass that has run
    if (rate >> 63) {
        const double r = (double)(rate & ~(1ull << 63)) / 1024.0;
        s->log_entries_per_read = r > 0.01 ? r * 1.1 : 0.011;
    }
    // (A first pass right behind a sample's first launch would tell early what kind of sample it is -- and made every
    // later launch of the metric's workload 3 % slower, profiles/r03/ab_early_pass.txt; the first regular pass comes
    // after 4 M reads.)
    const double room = 0.5 * (double)s->seen_log_cap * KID_LOG_SHARDS;
    if (s->log_dirty && ((double)(s->reads_since_apply + n_reads) * s->log_entries_per_read > room || s->launches_since_apply >= 256)) {
        int rc = kid_seenlog_apply(s, stream);
        if (rc != KID_OK) return rc;
    }
    s->reads_since_apply += n_reads;
    s->launches_since_apply++;
    s->log_dirty = true;
    return KID_OK;
}

// fastq: the batch is a block of FASTQ text with the host's line index (kid_classify_fastq_async); b.bases = the text,
// b.start / b.stop = device arrays that RECEIVE what process_qual computes
struct KidFastqIn {
    const KidFastqRec *recs;
};
static int kid_launch_classify(kid_sample *s, const KidBatch &b, uint64_t bases_nbytes, hipStream_t stream, int64_t max_kmers,
                               hipStream_t prep_stream, bool long_records = false, const KidFastqIn *fastq = nullptr)
{
    kid_db *db = s->db;
    if (b.n == 0) return KID_OK;
    if (b.n > 0x7FFFFFFFull) return kid_fail(KID_ERR_ARG, "at most 2^31-1 reads per batch");
    if (bases_nbytes >> 48) return kid_fail(KID_ERR_ARG, "a batch of 2^48 bytes or more");
    const bool fixed = b.offsets == nullptr && !fastq; // fixed layout: whole reads of b.fixed_len bases back to back
    const uint32_t long_cut = (long_records && !fastq && s->long_kmers > 0 && s->long_kmers < 0xFFFFFFFFll &&
                               bases_nbytes > (uint64_t)s->long_kmers) ? (uint32_t)s->long_kmers : 0u;
    if (long_cut) max_kmers = -1; // (the host's number counts the hidden records too: the device's does not)
    // The classify kernels of a sample's batches run one after the other (they share the sample's counters' timing
    // stamps and argument blocks): a batch issued on another stream than the one before is made to wait for it.
    if (s->has_last_stream && s->last_stream != stream) {
        if (!s->order_ev) KID_HIP(hipEventCreateWithFlags(&s->order_ev, hipEventDisableTiming));
        KID_HIP(hipEventRecord(s->order_ev, s->last_stream));
        KID_HIP(hipStreamWaitEvent(stream, s->order_ev, 0));
    }
    s->last_stream = stream;
    s->has_last_stream = true;
    {
        int rc = kid_seenlog_pace(s, b.n, stream);
        if (rc != KID_OK) return rc;
    }
    kid_sample::Scratch *scp = nullptr;
    KidRareArgs *rare = s->rare_fixed;
    if (!fixed) {
        kid_sample::Scratch &sc = s->sets[s->next_set++ % kid_sample::NSET];
        scp = &sc;
        rare = sc.rare;
        if (b.n > sc.desc_cap) KID_HIP(hipDeviceSynchronize()); // (scratch in use is not freed)
        if (b.n > sc.desc_cap) {
            if (sc.desc) hipFree(sc.desc);
            sc.desc = nullptr; sc.desc_cap = 0;
            KID_HIP(hipMalloc(&sc.desc, b.n * sizeof(KidReadDesc)));
            sc.desc_cap = b.n;
        }
        if (long_cut) {
            if (!sc.long_list) {
                KID_HIP(hipMalloc(&sc.long_list, sizeof(KidLongList)));
                KID_HIP(hipMalloc(&sc.long_plan, sizeof(KidLongPlan)));
                KID_HIP(hipMemset(sc.long_list, 0, 16));
            }
            // one word per k-mer position of the long records, one flag per 256: as many as the batch has bases (a bound
            // the host knows), at most 128 M (records beyond that stay with the classify kernels)
            const uint64_t want = bases_nbytes < (128ull << 20) ? bases_nbytes : (128ull << 20);
            if (want > s->long_hits_cap) {
                KID_HIP(hipDeviceSynchronize());
                if (s->long_hits) hipFree(s->long_hits);
                if (s->long_tiles) hipFree(s->long_tiles);
                s->long_hits = nullptr; s->long_tiles = nullptr; s->long_hits_cap = 0; s->long_tiles_cap = 0;
                const uint64_t cap = want + want / 4;
                KID_HIP(hipMalloc(&s->long_hits, cap * 4));
                KID_HIP(hipMalloc(&s->long_tiles, cap / 256 + KID_LONG_MAX + 16));
                s->long_hits_cap = cap;
                s->long_tiles_cap = cap / 256 + KID_LONG_MAX;
            }
        }
        // The batch that used this set three batches ago must be through its classify kernels before the set is overwritten.
        // On the stream those kernels ran on that is a matter of stream order; only a different prepare stream needs an
        // event -- recorded behind the set's classify kernels when those ran beside a prepare stream (the host path), else
        // now, behind everything queued on that stream so far (an event per batch, recorded and waited for, kept the GPU idle
        // for ~10 us of every step).
        if (sc.used && sc.used_stream != prep_stream) {
            if (sc.used_recorded || hipEventRecord(sc.ev_used, sc.used_stream) == hipSuccess) KID_HIP(hipStreamWaitEvent(prep_stream, sc.ev_used, 0));
            else { // (a caller's stream that is gone by now: everything queued on it has run or the device is in error)
                (void)hipGetLastError();
                KID_HIP(hipDeviceSynchronize());
            }
        }
        const bool fuse_rebase = prep_stream == stream;
        if (fastq)
            hipLaunchKernelGGL(kid_prepare_fastq_kernel, dim3(kid_grid_for(b.n, 256, db->num_cu * 8)), dim3(256), 0, prep_stream, b.bases,
                               fastq->recs, b.n, db->info.k, sc.desc, const_cast<int32_t *>(b.start), const_cast<int32_t *>(b.stop),
                               b.out_final, s->stats, s->gcount, sc.rare, ++s->batch_seq, fuse_rebase ? 1 : 0);
        else
            hipLaunchKernelGGL(kid_prepare_kernel, dim3(kid_grid_for(b.n, 256, db->num_cu * 8)), dim3(256), 0, prep_stream, b, db->info.k,
                               sc.desc, s->stats, sc.rare, ++s->batch_seq, long_cut, sc.long_list, fuse_rebase ? 1 : 0);
        if (prep_stream != stream) {
            KID_HIP(hipEventRecord(sc.ev_prep, prep_stream));
            KID_HIP(hipStreamWaitEvent(stream, sc.ev_prep, 0));
        }
        if (long_cut) // the flagged records -> their places in the hit array (in classify-stream order, before the kernels)
            hipLaunchKernelGGL(kid_long_plan_kernel, dim3(1), dim3(64), 0, stream, sc.long_list, sc.long_plan, sc.desc, sc.rare, s->batch_seq,
                               s->long_hits_cap, s->long_tiles_cap);
    }
    const int block = 512, wpb = block / 64;
    const uint32_t ntar = (uint32_t)db->info.ntar;
    // the gcount histogram lives in LDS while four workgroups per CU (160 KiB) still fit beside the waves' strips
    // and queues.  Minimizer-localised table: two 16-bit counters per word, so a workgroup must stay below 65536
    // reads per launch -- a larger batch is classified in several launches of the same grid (`span` reads each).
    const bool ml = db->d.minloc != 0;
    // Workgroups per CU in the grid.  Four are resident; the SIMDs serve their oldest waves first, so equal shares
    // finish far apart (45 % .. 100 % of a launch) and a grid of exactly the resident workgroups ends at a falling
    // occupancy.  With 16 per CU the dispatcher hands a new workgroup to a CU whenever one is through: 0.97 instead of
    // 1.02 ms per 2 M reads (profiles/r02/ab_grid_mult.txt).
    const int grid = kid_grid_for(b.n, wpb, db->num_cu * 16);
    const uint32_t hist_words32 = (ntar + 3u) & ~3u, hist_words16 = ((ntar + 1u) / 2u + 3u) & ~3u;
    const uint32_t hist_words = ml ? hist_words16 : hist_words32;
    const uint32_t wave_words = ml ? KID_GEN_ML_LDS_WORDS : KID_WAVE_LDS_WORDS;
    const bool hist = (hist_words + wpb * wave_words) * 4u + 32u <= 40u * 1024u;
    const bool hist_pair = (hist_words16 + wpb * KID_PAIR_LDS_WORDS) * 4u + 32u <= 40u * 1024u;
    uint64_t span = b.n;
    if (ml && (hist || hist_pair)) {
        // reads per launch: < 65536 per workgroup -- and with the tapered shares (KID_TAPER) the workgroups dispatched
        // first take up to 1.7 x the average (+ rounding to whole units per wave)
        const uint64_t per_wg = (uint64_t)(65535u - 2u * (uint32_t)wpb - 64u * (uint32_t)wpb) * (KID_TAPER + 1u) / (2u * KID_TAPER);
        // (the bound assumes the two halves of the grid are equal: an odd grid -- only ever a small one -- gets half of it)
        const uint64_t cap = (uint64_t)grid * ((grid & 1) ? per_wg / 2 : per_wg);
        if (span > cap) span = cap;
    }
    KidSampleDev sd{s->gcount, s->seen, s->stats};
    const bool rows = db->rows != nullptr;
    const int32_t fixed_nk = fixed ? (int32_t)(max_kmers > 0 ? max_kmers : 0) : 0;
    KidEvent ev0, ev1;
    if (s->timing) {
        KID_HIP(ev0.create());
        KID_HIP(ev1.create());
        KID_HIP(hipEventRecord(ev0.e, stream));
    }
    for (uint64_t r0 = 0; r0 < b.n; r0 += span) {
    const uint64_t cnt = b.n - r0 < span ? b.n - r0 : span;
    KidInput pk{b.bases, fixed ? nullptr : scp->desc + r0, b.out_final ? b.out_final + r0 : nullptr, cnt};
    // the kernels find this launch's descriptors (or the fixed layout) and result array in the device argument block
    if (fixed) {
        auto &h = s->fixed_held;
        if (!h.valid || h.out_final != pk.out_final || h.read0 != r0 || h.fixed_len != b.fixed_len || h.fixed_nk != fixed_nk) {
            hipLaunchKernelGGL(kid_rebase_kernel, dim3(1), dim3(64), 0, stream, rare, (const KidReadDesc *)nullptr, pk.out_final,
                               (unsigned long long)r0, b.fixed_len, fixed_nk, (1ull << 32) | (unsigned long long)fixed_nk);
            h.valid = true; h.out_final = pk.out_final; h.read0 = r0; h.fixed_len = b.fixed_len; h.fixed_nk = fixed_nk;
        }
    } else if (r0 != 0 || prep_stream != stream) { // (the first launch of a batch prepared on this stream: done by kid_prepare_kernel)
        hipLaunchKernelGGL(kid_rebase_kernel, dim3(1), dim3(64), 0, stream, rare, pk.desc, pk.out_final, 0ull, 0u, 0, 0ull);
    }
#define KID_LAUNCH1(R, H, M, KF, PK)                                                                                            \
    hipLaunchKernelGGL((kid_classify_kernel<2, R, H, M, KF, PK>), dim3(grid), dim3(block),                                     \
                       (((H) ? ((PK) ? hist_words16 : hist_words) : 0u) +                                                       \
                        (size_t)wpb * ((PK) ? KID_PAIR_LDS_WORDS : wave_words)) * 4 + 32, stream,                              \
                       db->d, pk, sd, (H) ? ((PK) ? hist_words16 : hist_words) : 0u, pk.desc, rare)
#define KID_LAUNCH(R, H, M)                                                                                                    \
    do {                                                                                                                       \
        if (db->info.k == 30) KID_LAUNCH1(R, H, M, 30, 0);                                                                     \
        else KID_LAUNCH1(R, H, M, 0, 0);                                                                                       \
    } while (0)
    // the minimizer-localised table has three kernels (see kid_classify_kernel): pair loop, duo loop, general loops
#define KID_LAUNCH_PK(R, H, MODE)                                                                                              \
    do {                                                                                                                       \
        if (db->info.k == 30) KID_LAUNCH1(R, H, true, 30, MODE);                                                               \
        else KID_LAUNCH1(R, H, true, 0, MODE);                                                                                 \
    } while (0)
    // kernel 1: pairs of single-group reads (<= 128 k-mers); kernel 2: the two groups of a read (<= 256); kernel 0: the rest
    const bool want_pair = ml && (max_kmers < 0 || max_kmers <= 2 * 64);
    const bool want_duo = ml && (max_kmers < 0 || (max_kmers > 2 * 64 && max_kmers <= 4 * 64));
    const bool want_general = !ml || max_kmers < 0 || max_kmers > 4 * 64;
    if (want_pair) {
        if (rows && hist_pair) KID_LAUNCH_PK(true, true, 1);
        else if (rows) KID_LAUNCH_PK(true, false, 1);
        else if (hist_pair) KID_LAUNCH_PK(false, true, 1);
        else KID_LAUNCH_PK(false, false, 1);
    }
    if (want_duo) {
        if (rows && hist_pair) KID_LAUNCH_PK(true, true, 2);
        else if (rows) KID_LAUNCH_PK(true, false, 2);
        else if (hist_pair) KID_LAUNCH_PK(false, true, 2);
        else KID_LAUNCH_PK(false, false, 2);
    }
    if (!want_general) { }
    else if (rows && hist && ml) KID_LAUNCH(true, true, true);
    else if (rows && hist) KID_LAUNCH(true, true, false);
    else if (rows && ml) KID_LAUNCH(true, false, true);
    else if (rows) KID_LAUNCH(true, false, false);
    else if (hist && ml) KID_LAUNCH(false, true, true);
    else if (hist) KID_LAUNCH(false, true, false);
    else if (ml) KID_LAUNCH(false, false, true);
    else KID_LAUNCH(false, false, false);
#undef KID_LAUNCH_PK
#undef KID_LAUNCH1
#undef KID_LAUNCH
    }
    if (long_cut) {
        // the very long records: every k-mer looked up by a lane of its own, then one workgroup per record folds its hits.
        // (The grids do not depend on how many there are -- only the device knows: without any, the kernels return at once.)
        kid_sample::Scratch &sc = *scp;
        hipLaunchKernelGGL(kid_long_hits_kernel, dim3((unsigned)db->num_cu * 8u), dim3(256), 0, stream, db->d, b.bases, sc.long_plan,
                           s->long_hits, s->long_tiles, s->seen, s->stats);
        hipLaunchKernelGGL(kid_long_fold_kernel, dim3(KID_LONG_MAX), dim3(256), 0, stream, db->d, sc.long_plan, s->long_hits, s->long_tiles,
                           s->gcount, b.out_final);
    }
    if (s->timing) {
        KID_HIP(hipEventRecord(ev1.e, stream));
        s->timed.emplace_back(ev0.release(), ev1.release());
        s->timed_batches++;
    }
    s->dev_clock_batches++;
    if (scp) {
        // prepare on a stream of its own (the host path: it runs beside the classify kernels of the batch before): the
        // prepare that overwrites this set three batches on waits for exactly these kernels, not for whatever the classify
        // stream holds by then (profiles/r02/ab_lazy_event.txt)
        scp->used_recorded = prep_stream != stream;
        if (scp->used_recorded) KID_HIP(hipEventRecord(scp->ev_used, stream));
        scp->used_stream = stream;
        scp->used = true;
    }
    KID_HIP(hipGetLastError());
    s->reads_submitted += b.n;
    return KID_OK;
}

extern "C" int kid_sample_set_option(kid_sample *s, int option, int value)
{
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    switch (option) {
    case KID_OPT_INPUTS_READY: s->inputs_ready = value != 0; return KID_OK;
    case KID_OPT_LONG_RECORD_KMERS:
        if (value < 0) return kid_fail(KID_ERR_ARG, "KID_OPT_LONG_RECORD_KMERS: negative threshold");
        s->long_kmers = value;
        return KID_OK;
    default: return kid_fail(KID_ERR_ARG, "unknown option %d", option);
    }
}

// pack + prepare stream of the *_device entry points
static int kid_prep_stream_for(kid_sample *s, hipStream_t stream, hipStream_t *out)
{
    *out = stream;
    if (!s->inputs_ready) return KID_OK;
    if (!s->prep_stream) KID_HIP(hipStreamCreateWithFlags(&s->prep_stream, hipStreamNonBlocking));
    *out = s->prep_stream;
    return KID_OK;
}

extern "C" int kid_sample_set_timing(kid_sample *s, int enabled)
{
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    s->timing = enabled != 0;
    return KID_OK;
}

extern "C" int kid_sample_kernel_time(kid_sample *s, double *total_ms, uint64_t *launches)
{
    if (!s || !total_ms || !launches) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    double sum = 0;
    for (auto &ev : s->timed) {
        float ms = 0;
        KID_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        sum += ms;
        hipEventDestroy(ev.first);
        hipEventDestroy(ev.second);
    }
    *total_ms = sum;
    *launches = s->timed_batches; // batches: the kernels of one batch count as one launch
    s->timed.clear();
    s->timed_batches = 0;
    return KID_OK;
}

extern "C" int kid_sample_kernel_time_device(kid_sample *s, double *total_ms, uint64_t *launches)
{
    if (!s || !total_ms || !launches) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    unsigned long long st[32];
    KID_HIP(hipMemcpy(st, s->stats, sizeof(st), hipMemcpyDeviceToHost));
    const unsigned long long ticks = st[6]; // banked by the last workgroup of every launch ([7]: launches)
    const unsigned long long zero2[2] = {0, 0};
    KID_HIP(hipMemcpy(s->stats + 6, zero2, 16, hipMemcpyHostToDevice));
    *total_ms = (double)ticks / 1e5; // s_memrealtime: 100 MHz
    *launches = s->dev_clock_batches; // batches, like kid_sample_kernel_time (a large batch is several launches)
    s->dev_clock_batches = 0;
    return KID_OK;
}

extern "C" int kid_classify_batch_device(kid_sample *s, const void *d_bases, uint64_t bases_nbytes, const void *d_offsets,
                                         const void *d_start, const void *d_stop, uint64_t n_reads, void *d_out_final_targ,
                                         void *stream)
{
    if (!s || (n_reads && (!d_bases || !d_offsets))) return kid_fail(KID_ERR_ARG, "null argument");
    if (((uintptr_t)d_bases & 15u) != 0) return kid_fail(KID_ERR_ARG, "d_bases must be 16-byte aligned");
    if ((d_start == nullptr) != (d_stop == nullptr)) return kid_fail(KID_ERR_ARG, "start and stop must both be given or both be null");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KidBatch b{};
    b.bases = (const uint8_t *)d_bases;
    b.offsets = (const uint64_t *)d_offsets;
    b.start = (const int32_t *)d_start;
    b.stop = (const int32_t *)d_stop;
    b.out_final = (uint32_t *)d_out_final_targ;
    b.n = n_reads;
    b.fixed_len = 0;
    hipStream_t prep;
    rc = kid_prep_stream_for(s, (hipStream_t)stream, &prep);
    if (rc != KID_OK) return rc;
    return kid_launch_classify(s, b, bases_nbytes, (hipStream_t)stream, -1, prep, /*long_records=*/true);
}

extern "C" int kid_classify_fixed_device(kid_sample *s, const void *d_bases, uint32_t read_len, uint64_t n_reads,
                                         void *d_out_final_targ, void *stream)
{
    if (!s || (n_reads && !d_bases)) return kid_fail(KID_ERR_ARG, "null argument");
    if (((uintptr_t)d_bases & 15u) != 0) return kid_fail(KID_ERR_ARG, "d_bases must be 16-byte aligned");
    if (read_len == 0 || read_len > 0x7FFFFFFFu) return kid_fail(KID_ERR_ARG, "read_len out of range");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KidBatch b{};
    b.bases = (const uint8_t *)d_bases;
    b.out_final = (uint32_t *)d_out_final_targ;
    b.n = n_reads;
    b.fixed_len = read_len;
    const int64_t nk = (int64_t)read_len - s->db->info.k + 1;
    hipStream_t prep;
    rc = kid_prep_stream_for(s, (hipStream_t)stream, &prep);
    if (rc != KID_OK) return rc;
    return kid_launch_classify(s, b, n_reads * (uint64_t)read_len, (hipStream_t)stream, nk > 0 ? nk : 0, prep);
}

// ---- host buffers: asynchronous slot pipeline -------------------------------------------------------------------
static int kid_slot_acquire(kid_sample *s, uint64_t n_reads, uint64_t nbytes, bool with_offsets, bool with_range,
                            kid_sample::Slot **out)
{
    if (!s->copy_stream) KID_HIP(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    if (!s->out_stream) KID_HIP(hipStreamCreateWithFlags(&s->out_stream, hipStreamNonBlocking));
    kid_sample::Slot &sl = s->slots[s->next_ticket % kid_sample::NSLOT];
    if (!sl.ev_h2d) {
        KID_HIP(hipEventCreateWithFlags(&sl.ev_h2d, hipEventDisableTiming));
        KID_HIP(hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
        KID_HIP(hipEventCreateWithFlags(&sl.ev_out, hipEventDisableTiming));
    }
    if (sl.busy) { // the batch that used this slot three tickets ago
        KID_HIP(hipEventSynchronize(sl.ev_out));
        sl.busy = false;
    }
    const uint64_t need = ((nbytes + 15) & ~15ull) + 32;
    if (need > sl.bases_cap) {
        if (sl.bases) KID_HIP(hipFree(sl.bases));
        sl.bases = nullptr; sl.bases_cap = 0;
        const uint64_t cap = need + need / 8; // a little slack: batches of a file differ slightly in size
        KID_HIP(hipMalloc(&sl.bases, cap));
        sl.bases_cap = cap;
    }
    if (n_reads > sl.reads_cap) {
        if (sl.offsets) KID_HIP(hipFree(sl.offsets));
        if (sl.start) KID_HIP(hipFree(sl.start));
        if (sl.stop) KID_HIP(hipFree(sl.stop));
        if (sl.out) KID_HIP(hipFree(sl.out));
        sl.offsets = nullptr; sl.start = sl.stop = nullptr; sl.out = nullptr; sl.reads_cap = 0;
        const uint64_t cap = n_reads + n_reads / 8;
        KID_HIP(hipMalloc(&sl.offsets, (cap + 1) * 8));
        KID_HIP(hipMalloc(&sl.start, cap * 4));
        KID_HIP(hipMalloc(&sl.stop, cap * 4));
        KID_HIP(hipMalloc(&sl.out, cap * 4));
        sl.reads_cap = cap;
    }
    (void)with_offsets; (void)with_range;
    *out = &sl;
    return KID_OK;
}

// upload issued on the copy stream -> kernels on the sample's stream -> results on the result stream
static int kid_slot_submit(kid_sample *s, kid_sample::Slot &sl, const KidBatch &b, uint64_t nbytes, int64_t max_kmers,
                           uint32_t *out_final_targ, uint64_t *ticket, bool long_records = false)
{
    // the prepare kernel follows the upload on the copy stream (beside the classify kernels of the batch before); the
    // classify kernels wait for it on the sample's stream
    KID_HIP(hipEventRecord(sl.ev_h2d, s->copy_stream));
    // (a fixed-layout batch has no prepare kernel on the copy stream for the classify kernels to wait for: they wait for the upload itself)
    if (!b.offsets) KID_HIP(hipStreamWaitEvent(s->stream, sl.ev_h2d, 0));
    int rc = kid_launch_classify(s, b, nbytes, s->stream, max_kmers, s->copy_stream, long_records);
    if (rc != KID_OK) return rc;
    KID_HIP(hipEventRecord(sl.ev_done, s->stream));
    if (out_final_targ) {
        KID_HIP(hipStreamWaitEvent(s->out_stream, sl.ev_done, 0));
        KID_HIP(hipMemcpyAsync(out_final_targ, sl.out, b.n * 4, hipMemcpyDeviceToHost, s->out_stream));
        KID_HIP(hipEventRecord(sl.ev_out, s->out_stream));
    } else {
        KID_HIP(hipEventRecord(sl.ev_out, s->stream));
    }
    sl.busy = true;
    sl.ticket = s->next_ticket++;
    if (ticket) *ticket = sl.ticket;
    return KID_OK;
}

extern "C" int kid_classify_batch_async(kid_sample *s, const uint8_t *bases, const uint64_t *offsets, const int32_t *start,
                                        const int32_t *stop, uint64_t n_reads, uint32_t *out_final_targ, uint64_t *ticket)
{
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    if (ticket) *ticket = 0;
    if (n_reads == 0) return KID_OK;
    if (!bases || !offsets) return kid_fail(KID_ERR_ARG, "null argument");
    if ((start == nullptr) != (stop == nullptr)) return kid_fail(KID_ERR_ARG, "start and stop must both be given or both be null");
    // A record of more than s->long_kmers k-mers is a long record (a FASTA contig, kmer_read_vf6.cpp:803-861): the classify
    // kernels would give it to one wave; the launch sorts those out on the device (kid_long_*).
    int64_t max_kmers = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        if (offsets[r + 1] < offsets[r]) return kid_fail(KID_ERR_ARG, "offsets not monotone at read %llu", (unsigned long long)r);
        const uint64_t len = offsets[r + 1] - offsets[r];
        const int64_t span = start ? (int64_t)stop[r] - (int64_t)start[r] + 1 : (int64_t)len;
        const int64_t nk = span - s->db->info.k + 1;
        if (nk > max_kmers) max_kmers = nk;
        if (len > 0x7FFFFFFFull) return kid_fail(KID_ERR_ARG, "read %llu longer than 2^31-1", (unsigned long long)r);
        if (start && start[r] <= stop[r] && (start[r] < 0 || (uint64_t)stop[r] >= len))
            return kid_fail(KID_ERR_ARG, "read %llu: [start,stop] = [%d,%d] outside the read of length %llu (string::at would throw)",
                            (unsigned long long)r, start[r], stop[r], (unsigned long long)len);
    }
    const bool long_records = s->long_kmers > 0 && max_kmers > s->long_kmers;
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    const uint64_t base0 = offsets[0], nbytes = offsets[n_reads] - base0;
    kid_sample::Slot *slp = nullptr;
    rc = kid_slot_acquire(s, n_reads, nbytes, true, start != nullptr, &slp);
    if (rc != KID_OK) return rc;
    kid_sample::Slot &sl = *slp;
    hipStream_t cs = s->copy_stream;
    const uint64_t *off_src = offsets;
    if (base0 != 0) {
        sl.rel.resize(n_reads + 1);
        for (uint64_t r = 0; r <= n_reads; r++) sl.rel[r] = offsets[r] - base0;
        off_src = sl.rel.data();
    }
    const uint64_t need = ((nbytes + 15) & ~15ull) + 32;
    KID_HIP(hipMemsetAsync(sl.bases + (nbytes & ~15ull), 0, need - (nbytes & ~15ull), cs));
    if (nbytes) KID_HIP(hipMemcpyAsync(sl.bases, bases + base0, nbytes, hipMemcpyHostToDevice, cs));
    KID_HIP(hipMemcpyAsync(sl.offsets, off_src, (n_reads + 1) * 8, hipMemcpyHostToDevice, cs));
    if (start) {
        KID_HIP(hipMemcpyAsync(sl.start, start, n_reads * 4, hipMemcpyHostToDevice, cs));
        KID_HIP(hipMemcpyAsync(sl.stop, stop, n_reads * 4, hipMemcpyHostToDevice, cs));
    }
    KidBatch b{};
    b.bases = sl.bases;
    b.offsets = sl.offsets;
    b.start = start ? sl.start : nullptr;
    b.stop = start ? sl.stop : nullptr;
    b.out_final = sl.out;
    b.n = n_reads;
    return kid_slot_submit(s, sl, b, nbytes, max_kmers, out_final_targ, ticket, long_records);
}

extern "C" int kid_classify_fixed_async(kid_sample *s, const uint8_t *bases, uint32_t read_len, uint64_t n_reads,
                                        uint32_t *out_final_targ, uint64_t *ticket)
{
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    if (ticket) *ticket = 0;
    if (n_reads == 0) return KID_OK;
    if (!bases) return kid_fail(KID_ERR_ARG, "null argument");
    if (read_len == 0 || read_len > 0x7FFFFFFFu) return kid_fail(KID_ERR_ARG, "read_len out of range");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    const uint64_t nbytes = n_reads * (uint64_t)read_len;
    kid_sample::Slot *slp = nullptr;
    rc = kid_slot_acquire(s, n_reads, nbytes, false, false, &slp);
    if (rc != KID_OK) return rc;
    kid_sample::Slot &sl = *slp;
    const uint64_t need = ((nbytes + 15) & ~15ull) + 32;
    KID_HIP(hipMemsetAsync(sl.bases + (nbytes & ~15ull), 0, need - (nbytes & ~15ull), s->copy_stream));
    KID_HIP(hipMemcpyAsync(sl.bases, bases, nbytes, hipMemcpyHostToDevice, s->copy_stream));
    KidBatch b{};
    b.bases = sl.bases;
    b.out_final = sl.out;
    b.n = n_reads;
    b.fixed_len = read_len;
    const int64_t nk = (int64_t)read_len - s->db->info.k + 1;
    return kid_slot_submit(s, sl, b, nbytes, nk > 0 ? nk : 0, out_final_targ, ticket);
}

// A block of FASTQ text whose lines the caller has found: quality trimming (process_qual), the ">= k" test and
// process_read all happen on the GPU; the host's share of process_fqgz (newkmer_10nx.cpp:762-816) is inflate + memchr.
extern "C" int kid_classify_fastq_async(kid_sample *s, const uint8_t *text, uint64_t text_nbytes, const kid_fastq_rec *recs,
                                        uint64_t n_reads, uint32_t *out_final_targ, int32_t *out_start, int32_t *out_stop,
                                        uint64_t *ticket)
{
    static_assert(sizeof(kid_fastq_rec) == sizeof(KidFastqRec), "kid_fastq_rec is the device record");
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    if (ticket) *ticket = 0;
    if (n_reads == 0) return KID_OK;
    if (!text || !recs || !out_start || !out_stop) return kid_fail(KID_ERR_ARG, "null argument");
    if (text_nbytes >= 0xFFFFFFFFull) return kid_fail(KID_ERR_ARG, "a FASTQ block of 4 GiB or more");
    for (uint64_t r = 0; r < n_reads; r++)
        if ((uint64_t)recs[r].seq_off + recs[r].seq_len > text_nbytes || (uint64_t)recs[r].qual_off + recs[r].qual_len > text_nbytes)
            return kid_fail(KID_ERR_ARG, "record %llu lies outside the text block", (unsigned long long)r);
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    kid_sample::Slot *slp = nullptr;
    rc = kid_slot_acquire(s, n_reads, text_nbytes, true, true, &slp);
    if (rc != KID_OK) return rc;
    kid_sample::Slot &sl = *slp;
    if (n_reads > sl.recs_cap) {
        if (sl.recs) KID_HIP(hipFree(sl.recs));
        sl.recs = nullptr; sl.recs_cap = 0;
        const uint64_t cap = n_reads + n_reads / 8;
        KID_HIP(hipMalloc(&sl.recs, cap * sizeof(KidFastqRec)));
        sl.recs_cap = cap;
    }
    hipStream_t cs = s->copy_stream;
    const uint64_t need = ((text_nbytes + 15) & ~15ull) + 32;
    KID_HIP(hipMemsetAsync(sl.bases + (text_nbytes & ~15ull), 0, need - (text_nbytes & ~15ull), cs));
    KID_HIP(hipMemcpyAsync(sl.bases, text, text_nbytes, hipMemcpyHostToDevice, cs));
    KID_HIP(hipMemcpyAsync(sl.recs, recs, n_reads * sizeof(KidFastqRec), hipMemcpyHostToDevice, cs));
    KidBatch b{};
    b.bases = sl.bases;
    b.start = sl.start; // (outputs of the prepare kernel here)
    b.stop = sl.stop;
    b.out_final = sl.out;
    b.n = n_reads;
    KidFastqIn fq{sl.recs};
    KID_HIP(hipEventRecord(sl.ev_h2d, s->copy_stream));
    rc = kid_launch_classify(s, b, text_nbytes, s->stream, -1, s->copy_stream, false, &fq);
    if (rc != KID_OK) return rc;
    KID_HIP(hipEventRecord(sl.ev_done, s->stream));
    KID_HIP(hipStreamWaitEvent(s->out_stream, sl.ev_done, 0));
    if (out_final_targ) KID_HIP(hipMemcpyAsync(out_final_targ, sl.out, n_reads * 4, hipMemcpyDeviceToHost, s->out_stream));
    KID_HIP(hipMemcpyAsync(out_start, sl.start, n_reads * 4, hipMemcpyDeviceToHost, s->out_stream));
    KID_HIP(hipMemcpyAsync(out_stop, sl.stop, n_reads * 4, hipMemcpyDeviceToHost, s->out_stream));
    KID_HIP(hipEventRecord(sl.ev_out, s->out_stream));
    sl.busy = true;
    sl.ticket = s->next_ticket++;
    if (ticket) *ticket = sl.ticket;
    return KID_OK;
}

extern "C" int kid_classify_wait(kid_sample *s, uint64_t ticket)
{
    if (!s) return kid_fail(KID_ERR_ARG, "null sample");
    if (ticket == 0) return KID_OK; // an empty batch
    if (ticket >= s->next_ticket) return kid_fail(KID_ERR_ARG, "ticket %llu has not been issued", (unsigned long long)ticket);
    kid_sample::Slot &sl = s->slots[ticket % kid_sample::NSLOT];
    if (sl.ticket != ticket || !sl.busy) return KID_OK; // its slot has been waited for (and maybe reused) already
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipEventSynchronize(sl.ev_out));
    sl.busy = false;
    return KID_OK;
}

extern "C" int kid_classify_batch(kid_sample *s, const uint8_t *bases, const uint64_t *offsets, const int32_t *start,
                                  const int32_t *stop, uint64_t n_reads, uint32_t *out_final_targ)
{
    uint64_t ticket = 0;
    int rc = kid_classify_batch_async(s, bases, offsets, start, stop, n_reads, out_final_targ, &ticket);
    if (rc != KID_OK) return rc;
    return kid_classify_wait(s, ticket);
}

// CPUs of the NUMA node the GPU's PCIe root port hangs off (sysfs); false when the box does not say
static bool kid_device_local_cpus(int device, cpu_set_t *set)
{
    char bdf[64] = {0};
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device) != hipSuccess) return false;
    for (char *c = bdf; *c; c++) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1) node = -1;
    fclose(f);
    if (node < 0) return false;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return false;
    char list[4096] = {0};
    const bool ok = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!ok) return false;
    CPU_ZERO(set);
    int n = 0;
    for (char *p = list; *p;) { // "0-63,128-191"
        char *end;
        long a = strtol(p, &end, 10);
        if (end == p) break;
        long b = a;
        if (*end == '-') { p = end + 1; b = strtol(p, &end, 10); }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) { CPU_SET((int)c, set); n++; }
        p = (*end == ',') ? end + 1 : end;
        if (*end != ',' ) break;
    }
    return n > 0;
}

extern "C" int kid_host_alloc(int device, uint64_t nbytes, void **ptr)
{
    if (!ptr) return kid_fail(KID_ERR_ARG, "null argument");
    *ptr = nullptr;
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    // Page-locked memory is placed where the allocating thread runs; DMA from the other socket's memory reaches the
    // GPU at little more than half the PCIe rate (measured 32 vs 55 GB/s).  So: allocate from a CPU next to the GPU.
    cpu_set_t old_set, local;
    const bool have_old = sched_getaffinity(0, sizeof(old_set), &old_set) == 0;
    bool moved = false;
    if (have_old && kid_device_local_cpus(device, &local)) {
        cpu_set_t both;
        CPU_AND(&both, &local, &old_set); // never leave the CPUs this process was given
        if (CPU_COUNT(&both) > 0) moved = sched_setaffinity(0, sizeof(both), &both) == 0;
    }
    hipError_t e = hipHostMalloc(ptr, nbytes ? nbytes : 16, hipHostMallocDefault);
    if (moved) sched_setaffinity(0, sizeof(old_set), &old_set);
    if (e != hipSuccess) {
        *ptr = nullptr;
        return kid_fail(e == hipErrorOutOfMemory ? KID_ERR_NOMEM : KID_ERR_HIP, "hipHostMalloc(%llu) failed: %s",
                        (unsigned long long)nbytes, hipGetErrorString(e));
    }
    return KID_OK;
}

extern "C" int kid_host_free(void *ptr)
{
    if (ptr) KID_HIP(hipHostFree(ptr));
    return KID_OK;
}

extern "C" int kid_trim_batch(kid_db *db, const uint8_t *quals, const uint64_t *offsets, uint64_t n_reads, int32_t *start,
                              int32_t *stop, uint8_t *keep)
{
    if (!db) return kid_fail(KID_ERR_ARG, "null db");
    if (n_reads == 0) return KID_OK;
    if (!quals || !offsets || !start || !stop || !keep) return kid_fail(KID_ERR_ARG, "null argument");
    for (uint64_t r = 0; r < n_reads; r++)
        if (offsets[r + 1] < offsets[r] || offsets[r + 1] - offsets[r] > 0x7FFFFFFFull)
            return kid_fail(KID_ERR_ARG, "bad offsets at read %llu", (unsigned long long)r);
    int rc = kid_use_device(db->device);
    if (rc != KID_OK) return rc;
    const uint64_t base0 = offsets[0], nbytes = offsets[n_reads] - base0;
    KidDevBuf dq, dkeep, doff, ds, de;
    std::vector<uint64_t> rel(n_reads + 1);
    for (uint64_t r = 0; r <= n_reads; r++) rel[r] = offsets[r] - base0;
    KID_HIP(dq.alloc(nbytes + 16));
    KID_HIP(doff.alloc((n_reads + 1) * 8));
    KID_HIP(ds.alloc(n_reads * 4));
    KID_HIP(de.alloc(n_reads * 4));
    KID_HIP(dkeep.alloc(n_reads));
    if (nbytes) KID_HIP(hipMemcpy(dq.p, quals + base0, nbytes, hipMemcpyHostToDevice));
    KID_HIP(hipMemcpy(doff.p, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kid_trim_kernel, dim3(kid_grid_for(n_reads, 256, db->num_cu * 16)), dim3(256), 0, 0, dq.as<uint8_t>(),
                       doff.as<uint64_t>(), n_reads, db->info.k, ds.as<int32_t>(), de.as<int32_t>(), dkeep.as<uint8_t>());
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(start, ds.p, n_reads * 4, hipMemcpyDeviceToHost));
    KID_HIP(hipMemcpy(stop, de.p, n_reads * 4, hipMemcpyDeviceToHost));
    KID_HIP(hipMemcpy(keep, dkeep.p, n_reads, hipMemcpyDeviceToHost));
    return KID_OK;
}

// ---------------------------------------------------------------- results
static int kid_check_errors(kid_sample *s)
{
    unsigned long long st[9];
    KID_HIP(hipMemcpy(st, s->stats, sizeof(st), hipMemcpyDeviceToHost));
    if (st[4] != 0)
        return kid_fail(KID_ERR_ARG, "%llu reads had [start,stop] outside the read (string::at would throw)", st[4]);
    if (st[8] != 0)
        return kid_fail(KID_ERR_FORMAT, "%llu FASTQ records have a quality line shorter than the sequence (qual.at() throws in the reference)", st[8]);
    // every read handed over was classified by exactly one of the kernels (they pick themselves by the batch's
    // longest read: a disagreement with the host's choice would show here, not as silently missing reads)
    if (st[0] != s->reads_submitted)
        return kid_fail(KID_ERR_STATE, "%llu reads classified, %llu submitted", st[0], (unsigned long long)s->reads_submitted);
    return KID_OK;
}

extern "C" int kid_sample_gcount(kid_sample *s, int64_t *gcount)
{
    if (!s || !gcount) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(gcount, s->gcount, (size_t)s->db->info.ntar * 8, hipMemcpyDeviceToHost));
    return kid_check_errors(s);
}

extern "C" int kid_sample_ucount_range(kid_sample *s, uint64_t slot_begin, uint64_t slot_end, int64_t *ucount)
{
    if (!s || !ucount) return kid_fail(KID_ERR_ARG, "null argument");
    if (slot_begin > slot_end || slot_end > s->seen_words * 32 || (slot_begin & 127) || (slot_end & 127))
        return kid_fail(KID_ERR_ARG, "bit range must be 128-aligned and inside the bitmap");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    rc = kid_seenlog_flush(s);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    const size_t nt = (size_t)s->db->info.ntar;
    KID_HIP(hipMemset(s->ucount, 0, nt * 8));
    const uint64_t w0 = slot_begin / 32, w1 = slot_end / 32;
    if (w1 > w0) {
        const uint32_t ntar = (uint32_t)s->db->info.ntar;
        const int ugrid = kid_grid_for((w1 - w0) / 4, 512, s->db->num_cu * 4);
        if (ntar * 4u <= 64u * 1024u)
            hipLaunchKernelGGL((kid_ucount_kernel<true>), dim3(ugrid), dim3(512), ntar * 4u, 0, s->seen, w0, w1, s->db->ord_target,
                               s->ucount, ntar);
        else
            hipLaunchKernelGGL((kid_ucount_kernel<false>), dim3(ugrid), dim3(512), 0, 0, s->seen, w0, w1, s->db->ord_target,
                               s->ucount, ntar);
        KID_HIP(hipGetLastError());
    }
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(ucount, s->ucount, nt * 8, hipMemcpyDeviceToHost));
    return KID_OK;
}

extern "C" int kid_sample_end(kid_sample *s, int64_t *gcount, int64_t *ucount)
{
    if (!s || !gcount || !ucount) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_sample_gcount(s, gcount);
    if (rc != KID_OK) return rc;
    return kid_sample_ucount_range(s, 0, s->seen_words * 32, ucount);
}

// The counters of ONE sample of the input whose batches were dealt out over n kid_sample objects -- one per GPU, each
// on its own replica of the database (kid_db_replicate, or kid_db_build from the same entries).  gcount adds; ucount is
// |distinct DB k-mers hit|: the seen-bitmaps (one bit per DB entry, the same numbering on every replica) are copied
// peer to peer into samples[0]'s GPU, OR-ed there and counted once.  This is the merge of the reference's globals
// (newkmer_10nx.cpp:61-64) over the shards; the process-per-GPU form of it over RCCL is kmer_id_amd/dist.py.
// samples[0]'s bitmap holds the union afterwards.
extern "C" int kid_sample_end_merged(kid_sample **samples, int n, int64_t *gcount, int64_t *ucount)
{
    if (!samples || n < 1 || !gcount || !ucount) return kid_fail(KID_ERR_ARG, "bad argument");
    for (int i = 0; i < n; i++) {
        if (!samples[i]) return kid_fail(KID_ERR_ARG, "samples[%d] is null", i);
        if (samples[i]->seen_words != samples[0]->seen_words || samples[i]->db->info.ntar != samples[0]->db->info.ntar ||
            samples[i]->db->info.n_entries != samples[0]->db->info.n_entries)
            return kid_fail(KID_ERR_ARG, "samples[%d] belongs to a database built from other entries", i);
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < i; j++)
            if (samples[i] == samples[j]) return kid_fail(KID_ERR_ARG, "samples[%d] and samples[%d] are the same sample (its reads would be counted twice)", j, i);
    kid_sample *s0 = samples[0];
    const size_t nt = (size_t)s0->db->info.ntar;
    int rc = kid_sample_gcount(s0, gcount);
    if (rc != KID_OK) return rc;
    if (n > 1) {
        std::vector<int64_t> g(nt);
        KidDevBuf tmp;
        rc = kid_use_device(s0->db->device);
        if (rc != KID_OK) return rc;
        const size_t nbytes = (size_t)s0->seen_words * 4;
        KID_HIP(tmp.alloc(nbytes));
        for (int i = 1; i < n; i++) {
            rc = kid_use_device(samples[i]->db->device);
            if (rc != KID_OK) return rc;
            rc = kid_seenlog_flush(samples[i]); // its bitmap is read below
            if (rc != KID_OK) return rc;
            rc = kid_sample_gcount(samples[i], g.data()); // (synchronises samples[i]'s device)
            if (rc != KID_OK) return rc;
            for (size_t t = 0; t < nt; t++) gcount[t] += g[t];
            rc = kid_use_device(s0->db->device);
            if (rc != KID_OK) return rc;
            if (samples[i]->db->device == s0->db->device) KID_HIP(hipMemcpy(tmp.p, samples[i]->seen, nbytes, hipMemcpyDeviceToDevice));
            else KID_HIP(hipMemcpyPeer(tmp.p, s0->db->device, samples[i]->seen, samples[i]->db->device, nbytes));
            rc = kid_sample_seen_or(s0, 0, nbytes, tmp.p, 1);
            if (rc != KID_OK) return rc;
        }
    }
    return kid_sample_ucount_range(s0, 0, s0->seen_words * 32, ucount);
}

extern "C" int kid_sample_stats(kid_sample *s, uint64_t out[4])
{
    if (!s || !out) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    unsigned long long st[8];
    KID_HIP(hipMemcpy(st, s->stats, 64, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) out[i] = st[i];
    return KID_OK;
}

extern "C" int kid_sample_seen_bytes(const kid_sample *s, uint64_t *nbytes)
{
    if (!s || !nbytes) return kid_fail(KID_ERR_ARG, "null argument");
    *nbytes = s->seen_words * 4;
    return KID_OK;
}

extern "C" int kid_sample_seen_export(kid_sample *s, uint64_t byte_off, uint64_t nbytes, void *dst, int dst_on_device)
{
    if (!s || (nbytes && !dst)) return kid_fail(KID_ERR_ARG, "null argument");
    if (byte_off + nbytes > s->seen_words * 4) return kid_fail(KID_ERR_ARG, "range outside the bitmap");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    rc = kid_seenlog_flush(s);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipMemcpy(dst, (const uint8_t *)s->seen + byte_off, nbytes, dst_on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return KID_OK;
}

extern "C" int kid_sample_seen_or(kid_sample *s, uint64_t byte_off, uint64_t nbytes, const void *src, int src_on_device)
{
    if (!s || (nbytes && !src)) return kid_fail(KID_ERR_ARG, "null argument");
    if ((byte_off & 15) || (nbytes & 15) || byte_off + nbytes > s->seen_words * 4)
        return kid_fail(KID_ERR_ARG, "range must be 16-byte aligned and inside the bitmap");
    int rc = kid_use_device(s->db->device);
    if (rc != KID_OK) return rc;
    if (nbytes == 0) return KID_OK;
    const uint32_t *dsrc = (const uint32_t *)src;
    KidDevBuf tmp;
    if (!src_on_device) {
        KID_HIP(tmp.alloc(nbytes));
        KID_HIP(hipMemcpy(tmp.p, src, nbytes, hipMemcpyHostToDevice));
        dsrc = tmp.as<uint32_t>();
    }
    KID_HIP(hipDeviceSynchronize());
    hipLaunchKernelGGL(kid_or_kernel, dim3(kid_grid_for(nbytes / 4, 256, s->db->num_cu * 16)), dim3(256), 0, 0,
                       s->seen + byte_off / 4, dsrc, nbytes / 4);
    KID_HIP(hipDeviceSynchronize());
    return KID_OK;
}

// ---------------------------------------------------------------- synthetic workload
extern "C" int kid_synth_db_keys_host(uint64_t seed, int k, const uint64_t *cum, int32_t ntar, uint64_t j0, uint64_t n,
                                      uint64_t *keys, uint32_t *targets)
{
    if (!cum || !keys || !targets || ntar < 1 || k < 1 || k > 31) return kid_fail(KID_ERR_ARG, "bad argument");
    for (uint64_t i = 0; i < n; i++) {
        keys[i] = kid_synth_db_key(seed, k, j0 + i);
        targets[i] = kid_synth_target_of(cum, ntar, j0 + i);
    }
    return KID_OK;
}

extern "C" int kid_synth_db_keys_device(uint64_t seed, int k, const uint64_t *cum_host, int32_t ntar, uint64_t j0, uint64_t n,
                                        void *d_keys, void *d_targets, int device)
{
    if (!cum_host || !d_keys || !d_targets || ntar < 1 || k < 1 || k > 31) return kid_fail(KID_ERR_ARG, "bad argument");
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    KidDevBuf dcum;
    KID_HIP(dcum.alloc(((size_t)ntar + 1) * 8));
    KID_HIP(hipMemcpy(dcum.p, cum_host, ((size_t)ntar + 1) * 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kid_synth_keys_kernel, dim3(kid_grid_for(n, 256, 256 * 16)), dim3(256), 0, 0, seed, k, dcum.as<uint64_t>(),
                       ntar, j0, n, (uint64_t *)d_keys, (uint32_t *)d_targets);
    KID_HIP(hipDeviceSynchronize());
    return KID_OK;
}

extern "C" int kid_synth_reads_host(uint64_t db_seed, uint64_t read_seed, int k, const uint64_t *cum, const int32_t *parent,
                                    int32_t ntar, uint64_t r0, uint64_t n_reads, uint32_t read_len, uint8_t *bases)
{
    if (!cum || !parent || !bases || ntar < 2 || k < 1 || k > 31 || read_len == 0) return kid_fail(KID_ERR_ARG, "bad argument");
    for (uint64_t i = 0; i < n_reads; i++)
        kid_synth_read(db_seed, read_seed, k, cum, parent, ntar, r0 + i, read_len, bases + i * (uint64_t)read_len);
    return KID_OK;
}

extern "C" int kid_synth_reads_device(uint64_t db_seed, uint64_t read_seed, int k, const uint64_t *cum_host,
                                      const int32_t *parent_host, int32_t ntar, uint64_t r0, uint64_t n_reads,
                                      uint32_t read_len, void *d_bases, int device)
{
    if (!cum_host || !parent_host || !d_bases || ntar < 2 || k < 1 || k > 31 || read_len == 0)
        return kid_fail(KID_ERR_ARG, "bad argument");
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    KidDevBuf dcum, dpar;
    KID_HIP(dcum.alloc(((size_t)ntar + 1) * 8));
    KID_HIP(dpar.alloc((size_t)ntar * 4));
    KID_HIP(hipMemcpy(dcum.p, cum_host, ((size_t)ntar + 1) * 8, hipMemcpyHostToDevice));
    KID_HIP(hipMemcpy(dpar.p, parent_host, (size_t)ntar * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kid_synth_reads_kernel, dim3(kid_grid_for(n_reads, 256, 256 * 16)), dim3(256), 0, 0, db_seed, read_seed, k,
                       dcum.as<uint64_t>(), dpar.as<int32_t>(), ntar, r0, n_reads, read_len, (uint8_t *)d_bases);
    KID_HIP(hipDeviceSynchronize());
    return KID_OK;
}

extern "C" int kid_bench_gather(kid_db *db, uint64_t n_loads, int inflight, int iters, float *ms_out, uint64_t *loads_out)
{
    if (!db || !ms_out || !loads_out || iters < 1) return kid_fail(KID_ERR_ARG, "bad argument");
    int rc = kid_use_device(db->device);
    if (rc != KID_OK) return rc;
    const int block = 256, grid = db->num_cu * 8;
    const uint64_t lanes = (uint64_t)block * grid;
    // inflight = 101 / 108: random LINES, runs of 1 / 8 lanes on a line, 4 loads in flight (kid_gather_lines_kernel);
    // *loads_out is then the number of distinct line requests
    const bool by_line = inflight == 101 || inflight == 108 || inflight == 111 || inflight == 121 || inflight == 131; // (1x1: development variants, see the kernel)
    if (!by_line && inflight != 1 && inflight != 2 && inflight != 4 && inflight != 8) return kid_fail(KID_ERR_ARG, "inflight must be 1,2,4 or 8 (or 101, 108: by line)");
    if (by_line && db->d.slot_mask < 7u) return kid_fail(KID_ERR_ARG, "table too small");
    uint64_t rounds = n_loads / (lanes * (uint64_t)(by_line ? 4 : inflight));
    if (rounds < 1) rounds = 1;
    KidDevBuf sinkb;
    KID_HIP(sinkb.alloc(16));
    uint32_t *const sink = sinkb.as<uint32_t>();
    KidEvent ev0, ev1;
    KID_HIP(ev0.create());
    KID_HIP(ev1.create());
    const hipEvent_t e0 = ev0.e, e1 = ev1.e;
    auto launch = [&]() {
        const uint32_t line_mask = db->d.slot_mask >> 3;
        switch (inflight) {
        case 101: hipLaunchKernelGGL((kid_gather_lines_kernel<1>), dim3(grid), dim3(block), 0, 0, db->table, line_mask, rounds, sink); break;
        case 111: hipLaunchKernelGGL((kid_gather_lines_kernel<1, 1>), dim3(grid), dim3(block), 0, 0, db->table, line_mask, rounds, sink); break;
        case 121: hipLaunchKernelGGL((kid_gather_lines_kernel<1, 2>), dim3(grid), dim3(block), 0, 0, db->table, line_mask, rounds, sink); break;
        case 131: hipLaunchKernelGGL((kid_gather_lines_kernel<1, 3>), dim3(grid), dim3(block), 0, 0, db->table, line_mask, rounds, sink); break;
        case 108: hipLaunchKernelGGL((kid_gather_lines_kernel<8>), dim3(grid), dim3(block), 0, 0, db->table, line_mask, rounds, sink); break;
        case 1: hipLaunchKernelGGL((kid_gather_kernel<1>), dim3(grid), dim3(block), 0, 0, db->table, db->d.slot_mask, rounds, sink); break;
        case 2: hipLaunchKernelGGL((kid_gather_kernel<2>), dim3(grid), dim3(block), 0, 0, db->table, db->d.slot_mask, rounds, sink); break;
        case 4: hipLaunchKernelGGL((kid_gather_kernel<4>), dim3(grid), dim3(block), 0, 0, db->table, db->d.slot_mask, rounds, sink); break;
        default: hipLaunchKernelGGL((kid_gather_kernel<8>), dim3(grid), dim3(block), 0, 0, db->table, db->d.slot_mask, rounds, sink); break;
        }
    };
    launch(); // warm-up
    KID_HIP(hipDeviceSynchronize());
    KID_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) launch();
    KID_HIP(hipEventRecord(e1, 0));
    KID_HIP(hipEventSynchronize(e1));
    float ms = 0;
    KID_HIP(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms / (float)iters;
    *loads_out = by_line ? rounds * (lanes / (inflight == 108 ? 8 : 1)) * 4 : rounds * lanes * (uint64_t)inflight; // loads (lines) actually asked for per launch
    return KID_OK;
}

// ---------------------------------------------------------------- device memory helpers
extern "C" int kid_dev_alloc(int device, uint64_t nbytes, void **d_ptr)
{
    if (!d_ptr) return kid_fail(KID_ERR_ARG, "null argument");
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipMalloc(d_ptr, nbytes ? nbytes : 16));
    return KID_OK;
}
extern "C" int kid_dev_free(int device, void *d_ptr)
{
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    if (d_ptr) KID_HIP(hipFree(d_ptr));
    return KID_OK;
}
extern "C" int kid_dev_upload(int device, void *d_dst, const void *src, uint64_t nbytes)
{
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    if (nbytes) KID_HIP(hipMemcpy(d_dst, src, nbytes, hipMemcpyHostToDevice));
    return KID_OK;
}
extern "C" int kid_dev_download(int device, void *dst, const void *d_src, uint64_t nbytes)
{
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    if (nbytes) KID_HIP(hipMemcpy(dst, d_src, nbytes, hipMemcpyDeviceToHost));
    return KID_OK;
}
extern "C" int kid_dev_sync(int device)
{
    int rc = kid_use_device(device);
    if (rc != KID_OK) return rc;
    KID_HIP(hipDeviceSynchronize());
    return KID_OK;
}
