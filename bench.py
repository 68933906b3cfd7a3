#!/usr/bin/env python3
"""bench.py -- paired reads classified / second on the bact10-synth DB (BASELINE.json).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU.  Every rank holds a full replica of the database in its HBM
(2^30 cells x 16 B = 16 GiB table, 108 585 519 synthetic canonical 30-mers laid
over the reference's real bact10 taxonomy) and classifies its own shard of the
reads (weak scaling: PAIRS_PER_STEP pairs per rank per step).  A "step" is one
pass of the hot path (kid_classify_fixed_device -> kid_classify_kernel) over one
batch of 150-bp reads that is already resident in HBM.  After the K steps the
sample is closed inside the timed region: ucount from the seen-bitmap and, for
N > 1, the RCCL merge (bitmap slices all_to_all + one all_reduce).

Rank 0 prints ONE JSON line (metric, roofline, cpu_baseline).

Defaults: 200 timed steps after 50 warm-up steps (1 M pairs each); the full-size cross-check against the reference's
cell placement, the random-line probe, the host-buffer leg, the nk10 FASTQ.gz leg and the CPU baseline all run AFTER the
timed region.  Development aids (never part of the metric): KID_BENCH_RANDOM_READS=1 (reads without a database k-mer),
KID_BENCH_ABLATION=1 (accept builds whose counters are wrong on purpose: tools/ab_bench.py with -DKID_ABLATE_* libraries).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import kmer_id_amd  # noqa: E402
from kmer_id_amd import KmerDB, synth  # noqa: E402
from kmer_id_amd.dist import merge_sample  # noqa: E402

K = 30
READ_LEN = 150
# BASELINE.json configs: [1] 1 M pairs per step (the configuration the metric is quoted on), [2] 100 M pairs in one
# sample (the roofline run): one step = one batch = kid_classify_fixed_device over the whole resident read set
# (steps / warmup: a step is ~1.1 ms; the first tens of steps after the database build run 5-9 % slower -- clocks and
#  translation caches settle -- so the default warms up for 50 steps and times 200: profiles/r02/ab_warmup.txt)
CONFIGS = {"1m": {"pairs": 1_000_000, "steps": 200, "warmup": 50, "batches": 4},
           "roofline100m": {"pairs": 100_000_000, "steps": 4, "warmup": 2, "batches": 1}}
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def build_db(device, scale, log2_slots, keep_host_copy, flags=0, keep_device_keys=False):
    """bact10-synth: keys generated on the device, table built on the device."""
    parent, cnt = synth.load_taxonomy("bact10")
    cum = synth.cumulative(synth.scaled_counts(cnt, scale))
    n = int(cum[-1])
    lib = kmer_id_amd.load()
    d_keys = torch.empty(n, dtype=torch.int64, device=device)
    d_targets = torch.empty(n, dtype=torch.int32, device=device)
    import ctypes as C
    kmer_id_amd._lib.check(lib.kid_synth_db_keys_device(synth.DB_SEED, K, cum.ctypes.data_as(C.c_void_p), parent.size, 0, n,
                                                        C.c_void_p(d_keys.data_ptr()), C.c_void_p(d_targets.data_ptr()),
                                                        device.index))
    t0 = time.time()
    db = KmerDB.from_device(d_keys.data_ptr(), d_targets.data_ptr(), n, parent, k=K, log2_slots=log2_slots, flags=flags,
                            device=device.index)
    build_s = time.time() - t0
    host = None
    if keep_host_copy:
        host = (d_keys.cpu().numpy().view(np.uint64), d_targets.cpu().numpy().view(np.uint32))
    dev = (d_keys, d_targets, n) if keep_device_keys else None
    del d_keys, d_targets
    torch.cuda.empty_cache()
    return db, parent, cum, build_s, host, dev


def gen_reads(device, cum, parent, r0, n_reads):
    import ctypes as C
    lib = kmer_id_amd.load()
    buf = torch.empty(n_reads * READ_LEN + 64, dtype=torch.uint8, device=device)
    kmer_id_amd._lib.check(lib.kid_synth_reads_device(synth.DB_SEED, synth.READ_SEED, K, cum.ctypes.data_as(C.c_void_p),
                                                      parent.ctypes.data_as(C.c_void_p), parent.size, r0, n_reads, READ_LEN,
                                                      C.c_void_p(buf.data_ptr()), device.index))
    return buf


def host_cores():
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:  # pragma: no cover
        return max(1, os.cpu_count() or 1)


def cpu_baseline(host_keys, parent, cum, log2_slots, n_reads, threads):
    """The oracle (plain-C restatement of the reference's loop, 24-byte cells) on the first n_reads
    reads of the same workload, same DB, on this host's cores: one thread (what the reference is), then
    `threads` threads over contiguous read ranges with one shared table (counters merged the way the
    reference's globals would have ended up: gcount adds, kmer_seen is a union)."""
    from oracle import binding as ob
    keys, targets = host_keys
    t0 = time.time()
    odb = ob.OracleDB(parent.size, K, log2_slots, parent=parent)
    odb.add(keys, targets)
    build_s = time.time() - t0
    os_ = ob.OracleSample(odb)
    bases = synth.reads(cum, parent, n_reads, READ_LEN, K)
    off = synth.fixed_offsets(n_reads, READ_LEN)
    warm = min(n_reads, 2000)
    os_.classify_timed(bases[:warm * READ_LEN], off[:warm + 1])
    os_.reset()
    sec = os_.classify_timed(bases, off)
    st = os_.stats()
    g, u = os_.counts()
    # all cores: the same reads first (the merged counters must equal the sequential ones), then a sample eight
    # times as large for the timing; the merge of the per-thread sets is inside the timed region
    _, gm, um, _ = odb.classify_mt(bases, off, threads)
    if not (np.array_equal(gm, g) and np.array_equal(um, u)):
        raise SystemExit("oracle: %d threads and 1 thread disagree" % threads)
    n_mt = 8 * n_reads
    sec_mt, _, _, st_mt = odb.classify_mt(synth.reads(cum, parent, n_mt, READ_LEN, K), synth.fixed_offsets(n_mt, READ_LEN), threads)
    odb.close()
    return {"value": (n_mt / 2) / sec_mt, "unit": "paired reads/s", "cores": threads, "kind": "port",
            "value_1_thread": (n_reads / 2) / sec, "lookups_per_s_1_thread": st["lookups"] / sec,
            "lookups_per_s": st_mt["lookups"] / sec_mt,
            "sample": "first %d reads (%d pairs) of the same synthetic stream, oracle/kmer_oracle.c (plain-C port of "
                      "newkmer_10nx.cpp:452-617), same %d-key DB in a 2^%d-cell table of 24-byte cells; 1 thread: %.2f s "
                      "classify (%.2f M lookups/s); %d threads over the first %d reads, shared table, counters merged: %.2f s "
                      "(%.2f M lookups/s); %.1f s table build" % (n_reads, n_reads // 2, keys.size, log2_slots, sec,
                                                                  st["lookups"] / sec / 1e6, threads, n_mt, sec_mt,
                                                                  st_mt["lookups"] / sec_mt / 1e6, build_s)}, (g, u)


def host_buffer_leg(db, device, batches, n_reads, steps):
    """The same batches handed over as HOST buffers (pinned, kid_host_alloc) through kid_classify_fixed_async: upload on
    a copy stream, kernels, per-read results back on a result stream, three batches in flight.  PCIe-inclusive
    pairs/s: never the headline value (the metric is quoted with reads resident in HBM)."""
    from kmer_id_amd import PinnedBuffer
    import ctypes as C
    lib = kmer_id_amd.load()
    nb = min(len(batches), 3)
    nbytes = n_reads * READ_LEN
    pins = [PinnedBuffer(nbytes, device.index) for _ in range(nb)]
    outs = [PinnedBuffer(n_reads * 4, device.index) for _ in range(3)]
    for b_, pin in zip(batches, pins):
        kmer_id_amd._lib.check(lib.kid_dev_download(device.index, C.c_void_p(pin.ptr), C.c_void_p(b_.data_ptr()), nbytes))
    s_ = db.sample()

    def run(k):
        tickets = []
        for i in range(k):
            tickets.append(s_.classify_fixed_async(pins[i % nb].ptr, READ_LEN, n_reads, outs[i % 3].ptr))
            if len(tickets) > 2:
                s_.wait(tickets.pop(0))
        for t_ in tickets:
            s_.wait(t_)
    run(3)
    secs, ok = [], True
    for _ in range(3):  # the copy engines may still be busy wiping the gigabytes freed just before (cross-check table)
        s_.reset()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        run(steps)
        g, u = s_.end()
        secs.append(time.perf_counter() - t0)
        ok = ok and int(g.sum()) == steps * n_reads
    s_.close()
    for p_ in pins + outs:
        p_.close()
    sec = sorted(secs)[1]
    return {"pairs_per_s": steps * (n_reads // 2) / sec, "steps": steps, "GBps_h2d": steps * nbytes / sec / 1e9,
            "counts_add_up": ok, "pairs_per_s_each_repetition": [steps * (n_reads // 2) / x for x in secs],
            "what": "kid_classify_fixed_async from pinned host memory (kid_host_alloc), 3 batches in flight, per-read results "
                    "copied back; sample closed (ucount) inside the timed region; median of 3 repetitions of %d steps" % steps}


def cli_e2e_leg(threads):
    """nk10 (the reference's command line) on directories of FASTQ.gz pairs: files in -> _result.txt/_reads.txt out, one
    sample and two samples of 1 M pairs each.  Host-bound: zlib inflates a stream at ~0.4 GB/s of text, everything behind
    it (finding the lines; trimming and classification on the GPU) is faster.  A small DB scale, so that the text DB load
    is not what is measured; the files come from tools/kid_synth_files.cpp."""
    import subprocess
    import tempfile
    from kmer_id_amd import _build
    nk10 = _build.cli_path("nk10")
    tool = _build.cli_path("kid_synth_files")
    if not os.path.exists(nk10) or not os.path.exists(tool):
        return None
    P = 1_000_000
    cwd = tempfile.mkdtemp(prefix="kid_e2e_")
    try:
        parent, cnt = synth.load_taxonomy("bact10")
        os.makedirs(os.path.join(cwd, "bact10"))
        with open(os.path.join(cwd, "counts.txt"), "w") as fh:
            fh.write("".join("%d,%d\n" % (t, c) for t, c in enumerate(cnt.tolist())))
        tree = os.path.join(cwd, "bact10", "btree_10.txt")
        with open(tree, "w") as fh:
            fh.write("".join("%d\t%d\n" % (x, y) for y, x in enumerate(parent.tolist()) if y >= 2 and x != 1))
        open(os.path.join(cwd, "bact10", "bData10.txt"), "w").write("4\tX\n")
        counts = os.path.join(cwd, "counts.txt")
        subprocess.check_call([tool, "probes", "--counts", counts, "--scale", "1e-3", "--out", os.path.join(cwd, "bact10", "probes10.txt.gz")],
                              stderr=subprocess.DEVNULL)
        dirs = {}
        for name, S in (("one_sample", 1), ("two_samples", 2)):
            d = os.path.join(cwd, name) + "/"
            os.makedirs(d)
            subprocess.check_call([tool, "fastq", "--counts", counts, "--tree", tree, "--scale", "1e-3", "--out-dir", d, "--samples", str(S),
                                   "--pairs", str(P)], stderr=subprocess.DEVNULL)
            dirs[name] = (d, S)
        empty = os.path.join(cwd, "empty") + "/"
        os.makedirs(empty)
        cache = os.path.join(cwd, "db.kidx")

        def run(d):
            t = time.perf_counter()
            r = subprocess.run([nk10, d, "--log2-slots", "22", "--db-cache", cache, "--threads", str(threads), "--timing"], cwd=cwd,
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            tm = [json.loads(l)["nk10_timing"] for l in r.stderr.decode("latin-1").splitlines() if l.startswith('{"nk10_timing"')]
            return time.perf_counter() - t, (tm[0] if tm else None)
        run(empty)
        base = min(run(empty)[0] for _ in range(2))
        out = {"startup_s": base, "reader_threads": threads, "pairs_per_sample": P,
               "what": "nk10 <dir> on 1 and on 2 samples x %d pairs of 150 bp FASTQ.gz (mixed qualities; process_qual and "
                       "classification on the GPU), wall minus the program's startup on an empty directory; bounded by the "
                       "inflate thread: one stream per file, the two mates of a sample at the same time" % P}
        for name, (d, S) in dirs.items():
            wall, tm = run(d)
            g = sum(int(line.split(",")[1]) for line in open(d + "S0_result.txt"))
            by_clock = None
            if tm and tm.get("samples"):   # the program's own clock: first sample begun -> last result written
                t0_ = min(x["begin_s"] for x in tm["samples"]); t1_ = max(x["result_written_s"] for x in tm["samples"])
                by_clock = S * P / max(t1_ - t0_, 1e-6)
            out[name] = {"pairs_per_s": S * P / max(wall - base, 1e-6), "pairs_per_s_by_program_clock": by_clock, "wall_s": wall, "reads_counted_sample0": g,
                         "stages": None if tm is None else {k_: tm[k_] for k_ in ("consumer_waited_for_host_stages_s", "consumer_waited_for_gpu_s",
                                                                                  "consumer_submit_s", "gpu_ready_at_s", "total_s", "files", "samples") if k_ in tm}}
        out["pairs_per_s"] = out["two_samples"]["pairs_per_s"]
        return out
    finally:
        import shutil
        shutil.rmtree(cwd, ignore_errors=True)


def kernel_source_sha():
    """what a counter profile is tied to: the text of the kernels it was taken on"""
    import hashlib
    h = hashlib.sha256()
    for rel in ("kmer_id_amd/csrc/kid_kernels.hip.h", "kmer_id_amd/csrc/kid_api.hip", "kmer_id_amd/csrc/kid_common.h"):
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


PMC_PROFILES = {"1m": ["profiles/r03/pmc_final.json"], "roofline100m": ["profiles/r03/pmc_100m.json"]}


def traffic_from_profile(args, info):
    """HBM bytes per classify launch from the committed rocprofv3 counter passes (profiles/):
    TCC_EA0_RDREQ_128B x 128 B + 64-byte requests + WRITE_SIZE.  Only reported when the profile
    was taken on this exact workload; the live run does not collect counters (rocprofv3 --pmc is a
    separate pass over the same command).  -> (bytes or None, where it came from)"""
    if (args.scale != 1.0 or args.pairs != CONFIGS[args.config]["pairs"] or args.log2_slots != 30 or
            args.geometry != "minloc" or args.read_len != 150):
        return None, "none: not the profiled workload"
    sha = kernel_source_sha()
    why = "none: no committed counter profile for this workload"
    for rel in PMC_PROFILES.get(args.config, []):
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
            # a profile of other kernels says nothing about these: the summary carries the hash of the sources it was taken on
            if d.get("kernel_source_sha256_16") != sha:
                why = "none: %s was taken on other kernel sources (%s, these are %s): rerun tools/pmc_traffic.sh" % (
                    rel, d.get("kernel_source_sha256_16"), sha)
                continue
            rd = d["TCC_EA0_RDREQ_128B_sum"]["avg"] * 128 + d["TCC_EA0_RDREQ_64B_sum"]["avg"] * 64 + d["TCC_EA0_RDREQ_32B_sum"]["avg"] * 32
            return rd + d["WRITE_SIZE"]["avg"] * 1024, ("committed counter profile %s (rocprofv3 --pmc passes of this command; kernel sources "
                                                        "%s = the running ones, commit %s, kernel %s; not collected live)" % (
                                                            rel, sha, d.get("commit"), (d.get("kernel") or "")[:60]))
        except Exception as e:
            why = "none: %s unreadable (%r)" % (rel, e)
            continue
    return None, why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="1m",
                    help="1m: BASELINE configs[1], 1 M pairs per step; roofline100m: configs[2], 100 M pairs resident, one batch per step")
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--pairs", type=int, default=None, help="read pairs per rank per step (default: the config's)")
    ap.add_argument("--scale", type=float, default=1.0, help="DB scale (1.0 = 108.6 M k-mers)")
    ap.add_argument("--log2-slots", type=int, default=30)
    ap.add_argument("--cpu-reads", type=int, default=300_000, help="reads of the CPU baseline sample (0 = skip)")
    ap.add_argument("--batches", type=int, default=None, help="distinct resident read batches cycled over the steps")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the all-cores CPU baseline leg (0 = every core of this host)")
    ap.add_argument("--overlap-pack", type=int, default=0,
                    help="KID_OPT_INPUTS_READY: pack + prepare of the next resident batch run beside the classify kernels of this one "
                         "(measured: same pairs/s, the classify kernel is 5-8 %% slower while pack shares the chip: profiles/r02/ab_overlap_pack.txt)")
    ap.add_argument("--host-leg", type=int, default=1, help="N = 1: also time the host-buffer (PCIe-inclusive) path")
    ap.add_argument("--e2e-leg", type=int, default=1, help="N = 1: also time the nk10 command line on FASTQ.gz files")
    ap.add_argument("--verify-ranks", type=int, default=1,
                    help="N > 1: rank 0 re-classifies every rank's reads on its own table after the timed region and compares "
                         "the merged counters with that single-table result")
    ap.add_argument("--gather", type=int, default=1, help="also measure the random 16-byte gather ceiling")
    ap.add_argument("--step-events", type=int, default=0, help="1: events around every step as well as around every classify kernel")
    ap.add_argument("--read-len", type=int, default=150, help="read length (configs[1]: 150; configs[4]: 250)")
    ap.add_argument("--geometry", choices=["minloc", "ref"], default="minloc",
                    help="table placement: minimizer-localised (default) or the reference's fmix64/triangular one")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                                                      "several ranks on one GPU: the merge then goes through host memory)")
    ap.add_argument("--xcheck", type=int, default=1,
                    help="classify the first batch on a second table built with the reference's geometry too: "
                         "full-size parity of the two placements + the reference-geometry probe count")
    args = ap.parse_args()
    for k_, v_ in CONFIGS[args.config].items():
        if getattr(args, k_) is None:
            setattr(args, k_, v_)
    global READ_LEN
    READ_LEN = args.read_len

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no HIP device is visible (there is no CPU fallback)")
    dev_index = local_rank % torch.cuda.device_count() if args.backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    n_reads = 2 * args.pairs  # R1 + R2, classified independently (newkmer_10nx.cpp:1029-1031)
    want_cpu = (rank == 0 and world == 1 and args.cpu_reads > 0)
    log("building bact10-synth DB (scale %g, 2^%d cells) ..." % (args.scale, args.log2_slots))
    flags = kmer_id_amd.KID_FLAG_REF_GEOMETRY if args.geometry == "ref" else 0
    want_x = bool(args.xcheck) and args.geometry == "minloc" and rank == 0
    db, parent, cum, build_s, host_keys, dev_keys = build_db(device, args.scale, args.log2_slots, want_cpu, flags, want_x)
    info = db.info
    log("DB: %d entries, %d cells occupied, table %.1f GiB, GPU build %.2f s" % (info.n_entries, info.n_occupied,
                                                                                 info.table_bytes / 2**30, build_s))
    nb = max(1, min(args.batches, args.steps + args.warmup))
    batches = [gen_reads(device, cum, parent, (rank * nb + b) * n_reads, n_reads) for b in range(nb)]
    if os.environ.get("KID_BENCH_RANDOM_READS") == "1":  # development aid: reads without a single database k-mer (not the metric's workload)
        lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
        batches = [lut[torch.randint(0, 4, (bt.numel(),), device=device)] for bt in batches]
    out_final = torch.empty(n_reads, dtype=torch.int32, device=device)
    sample = db.sample()
    # the batches are resident in HBM and final: the library may pack batch i + 1 while batch i is being classified
    if args.overlap_pack:
        sample.set_option(kmer_id_amd.KID_OPT_INPUTS_READY, 1)
    stream = torch.cuda.current_stream(device)

    def step(i):
        sample.classify_fixed_device(batches[i % nb].data_ptr(), READ_LEN, n_reads, d_out=out_final.data_ptr(),
                                     stream=stream.cuda_stream)

    xcheck = None
    ref_probes_per_lookup = None
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(device)
    sample.reset()
    sample.set_timing(True)   # HIP events around every kid_classify_kernel launch, on the launch stream
    sample.kernel_time()
    sample.kernel_time_device()

    # (--step-events 1: a second pair of events around every step, from this side of the C ABI.  A step is one kernel
    # since round 3, so they say what the library's own pair says -- and four event packets between two kernels cost
    # more than two.)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps if args.step_events else 0)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if ev:
            ev[i][0].record(stream)
        step(i)
        if ev:
            ev[i][1].record(stream)
    t_queued = time.perf_counter()
    # close the sample: ucount from the seen-bitmap (+ RCCL merge over the ranks)
    merge_timing = {}
    if world > 1:
        g, u = merge_sample(sample, device if args.backend == "nccl" else torch.device("cpu"), timing=merge_timing)
    else:
        g, u = sample.end()
    torch.cuda.synchronize(device)
    t_closed = time.perf_counter()
    if world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    classify_ms, classify_launches = sample.kernel_time()  # the dominant kernel (HIP events around every launch of it, on the launch stream)
    kernel_ms = [a.elapsed_time(b) for a, b in ev] if ev else [classify_ms]
    dev_ms, dev_launches = sample.kernel_time_device()      # the same launches on the device's own clock (no event overhead)
    assert classify_launches == args.steps
    st = sample.stats()
    total_reads = st["reads"]
    assert total_reads == args.steps * n_reads, (total_reads, args.steps * n_reads)
    if os.environ.get("KID_BENCH_ABLATION") != "1":  # (timing experiments with builds whose counters are wrong on purpose)
        assert int(g.sum()) == args.steps * n_reads * world, "gcount does not add up to the reads processed"
    # full-size cross-check (untimed, AFTER the timed region: freeing its 16 GiB table sets the driver wiping VRAM, which
    # took 5-10 % out of the steps that followed when the check ran first): the same batch through a table with the
    # reference's cell placement must give the same per-read targets and counters; its probe count is the "algorithmic"
    # number of cells the reference's own table would have read.
    if want_x:
        d_keys, d_targets, n_keys = dev_keys
        rdb = KmerDB.from_device(d_keys.data_ptr(), d_targets.data_ptr(), n_keys, parent, k=K, log2_slots=args.log2_slots,
                                 flags=kmer_id_amd.KID_FLAG_REF_GEOMETRY, device=device.index)
        del d_keys, d_targets, dev_keys
        rs = rdb.sample()
        xs = db.sample()
        out_ref = torch.empty(n_reads, dtype=torch.int32, device=device)
        rs.classify_fixed_device(batches[0].data_ptr(), READ_LEN, n_reads, d_out=out_ref.data_ptr(), stream=stream.cuda_stream)
        xs.classify_fixed_device(batches[0].data_ptr(), READ_LEN, n_reads, d_out=out_final.data_ptr(), stream=stream.cuda_stream)
        torch.cuda.synchronize(device)
        g1, u1 = xs.end()
        g2, u2 = rs.end()
        st_ref = rs.stats()
        xcheck = bool(torch.equal(out_ref, out_final) and np.array_equal(g1, g2) and np.array_equal(u1, u2))
        ref_probes_per_lookup = st_ref["probes"] / max(st_ref["lookups"], 1)
        log("cross-check vs reference geometry on %d reads: %s (reference geometry reads %.4f cells per lookup)" % (
            n_reads, "identical" if xcheck else "MISMATCH", ref_probes_per_lookup))
        rs.close(); xs.close(); rdb.close(); del out_ref
        torch.cuda.empty_cache()
        if not xcheck:
            raise SystemExit("minimizer-localised and reference geometries disagree")
    # N > 1: where the time of every rank went (so that a scaling record shows WHERE scaling is lost): its own steps,
    # closing its sample + the merge phases, waiting for the slowest rank
    per_rank = None
    if world > 1:
        mine = {"rank": rank, "steps_gpu_ms": sum(kernel_ms), "classify_kernel_ms": classify_ms, "queued_after_s": t_queued - t0,
                "closed_after_s": t_closed - t0, "barrier_wait_s": t1 - t_closed, **{k_: round(v_, 6) for k_, v_ in merge_timing.items()}}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        per_rank = gathered
    kern_s = sum(kernel_ms) / 1e3
    # algorithmic bytes: 16 B per table cell the REFERENCE's table geometry reads for
    # these lookups (counted exactly on the reference-geometry table above), whatever our own
    # placement needs; without the cross-check, our own cell count stands in.
    cells_read_per_launch = st["probes"] / args.steps
    if ref_probes_per_lookup is not None:
        probes_per_launch = ref_probes_per_lookup * st["lookups"] / args.steps
    else:
        probes_per_launch = cells_read_per_launch
    avg_step_gpu_s = kern_s / args.steps
    avg_kernel_s = classify_ms / 1e3 / args.steps
    # ... + the read text: since round 3 the classify kernel reads the ASCII bases itself (there is no pack kernel in front
    # of it any more), one byte per base of every read, like the reference's loop does (newkmer_10nx.cpp:475-477)
    table_bytes_per_launch = probes_per_launch * 16
    text_bytes_per_launch = float(n_reads) * READ_LEN
    algorithmic_bytes = table_bytes_per_launch + text_bytes_per_launch
    achieved = algorithmic_bytes / avg_kernel_s / 1e9

    extra = {}
    if args.gather and rank == 0:
        # what the memory system gives a kernel that does nothing but ask for random 128-byte lines of THIS table with
        # the classify kernel's load (one 16-byte header per lane, runs of 8 lanes on a line / 64 lines per load)
        for name, code in (("runs_of_8", 108), ("64_per_load", 101)):
            ms, lines = db.gather_ceiling(n_loads=1 << 29, inflight=code, iters=3)
            extra["random_lines_%s_Glines_per_s" % name] = round(lines / (ms / 1e3) / 1e9, 2)
            extra["random_lines_%s_GBps_of_128B" % name] = round(lines * 128 / (ms / 1e3) / 1e9, 1)

    host_leg = e2e_leg = None
    if rank == 0 and world == 1 and args.config == "1m":
        if args.host_leg:
            log("host-buffer leg ...")
            host_leg = host_buffer_leg(db, device, batches, n_reads, max(4, min(args.steps, 12)))
        if args.e2e_leg:
            log("nk10 FASTQ.gz leg ...")
            try:
                e2e_leg = cli_e2e_leg(min(16, host_cores()))  # (one GPU's share of the pool's hosts)
            except Exception as e:  # the CLI leg must not take the metric down with it
                e2e_leg = {"error": repr(e)}

    cpu = None
    if want_cpu:
        # "all host cores" = this GPU's share of the box: the pool's hosts have 256 hardware threads for 8 GPUs
        threads = args.cpu_threads or min(host_cores(), 32)
        log("CPU baseline: oracle, 1 thread and %d threads, %d reads ..." % (threads, args.cpu_reads))
        cpu, (cg, cu) = cpu_baseline(host_keys, parent, cum, args.log2_slots, args.cpu_reads, threads)
        # the same reads through the GPU path must give the same counts
        chk = db.sample()
        hb = synth.reads(cum, parent, args.cpu_reads, READ_LEN, K)
        chk.classify(hb, synth.fixed_offsets(args.cpu_reads, READ_LEN), want_final=False)
        gg, gu = chk.end()
        cpu["gpu_equals_cpu_on_sample"] = bool(np.array_equal(gg, cg) and np.array_equal(gu, cu))
        chk.close()

    # N > 1: the merged counters against ONE table that saw every rank's reads (rank 0's; untimed)
    ranks_verified = None
    if world > 1 and args.verify_ranks:
        if rank == 0:
            chk = db.sample()
            for r in range(world):
                rb = [gen_reads(device, cum, parent, (r * nb + b_) * n_reads, n_reads) for b_ in range(nb)] if r else batches
                for i in range(args.steps):
                    chk.classify_fixed_device(rb[i % nb].data_ptr(), READ_LEN, n_reads, stream=stream.cuda_stream)
                torch.cuda.synchronize(device)
                if r:
                    del rb
            g1, u1 = chk.end()
            chk.close()
            ranks_verified = bool(np.array_equal(g1, g) and np.array_equal(u1, u))
            log("merged counters of %d ranks vs one table over all their reads: %s" % (world, "identical" if ranks_verified else "MISMATCH"))
        dist.barrier()
        if rank == 0 and not ranks_verified:
            raise SystemExit("multi-rank merge disagrees with the single-table result")

    if rank == 0:
        traffic, traffic_source = traffic_from_profile(args, info)
        pairs_total = args.steps * args.pairs * world
        line = {
            "metric": "paired reads classified/sec on bact10 DB",
            "value": pairs_total / elapsed,
            "unit": "paired reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "bact10-synth DB (%d 30-mers on the real bact10 taxonomy, 2^%d-cell table resident in HBM), "
                                   "%d synthetic %d bp read pairs per GPU per step, reads resident in HBM" % (
                                       info.n_entries, args.log2_slots, args.pairs, READ_LEN),
                       "name": args.config, "inputs_ready_option": bool(args.overlap_pack), "db_scale": args.scale, "pairs_per_gpu_per_step": args.pairs, "read_len": READ_LEN, "k": K,
                       "merged_counters_equal_single_table": ranks_verified, "per_rank": per_rank,
                       "slowest_rank": None if per_rank is None else max(per_rank, key=lambda r_: r_["closed_after_s"])["rank"],
                       "sharding": "reads sharded over %d rank(s), DB replicated" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "kid_classify_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "avg_kernel_ms_device_clock": (dev_ms / dev_launches) if dev_launches else None,
                         "kernel_time_source": "HIP events on the launch stream around the batch's kid_classify_kernel launch (a "
                                               "fixed-layout batch is ONE kernel: it reads the ASCII text, no pack / prepare kernels)",
                         "avg_step_gpu_ms": avg_step_gpu_s * 1e3,
                         "algorithmic_bytes_per_launch": algorithmic_bytes,
                         "algorithmic_table_bytes_per_launch": table_bytes_per_launch, "algorithmic_text_bytes_per_launch": text_bytes_per_launch,
                         "frac_table_bytes_only": table_bytes_per_launch / avg_kernel_s / 1e9 / HBM_PEAK_GBPS,
                         "kernel_source_sha256_16": kernel_source_sha(),
                         "lookups_per_launch": st["lookups"] / args.steps, "probes_per_launch": probes_per_launch,
                         "cells_read_per_launch": cells_read_per_launch, "geometry": args.geometry,
                         "fullsize_parity_vs_reference_geometry": xcheck,
                         "lookups_per_s": st["lookups"] / (classify_ms / 1e3), "kernel_only_pairs_per_s": args.pairs / avg_kernel_s,
                         **extra},
            "cpu_baseline": cpu,
            "host_buffer_path": host_leg,
            "cli_fastq_gz": e2e_leg,
        }
        print(json.dumps(line), flush=True)
    sample.close()
    db.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
