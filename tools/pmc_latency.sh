#!/bin/bash
# Is the memory system saturated under the classify kernel?  Average read latencies (level / requests) and stall counters of
# the L2 and of the vector L1, for the classify kernel and for the line-gather kernel on the same table in the same kind
# of process (tools/gather_in_process.py).  One rocprofv3 counter pass per group, no tracing.
# usage: tools/pmc_latency.sh <tag>  -> gpurun_out/pmc_lat_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_lat_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
    prog=$1; name=$2; shift; shift
    if [ $prog = classify ]; then B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 --gather 0 --xcheck 0 --host-leg 0 --e2e-leg 0"; else B="python3 $R/tools/gather_in_process.py"; fi
    timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $OUT/${prog}_$name -- $B > $OUT/${prog}_$name.log 2>&1
    echo "$prog $name rc=$? : $*"
}
for prog in classify gather; do
    pass $prog ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum
    pass $prog tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
    pass $prog tcc TCC_REQ_sum TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_TAG_STALL_sum
done
python3 - $OUT <<'PY'
import collections, csv, glob, os, sys
root = sys.argv[1]
for d in sorted(os.listdir(root)):
    p = os.path.join(root, d)
    if not os.path.isdir(p):
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(p + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "classify_kernel" in k or "gather_lines" in k:
                per[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in per.items():
        tot = sum(sum(v) for v in c.values())
        if tot == 0:
            continue
        print(d, "|", k, "|", "  ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items())), "| dispatches", len(next(iter(c.values()))))
PY
