#!/bin/bash
# HBM traffic of the classify kernel: kernel trace + the memory-side counters, one rocprofv3 pass each (counters are never
# combined with tracing).  usage: tools/pmc_traffic.sh <tag> [bench args...]  -> gpurun_out/pmc_<tag>/ + summary.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --cpu-reads 0 --gather 0 --xcheck 0 --host-leg 0 --e2e-leg 0 $*"
run() { name=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $B > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1; echo "trace rc=$?"
run tcc_a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc_b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.json
tail -c 1500 $OUT/summary.json
