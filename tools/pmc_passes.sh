#!/bin/bash
# rocprofv3 counter passes over bench.py (one pass per counter group; never combined with tracing).
# usage: tools/pmc_passes.sh <tag> [bench args...]   -> gpurun_out/pmc_<tag>/<pass>/...
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 --gather 0 --xcheck 0 $*"
run() { name=$1; shift; timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- $B > $OUT/$name.log 2>&1; echo "$name rc=$?"; }
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B > $OUT/trace.log 2>&1; echo "trace rc=$?"
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY
run sq_b SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM
run tcc_a TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum
run tcc_b TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcp_a TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum
run grbm GRBM_GUI_ACTIVE
