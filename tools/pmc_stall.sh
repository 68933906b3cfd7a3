#!/bin/bash
# Where does the classify kernel stall?  Instruction-cache, LDS and vector-memory-pipe counters, one rocprofv3 pass per
# group (counters only, never combined with tracing).  Counter names are taken from `rocprofv3 -L` on the box; a name the
# box does not list is dropped from its group.
# usage: tools/pmc_stall.sh <tag> [bench args...]   -> gpurun_out/pmc_stall_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_stall_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 --gather 0 --xcheck 0 --host-leg 0 --e2e-leg 0 $*"
pass() {
    name=$1; shift
    have=""
    for c in "$@"; do
        if grep -qw "$c" $OUT/counters_list.txt; then have="$have $c"; else echo "$name: no counter $c"; fi
    done
    [ -z "$have" ] && return
    timeout -k 10 240 rocprofv3 --pmc $have --output-format csv -d $OUT/$name -- $B > $OUT/$name.log 2>&1
    echo "$name rc=$? :$have"
}
pass ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
pass sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_STALL SQ_INSTS_SMEM SQ_WAIT_ANY
pass sq_vm SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT
pass tcp_b TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass tcp_c TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
# (TA_* / TD_* groups: every pass with them ran into its time limit on this pool's rocprofv3 (profiles/r02/pmc_stall.txt) -- not collected)
pass tcc_c TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.json 2>$OUT/summary.err
python3 - <<PY
import json
try:
    d = json.load(open("$OUT/summary.json"))
    for k, v in sorted(d.items()):
        if isinstance(v, dict) and "avg" in v:
            print(k, v["avg"])
except Exception as e:
    print("summary failed", e)
PY
