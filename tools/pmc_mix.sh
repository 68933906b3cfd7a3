#!/bin/bash
# Instruction mix and issue-port occupancy of the classify kernel: is the hot loop bound by VALU issue?  One rocprofv3
# counter pass per group (SQ counters only; no tracing).  usage: tools/pmc_mix.sh <tag> [bench args...] -> gpurun_out/pmc_mix_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_mix_$TAG
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
    timeout -k 10 150 rocprofv3 --pmc $have --output-format csv -d $OUT/$name -- $B > $OUT/$name.log 2>&1
    echo "$name rc=$? :$have"
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAVES
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_CYCLES
pass cyc SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass dep SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAVES_EQ_64 SQ_INSTS_VSKIPPED
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.json 2>$OUT/summary.err
python3 - <<PY
import json
d = json.load(open("$OUT/summary.json"))
for k, v in sorted(d.items()):
    if isinstance(v, dict) and "avg" in v:
        print(k, v["avg"])
PY
