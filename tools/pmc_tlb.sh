#!/bin/bash
# address-translation counters of the classify kernel: tools/pmc_tlb.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_tlb_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 3 --warmup 1 --cpu-reads 0 --gather 0 --xcheck 0 $*"
timeout -k 10 240 rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum --output-format csv -d $OUT/tlb_a -- $B > $OUT/tlb_a.log 2>&1
timeout -k 10 240 rocprofv3 --pmc TCP_UTCL1_SERIALIZATION_STALL TCP_UTCL1_THRASHING_STALL TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_UTCL1_STALL_INFLIGHT_MAX GRBM_GUI_ACTIVE --output-format csv -d $OUT/tlb_b -- $B > $OUT/tlb_b.log 2>&1
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.json
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json"))
for k,v in d.items():
    if isinstance(v,dict) and "avg" in v: print(k, v["avg"])
PY
