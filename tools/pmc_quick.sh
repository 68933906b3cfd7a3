#!/bin/bash
# quick SQ instruction-mix pass for one library build: tools/pmc_quick.sh <tag> <lib.so>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; LIB=$2; shift 2
OUT=$R/gpurun_out/pmcq_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export KMER_ID_AMD_LIB=$R/$LIB
B="python3 $R/bench.py --steps 4 --warmup 1 --cpu-reads 0 --gather 0 --xcheck 0 $*"
timeout -k 10 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $OUT/sq_a -- $B > $OUT/sq_a.log 2>&1
timeout -k 10 240 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq_b -- $B > $OUT/sq_b.log 2>&1
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.json
python3 - <<PY
import json
d=json.load(open("$OUT/summary.json")); g=lambda k: d[k]["avg"]; T=241430833.9/64
cyc=g("GRBM_GUI_ACTIVE")/8
print("$TAG: per tile VALU %.1f SALU %.1f LDS %.1f VMEM %.2f | kernel cycles %.0f | VALU busy %.2f SALU %.2f | wait_any %.2f wait_inst %.2f active %.2f" % (
 g("SQ_INSTS_VALU")/T, g("SQ_INSTS_SALU")/T, g("SQ_INSTS_LDS")/T, g("SQ_INSTS_VMEM_RD")/T, cyc,
 g("SQ_ACTIVE_INST_VALU")*4/1024/cyc, g("SQ_ACTIVE_INST_SCA")*4/1024/cyc,
 g("SQ_WAIT_ANY")/g("SQ_WAVE_CYCLES"), g("SQ_WAIT_INST_ANY")/g("SQ_WAVE_CYCLES"), g("SQ_ACTIVE_INST_ANY")/g("SQ_WAVE_CYCLES")))
PY
