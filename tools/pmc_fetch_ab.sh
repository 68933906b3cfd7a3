#!/bin/bash
# FETCH_SIZE (the two derived counters do not fit one pass: error 38, and the aborted tool then hangs) of the classify kernel for several builds of the library (one counter pass each, no tracing).
# usage: tools/pmc_fetch_ab.sh <tag> lib1.so lib2.so ...   -> gpurun_out/pmcab_<tag>/<lib>/..., one line per lib
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmcab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  export KMER_ID_AMD_LIB=$R/$lib
  timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/$name -- python3 $R/bench.py --cpu-reads 0 --gather 0 --xcheck 0 --host-leg 0 --e2e-leg 0 --steps 10 --warmup 2 > $OUT/$name.log 2>&1
  echo "$name rc=$?"
  python3 - $OUT/$name <<'PY'
import csv, glob, sys
tot = {}
n = 0
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "classify" in r["Kernel_Name"]:
            tot.setdefault(r["Kernel_Name"][:40] + " " + r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in tot.items():
    print("   %s: %d launches, mean %.4g (x 1 KiB... see guide: FETCH_SIZE is in KiB units)" % (k, len(v), sum(v) / len(v)))
PY
done
