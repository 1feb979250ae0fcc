#!/bin/bash
# Per-kernel cost of building a batch plan for a fresh order, without the epoch running beside it (NFM_PLAN_PREFETCH=0):
# rocprofv3 kernel statistics of `bench.py --workload WL`, reduced to the plan kernels.
# usage: tools/plan_prof.sh TAG WORKLOAD [extra bench.py flags]
set -e
TAG=$1; WL=$2; shift 2
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
export NFM_PLAN_PREFETCH=0
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_planprof -o p -- python3 $ROOT/bench.py --workload $WL --no-cpu-baseline --no-extra "$@" > $OUT/${TAG}_planprof.json 2> $OUT/${TAG}_planprof.err
cd $ROOT
python3 - "$OUT/${TAG}_planprof" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:34]:
    print("%6d %9.2f ms %9.1f us  %s" % (int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Name"][:96]))
PY
find $OUT/${TAG}_planprof -name "*kernel_trace.csv" -delete
