#!/bin/bash
# SQ / L2 counters per kernel for one bench configuration (single --pmc pass, no other tracing).
# usage: tools/pmc_sq.sh TAG WORKLOAD N BATCH "COUNTER LIST"
set -e
TAG=$1; WL=$2; N=$3; B=$4; CTRS=$5
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/${TAG}_sq -o s -- python3 $ROOT/bench.py --workload $WL --n $N --batch $B --no-cpu-baseline --steps 1 --warmup 1 > $OUT/${TAG}_sq.log 2>&1
cd $ROOT
python3 - "$(find $OUT/${TAG}_sq -name '*counter_collection.csv' | head -1)" <<'PY'
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    key = next((k for k in ("k_row_phase", "k_col_phase", "k_heavy_partial", "k_heavy_apply", "k_fm_predict") if k in n), None)
    if key:
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
rm -rf $OUT/${TAG}_sq
