#!/bin/bash
# Produces the judged artefacts for one bench configuration on the GPU box:
#   gpurun_out/<tag>_bench.json            bench.py's JSON line (with cpu_baseline)
#   gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary of the same command
#   gpurun_out/<tag>_pmc_{fetch,write}/    separate --pmc passes (FETCH_SIZE, WRITE_SIZE)
#   profiles/<tag>_pmc_traffic.json        via tools/pmc_traffic.py (copy back from gpurun_out/)
# usage: tools/profile.sh TAG WORKLOAD BATCH [extra bench.py flags]
set -e
TAG=$1; WL=$2; B=$3; shift 3
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python3 $ROOT/bench.py --workload $WL --batch $B --no-cpu-baseline --no-extra --no-t2t --no-shuffled --no-exact "$@" > $OUT/${TAG}_prof.log 2>&1
cp $(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -o f -- python3 $ROOT/bench.py --workload $WL --batch $B --no-cpu-baseline --no-extra --no-t2t --no-shuffled --no-exact --steps 2 --warmup 1 "$@" > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -o w -- python3 $ROOT/bench.py --workload $WL --batch $B --no-cpu-baseline --no-extra --no-t2t --no-shuffled --no-exact --steps 2 --warmup 1 "$@" > $OUT/${TAG}_pmc_write.log 2>&1
cd $ROOT
python3 tools/pmc_traffic.py $(find $OUT/${TAG}_pmc_fetch -name "*counter_collection.csv" | head -1) \
  $(find $OUT/${TAG}_pmc_write -name "*counter_collection.csv" | head -1) $WL $B $TAG
cp profiles/${TAG}_pmc_traffic.json $OUT/
# the bench line last, so that its roofline.traffic is this run's PMC result
python3 bench.py --workload $WL --batch $B --no-extra --no-t2t "$@" > $OUT/${TAG}_bench.json
cat $OUT/${TAG}_bench.json
# the raw per-dispatch counter CSVs are large; keep only the summary
rm -rf $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
find $OUT/${TAG}_prof -name "*kernel_trace.csv" -delete
