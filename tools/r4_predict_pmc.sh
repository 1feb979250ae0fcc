# PMC traffic of decisionFunction on cfg5 (two orders): the one-order-at-a-time kernel (NFM_PREDICT_ORDERS=0) and the interleaved one
set -x
R=$(pwd); export TMPDIR=/tmp; cd /tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    NFM_PREDICT_ORDERS=$v rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/r04d_pred${v}_$c -o p -- python3 $R/bench.py --workload cfg5 --no-cpu-baseline --no-extra --no-t2t --steps 1 --warmup 1 > $R/gpurun_out/r04d_pred${v}_$c.log 2>&1
  done
  cd $R
  python3 tools/pmc_traffic.py $(find gpurun_out/r04d_pred${v}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find gpurun_out/r04d_pred${v}_WRITE_SIZE -name "*counter_collection.csv" | head -1) cfg5 32768 r04d_cfg5_predict_orders$v
  cp profiles/r04d_cfg5_predict_orders$v\_pmc_traffic.json gpurun_out/
  rm -rf gpurun_out/r04d_pred${v}_FETCH_SIZE gpurun_out/r04d_pred${v}_WRITE_SIZE
  cd /tmp
done
