set -x
python -m pytest tests/test_gpu_predict.py tests/test_gpu_configs.py -x -q -m gpu -k "predict or cfg5" > gpurun_out/r2_tests.log 2>&1; tail -n 3 gpurun_out/r2_tests.log
for v in 0 1; do NFM_PREDICT_ORDERS=$v python bench.py --workload cfg5 --no-cpu-baseline --no-extra --no-t2t --steps 3 --warmup 1 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('orders=$v', d['predict'])"; done
A='{"n_t": 2000000, "planted_P": 0.05, "adagrad": {"eta0": %s, "alpha0": 1e-6, "alpha": %s, "beta": %s}}'
python tools/t2t_gpu.py cfg3 "$(printf "$A" 0.1 3e-5 3e-5)" "$(printf "$A" 0.05 3e-5 3e-5)" 2>&1 | grep '^{'
export TMPDIR=/tmp; R=$(pwd); cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04b_cfg4_B2048_prof -o r04b -- python3 $R/bench.py --workload cfg4 --no-cpu-baseline --no-extra --no-t2t --steps 3 --warmup 1 > $R/gpurun_out/r04b_prof.log 2>&1
cd $R; cp $(find gpurun_out/r04b_cfg4_B2048_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r04b_cfg4_B2048_kernel_stats.csv; find gpurun_out/r04b_cfg4_B2048_prof -name "*kernel_trace.csv" -delete
head -12 gpurun_out/r04b_cfg4_B2048_kernel_stats.csv | cut -c1-200
