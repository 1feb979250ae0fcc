set -x
python -m pytest tests/test_gpu_seqwin.py -x -q -m gpu > gpurun_out/r3_tests.log 2>&1; tail -n 3 gpurun_out/r3_tests.log
python -m pytest tests/test_gpu_sequential.py tests/test_gpu_dp.py tests/test_gpu_predict.py -x -q -m gpu > gpurun_out/r3_tests2.log 2>&1; tail -n 3 gpurun_out/r3_tests2.log
for wlname in cfg2 headline; do
python bench.py --workload $wlname --n 2200000 --no-cpu-baseline --no-extra --no-t2t --steps 2 --warmup 1 2> gpurun_out/r3_$wlname.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wlname', d['exact_order'])"
done
for v in 2 4; do NFM_SPLIT=$v python bench.py --workload cfg5 --no-cpu-baseline --no-extra --no-t2t --steps 2 --warmup 1 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split=$v', d['predict'])"; done
