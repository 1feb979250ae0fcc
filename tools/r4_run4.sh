python -m pytest tests/test_gpu_dp.py -x -q -m gpu > gpurun_out/r4_dp_tests.log 2>&1; tail -n 2 gpurun_out/r4_dp_tests.log
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 NIMFM_BENCH_FORCE_DP=1
for g in 0 1; do
NFM_DP_GRAPH=$g python bench.py --workload headline --no-cpu-baseline --no-extra --no-t2t --steps 5 --warmup 2 2> gpurun_out/r4_dp_$g.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp_graph=$g headline', d['value'], d['ms_per_step'], d['dp'])"
NFM_DP_GRAPH=$g python bench.py --workload cfg3 --n 4000000 --no-cpu-baseline --no-extra --no-t2t --steps 5 --warmup 2 2> gpurun_out/r4_dp3_$g.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dp_graph=$g cfg3', d['value'], d['ms_per_step'], d['dp'])"
done
unset NIMFM_BENCH_FORCE_DP
python bench.py --workload headline --no-cpu-baseline --no-extra --no-t2t --steps 5 --warmup 2 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no dp headline', d['value'], d['ms_per_step'])"
