# rocprofv3 kernel stats of the headline workload at the driver's step counts (the per-kernel averages bench.py's roofline leg is checked against)
set -x
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04g_headline_prof -o r04g -- python3 $R/bench.py --workload headline --no-cpu-baseline --no-extra --no-t2t --no-shuffled --steps 20 --warmup 5 > $R/gpurun_out/r04g_prof.log 2>&1
cd $R; cp $(find gpurun_out/r04g_headline_prof -name "*kernel_stats.csv" | head -1) gpurun_out/r04g_headline_B8192_kernel_stats.csv; find gpurun_out/r04g_headline_prof -name "*kernel_trace.csv" -delete
head -5 gpurun_out/r04g_headline_B8192_kernel_stats.csv | cut -c1-180; tail -n 1 gpurun_out/r04g_prof.log | cut -c1-400
