set -x
bash tools/profile.sh r04c_headline_B8192 headline 8192 --steps 5 --warmup 2 > gpurun_out/r04c_headline.log 2>&1; tail -c 600 gpurun_out/r04c_headline.log
bash tools/profile.sh r04c_cfg5_B32768 cfg5 32768 --steps 5 --warmup 2 > gpurun_out/r04c_cfg5.log 2>&1; tail -c 300 gpurun_out/r04c_cfg5.log
bash tools/profile.sh r04c_cfg4_B2048 cfg4 2048 --steps 5 --warmup 2 > gpurun_out/r04c_cfg4.log 2>&1; tail -c 300 gpurun_out/r04c_cfg4.log
bash tools/profile.sh r04c_cfg3_B8192 cfg3 8192 --steps 3 --warmup 1 --n 4000000 > gpurun_out/r04c_cfg3.log 2>&1; tail -c 300 gpurun_out/r04c_cfg3.log
ls gpurun_out | grep r04c
