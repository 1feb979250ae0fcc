A='{"n_t": 1000000, "planted_P": 0.05, "adagrad": {"eta0": %s, "alpha0": 1e-6, "alpha": %s, "beta": %s}}'
python tools/t2t_gpu.py cfg3 "$(printf "$A" 0.1 1e-5 1e-5)" "$(printf "$A" 0.1 1e-4 1e-4)" "$(printf "$A" 0.1 1e-3 1e-3)" "$(printf "$A" 0.03 1e-4 1e-4)" > gpurun_out/t2t_cfg3.log 2>&1
B='{"adagrad": {"eta0": %s, "alpha0": 1e-6, "alpha": %s, "beta": %s}}'
python tools/t2t_gpu.py cfg4 "$(printf "$B" 0.1 1e-5 1e-5)" "$(printf "$B" 0.1 1e-4 1e-4)" "$(printf "$B" 0.1 1e-3 1e-3)" "$(printf "$B" 0.03 1e-4 1e-4)" > gpurun_out/t2t_cfg4.log 2>&1
C='{"sgd": {"eta0": %s, "alpha0": 1e-6, "alpha": %s, "beta": %s}}'
python tools/t2t_gpu.py cfg5 "$(printf "$C" 0.02 1e-5 1e-5)" "$(printf "$C" 0.005 1e-5 1e-5)" "$(printf "$C" 0.01 1e-3 1e-3)" > gpurun_out/t2t_cfg5.log 2>&1
grep -h '^{' gpurun_out/t2t_cfg3.log gpurun_out/t2t_cfg4.log gpurun_out/t2t_cfg5.log
