#!/bin/bash
# round 5: the one-term window, timing only (argument: shapes; NFM_* from the environment)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/seqwin_time.py ${N:-400000} ${WS:-64,128} ${1:-cfg2,headline,nodep64,nodep32} 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r5_seqwin_time.log
