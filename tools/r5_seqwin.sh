#!/bin/bash
# round 5: the one-term window -- parity tests of both flavours, then samples/s by worker count and row shape
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_seqwin.py -x -q > gpurun_out/r5_seqwin_tests.log 2>&1 || { tail -30 gpurun_out/r5_seqwin_tests.log; exit 1; }
tail -3 gpurun_out/r5_seqwin_tests.log
for ex in 0 1; do
  echo "== NFM_SEQ_WIN_EXACT=$ex" | tee -a gpurun_out/r5_seqwin_time.log
  NFM_SEQ_WIN_EXACT=$ex timeout -k 10 300 python tools/seqwin_time.py 400000 64,128 cfg2,headline,nodep64,nodep32 2>&1 | tee -a gpurun_out/r5_seqwin_time.log || exit 1
done
