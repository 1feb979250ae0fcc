#!/bin/bash
# A/B of environment knobs on ONE box: usage r5_ab.sh "VAR=a VAR=b ..." shapes  (each setting runs seqwin_time once)
mkdir -p gpurun_out
for kv in $1; do
  echo "== $kv" | tee -a gpurun_out/r5_seqwin_time.log
  env $kv N=${N:-400000} WS=${WS:-64} bash tools/r5_seqwin_quick.sh $2 || exit 1
done
