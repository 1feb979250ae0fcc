#!/bin/bash
# usage: tools/sweep.sh "workload n batch" ...   (runs bench.py per config, prints one summary line each)
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload $1 --n $2 --batch $3 > gpurun_out/b.log 2>&1 || { tail -5 gpurun_out/b.log; exit 1; }
  python - "$cfg" <<'PY'
import json, sys
l=[x for x in open("gpurun_out/b.log") if x.startswith("{")][-1]
j=json.loads(l); r=j["roofline"]
print(sys.argv[1], "value %.3e"%j["value"], "ms/step %.4f"%j["ms_per_step"], {k:v for k,v in r["avg_ms"].items() if v}, "frac", r["frac"], "loss %.5f"%j["last_step"]["mean_loss"])
PY
done
