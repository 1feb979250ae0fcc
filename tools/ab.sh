#!/bin/bash
# A/B runs of bench.py over alternative builds of the library (NIMFM_HIP_LIB).
# usage: tools/ab.sh "workload n batch" lib1.so lib2.so ...   (libs relative to nimfm_amd/lib/)
set -e
read WL N B <<< "$1"; shift
mkdir -p gpurun_out
for L in "$@"; do
  NIMFM_HIP_LIB=$(pwd)/nimfm_amd/lib/$L python3 bench.py --workload $WL --n $N --batch $B --no-cpu-baseline --no-extra > gpurun_out/ab_tmp.json
  python3 - "$L" "$WL" "$B" <<'PY'
import json, sys
j = json.load(open("gpurun_out/ab_tmp.json"))
r = j["roofline"]
print("%-24s %-9s B=%-6s %.4g samples/s frac %.4f shuffled %.4g (%.2fx)  %s" % (sys.argv[1], sys.argv[2], sys.argv[3], j["value"], r["frac"],
      j.get("value_shuffled") or 0.0, (j.get("value_shuffled") or 0.0) / j["value"],
      {k: round(v * 1e3, 1) for k, v in r["avg_ms"].items() if v}))
PY
done
