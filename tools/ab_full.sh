#!/bin/bash
# like tools/ab.sh but also prints predict
read WL N B <<< "$1"; shift
for L in "$@"; do
  NIMFM_HIP_LIB=$(pwd)/nimfm_amd/lib/$L python3 bench.py --workload $WL --n $N --batch $B --no-cpu-baseline > gpurun_out/ab_tmp.json
  python3 - "$L" "$WL" <<'PY'
import json, sys
j = json.load(open("gpurun_out/ab_tmp.json")); r = j["roofline"]
print("%-22s %-9s %.4g samples/s frac %.4f %s predict %.4g (%.3f)" % (sys.argv[1], sys.argv[2], j["value"], r["frac"], {k: round(v * 1e3, 1) for k, v in r["avg_ms"].items() if v and k != "schedule"}, j["predict"]["value"], j["predict"]["roofline_frac"]))
PY
done
