#!/bin/bash
# round 5: the mini-batch rule at LARGER batches -- throughput + roofline per batch, then time_to_target
# over the same batches (one run: the sequential targets are computed once).  usage: [BATCHES="..." T2T=a,b] r5_batch_sweep.sh headline|cfg3 [samples]
set -o pipefail
W=${1:-headline}; NS=${2:-0}
mkdir -p gpurun_out
EXTRA=""; [ "$NS" != "0" ] && EXTRA="--samples $NS"
BATCHES=${BATCHES:-"8192 16384 32768 65536"}; T2T=${T2T:-16384,32768,65536}
for B in $BATCHES; do
  timeout -k 10 420 python bench.py --workload $W --batch $B $EXTRA --no-extra --no-t2t --no-exact --no-cpu-baseline --steps 10 --warmup 3 \
    > gpurun_out/r5_sweep_${W}_B${B}.json 2> gpurun_out/r5_sweep_${W}_B${B}.err || { tail -5 gpurun_out/r5_sweep_${W}_B${B}.err; exit 1; }
  python - <<PY
import json
r = json.loads(open("gpurun_out/r5_sweep_${W}_B${B}.json").read().strip().splitlines()[-1])
print("$W B=$B: %.4g samples/s, frac %.3f, shuffled %s, avg_ms %s" % (r["value"], r["roofline"]["frac"], r.get("value_shuffled"), r["roofline"]["avg_ms"]), flush=True)
PY
done
timeout -k 10 900 python bench.py --workload $W $EXTRA --no-extra --no-exact --no-cpu-baseline --steps 3 --warmup 1 --t2t-batches $T2T \
  > gpurun_out/r5_sweep_${W}_t2t.json 2> gpurun_out/r5_sweep_${W}_t2t.err || { tail -5 gpurun_out/r5_sweep_${W}_t2t.err; exit 1; }
cp gpurun_out/bench_detail.json gpurun_out/r5_sweep_${W}_t2t_detail.json
python - <<PY
import json
d = json.load(open("gpurun_out/r5_sweep_${W}_t2t_detail.json"))
t = d["time_to_target"]
print("sequential targets:", [(s["epochs"], s["seconds"], s["held_out_loss"], s["gap_closed"]) for s in t["sequential"]])
for r in t["minibatch"]:
    print("  B=%d: %.5f s/epoch; %s" % (r["batch"], r["seconds_per_epoch"], [(h["seq_epochs"], h["epochs"] if h["reached"] else None, h["seconds"] if h["reached"] else None, h["speedup"]) for h in r["targets"]]))
print("best_batch", t["best_batch"])
PY
