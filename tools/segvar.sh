for s in 256 512; do
echo "== NFM_SEG_S=$s"
NFM_SEG_S=$s NFM_PLAN_PREFETCH=0 python tools/shuffle_cost.py headline 4000000 2>&1 | grep "fresh\|plan_"
done
for wl in headline cfg3; do
  echo "== $wl"
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['value_shuffled'], round(j['value_shuffled']/j['value'],3), j['value_shuffled_host_perm'], j['roofline']['avg_ms'])"
done
