for v in 1 0; do
  export NFM_SORT_BY_COUNT=$v
  for wl in headline; do
  echo "== $wl SORT_BY_COUNT=$v"
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['value'], j['value_shuffled'], round(j['value_shuffled']/j['value'],3), j['value_shuffled_host_perm'], j['roofline']['avg_ms'])"
done; done
