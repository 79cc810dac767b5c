for b in 1024 1030 1040 1050 1060; do
  for tm in 512 3584; do
    echo -n "batch $b tail_m $tm: "
    ICREC_TAIL_M=$tm python3 bench.py --no-cpu-baseline --no-latency --steps 30 --warmup 5 --batch $b 2>/dev/null | python3 -c "
import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'], 4), 'ms', round(d['value']), 'QPS')"
  done
done
