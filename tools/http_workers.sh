export METRICS_LOG_LEVEL=WARNING
for cfg in "1 8 4" "2 8 4" "2 10 4" "3 10 4"; do
  set -- $cfg
  python tools/http_load.py --gpu-workers $1 --frontends $2 --client-procs $3 --clients 256 --seconds 5 --port $((18100 + $1 * 10 + $2)) 2>/dev/null | tail -1 | python3 -c "
import sys, json; d = json.loads(sys.stdin.read()); print('gpu_workers', d['gpu_workers'], 'frontends', d['frontends'], 'qps', d['qps'], 'p50', d['p50_ms'], 'p99', d['p99_ms'], 'failed', d['requests_failed'], d['cpus_busy_per_process'])"
done
