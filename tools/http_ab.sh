#!/bin/bash
# A/B of the HTTP figure between round 2's tree (tools/_r02tree, built in the container) and this tree on ONE lease:
# the same tools/http_load.py arguments, alternated.  usage: bash tools/http_ab.sh <rounds> <out.jsonl>
R=${1:-2}; O=${2:-gpurun_out/http_ab.jsonl}
export METRICS_LOG_LEVEL=WARNING
for i in $(seq 1 $R); do
  (cd tools/_r02tree && python tools/http_load.py --frontends 8 --client-procs 4 --clients 256 --seconds 6 --port 18090 2>/dev/null | sed 's/^{/{"tree": "r02", /') >> $O
  python tools/http_load.py --frontends 8 --client-procs 4 --clients 256 --seconds 6 --port 18091 2>/dev/null | sed 's/^{/{"tree": "head", /' >> $O
done
cat $O | cut -c1-220
