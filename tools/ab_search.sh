#!/bin/bash
# Same-box A/B of the search kernels between two builds of libicrec.so (A = $ICREC_AB_A, default tools/_libicrec_r03.so;
# B = in-tree): the resident filter pass at the headline catalog and at 10 M rows.  usage: bash tools/ab_search.sh <out-prefix>
P=instacart_next_order_recommendation_amd
O=${1:-gpurun_out/ab_search}
cp $P/libicrec.so /tmp/icrec_B.so
A=${ICREC_AB_A:-tools/_libicrec_r03.so}
for v in A B A B; do
  if [ $v = A ]; then cp $A $P/libicrec.so; else cp /tmp/icrec_B.so $P/libicrec.so; fi
  python3 tools/search_roofline.py --rows 49688 --storage f32+filter --queries 256 1024 4096 2>/dev/null | sed "s/^{/{\"build\": \"$v\", /" >> ${O}_49k.jsonl
done
for v in A B; do
  if [ $v = A ]; then cp $A $P/libicrec.so; else cp /tmp/icrec_B.so $P/libicrec.so; fi
  python3 tools/search_roofline.py --rows 10000000 --storage bf16+filter --queries 1024 4096 2>/dev/null | sed "s/^{/{\"build\": \"$v\", /" >> ${O}_10m.jsonl
done
cp /tmp/icrec_B.so $P/libicrec.so
python3 - <<PY
import json
for f in ("${O}_49k.jsonl", "${O}_10m.jsonl"):
    for l in open(f):
        d = json.loads(l); print(d["build"], d["rows"], d["queries"], d["kernel_ms"], d["search_call_ms"], d["TFLOPs_algorithmic"])
PY
