#!/bin/bash
# Same-box comparison of the single-request latency across several builds of libicrec.so (tools/_libicrec_<tag>.so) and the
# in-tree one (HEAD); alternated R times.  usage: bash tools/ab_latency_multi.sh <rounds> tag1 tag2 ...
P=instacart_next_order_recommendation_amd
R=$1; shift
cp $P/libicrec.so /tmp/icrec_HEAD.so
for i in $(seq 1 $R); do
  for v in HEAD "$@"; do
    if [ $v = HEAD ]; then cp /tmp/icrec_HEAD.so $P/libicrec.so; else cp tools/_libicrec_$v.so $P/libicrec.so; fi
    echo -n "$v: "; python3 tools/latency_graph_trace.py run 99 2>/dev/null | tail -1 | cut -c1-140
  done
done
cp /tmp/icrec_HEAD.so $P/libicrec.so
