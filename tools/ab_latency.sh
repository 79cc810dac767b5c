#!/bin/bash
# Same-box A/B of the single-request latency between two builds of libicrec.so: A = tools/_libicrec_r03.so (round 3's final
# library, built in the container), B = the in-tree library; alternated.  usage: bash tools/ab_latency.sh <rounds> [n_tokens]
P=instacart_next_order_recommendation_amd
R=${1:-3}; N=${2:-99}
cp $P/libicrec.so /tmp/icrec_B.so
A=${ICREC_AB_A:-tools/_libicrec_r03.so}
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then cp $A $P/libicrec.so; else cp /tmp/icrec_B.so $P/libicrec.so; fi
    echo -n "$v: "; python3 tools/latency_graph_trace.py run $N 2>/dev/null | tail -1
  done
done
cp /tmp/icrec_B.so $P/libicrec.so
