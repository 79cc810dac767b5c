#!/bin/bash
# A/B of two builds of libicrec.so inside ONE gpurun call (boxes differ by several percent, so builds are only
# comparable on the same box): alternates bench.py between the in-tree library (B) and libicrec_prev.so (A).
# usage: tools/ab_bench.sh <rounds> <out-prefix> [bench args...]
set -e
P=instacart_next_order_recommendation_amd
R=${1:-2}; O=${2:-gpurun_out/ab}; shift 2 || true
cp $P/libicrec.so /tmp/icrec_B.so
cp ${ICREC_AB_A:-$P/libicrec_prev.so} /tmp/icrec_A.so
for i in $(seq 1 $R); do
  for v in A B; do
    cp /tmp/icrec_$v.so $P/libicrec.so
    python bench.py --no-cpu-baseline --no-latency --steps 30 "$@" > ${O}_${v}_$i.json 2>/dev/null
  done
done
cp /tmp/icrec_B.so $P/libicrec.so
