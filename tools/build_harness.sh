#!/bin/bash
# Builds libicrec.so and the two kernel harness binaries (timing: tools/_ffn_bench, phase stamps: tools/_ffn_stamps).
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
make -C "$R/instacart_next_order_recommendation_amd/csrc" -j8 2>&1 | grep -E "error|Error" -A3 || true
F="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off"
hipcc $F "$R/tools/ffn_bench.hip" "$R/instacart_next_order_recommendation_amd/csrc/api.hip" -o "$R/tools/_ffn_bench" 2>&1 | grep -E "error" -A3 || true
hipcc $F -DICREC_STAMPS "$R/tools/ffn_bench.hip" "$R/instacart_next_order_recommendation_amd/csrc/api.hip" -o "$R/tools/_ffn_stamps" 2>&1 | grep -E "error" -A3 || true
ls -la "$R/tools/_ffn_bench" "$R/tools/_ffn_stamps" "$R/instacart_next_order_recommendation_amd/libicrec.so"
