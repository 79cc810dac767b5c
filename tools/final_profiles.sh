#!/bin/bash
# One gpurun call that refreshes the round's records under gpurun_out/final/ (copied into profiles/ afterwards):
# full GPU test log, bench line, kernel trace, per-kernel PMC (clock / MFMA-busy / stalls), HBM traffic (two passes),
# L2 -> L1 request stream of the weight-streaming kernels.
# usage: bash tools/final_profiles.sh   (on the GPU box, from the repo root)
set -o pipefail
O=gpurun_out/final; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="python3 bench.py --no-cpu-baseline --no-latency"
if [ "$1" != "pmc" ]; then   # `pmc`: only the counter passes
python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo "exit=$?" >> $O/gpu_tests.log; tail -2 $O/gpu_tests.log
python bench.py > $O/bench.json 2> $O/bench.err && echo bench-ok
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/trace.err && echo trace-ok
python tools/prof_summary.py $O/trace > $O/kernel_summary.txt
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
fi
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS -d $O/pmc_a -- $B --steps 2 --warmup 1 > /dev/null 2> $O/pmc_a.err && echo pmc-a-ok
python tools/pmc_summary.py $O/pmc_a > $O/pmc_per_kernel.txt
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/pmc_f -- $B --steps 3 --warmup 1 > /dev/null 2> $O/pmc_f.err && echo pmc-f-ok
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/pmc_w -- $B --steps 3 --warmup 1 > /dev/null 2> $O/pmc_w.err && echo pmc-w-ok
python tools/pmc_bytes.py $O/pmc_f $O/pmc_w > $O/pmc_hbm_bytes.txt
python tools/pmc_traffic_json.py $O/pmc_f $O/pmc_w $O/roofline_traffic.json ffn_fused2_kernel=ffn_fused2_kernel,1048576 qkv_resident_kernel=qkv_resident_kernel attention_6tile=attention_x3_kernelILi6 attention_8tile=attention_x3_kernelILi8 attention_4tile=attention_x3_kernelILi4 search_resident="TileCfg<8, 1, 1, 2>" > /dev/null
rocprofv3 --output-format csv --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum -d $O/pmc_l2 -- $B --steps 2 --warmup 1 > /dev/null 2> $O/pmc_l2.err && echo pmc-l2-ok
python tools/pmc_raw.py $O/pmc_l2 > $O/pmc_l2_stream.txt
rm -rf $O/trace $O/pmc_a $O/pmc_f $O/pmc_w $O/pmc_l2
ls -la $O
