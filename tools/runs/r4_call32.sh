#!/bin/bash
# round 4, call 32: RoPE + KV append + V^T in the qkv epilogue: bit-identity test, model tests, timing A/B in one process order
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "rope_epilogue or true_width_single or generate_greedy or tiny" 2>&1 | tail -n 12 > gpurun_out/r4_c32_tests.log
cat gpurun_out/r4_c32_tests.log
for i in 1 2; do
timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
EMU_PREFILL_FUSION=0 timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
done > gpurun_out/r4_c32_ab.log
timeout 300 python tools/prefill_time.py 1544 4 2>&1 | grep "prefill S" >> gpurun_out/r4_c32_ab.log
EMU_PREFILL_FUSION=0 timeout 300 python tools/prefill_time.py 1544 4 2>&1 | grep "prefill S" >> gpurun_out/r4_c32_ab.log
cat gpurun_out/r4_c32_ab.log
