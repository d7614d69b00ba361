#!/bin/bash
# round 5, call 25: two-lane tensor-parallel prefill against the fp32 oracle of the shard's layer 0 (yardstick for the distance between the
# schedules), and the per-rank cost of both schedules for a TP = 2 shard (fused RoPE / KV / V^T epilogue per lane)
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py -q -s > gpurun_out/r5_c25_tests.log 2>&1
grep -n "Fatal\|passed\|failed\|Error\|assert\|oracle" gpurun_out/r5_c25_tests.log | head -30
timeout 900 python tools/tp_prefill_emulate.py 2 1544 3 rccl 2>&1 | grep "summary\|Error\|error" > gpurun_out/r5_c25_tp2_prefill.log
cat gpurun_out/r5_c25_tp2_prefill.log
