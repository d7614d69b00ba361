#!/bin/bash
# round 5, call 32: two-lane prefill with the lanes issued stage by stage (o_A, o_B, down_A, down_B): tests, per-rank cost at TP = 8
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py -q -s > gpurun_out/r5_c32_tests.log 2>&1
grep -n "Fatal\|passed\|failed\|oracle:" gpurun_out/r5_c32_tests.log | head
timeout 600 python tools/tp_prefill_emulate.py 8 1544 4 rccl 2>&1 | grep "summary\|Error\|error" > gpurun_out/r5_c32_tp8_prefill.log
cat gpurun_out/r5_c32_tp8_prefill.log
