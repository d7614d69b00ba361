#!/bin/bash
# round 5, call 23: tensor-parallel prefill as two concurrent lanes (second version: each half on its own stream): tests and the per-rank
# prefill cost of both schedules for TP = 8 / 4 shards
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py -q > gpurun_out/r5_c23_tests.log 2>&1
tail -n 30 gpurun_out/r5_c23_tests.log
timeout 600 python tools/tp_prefill_emulate.py 8 1544 4 rccl 2>&1 | grep "tp=\|summary\|Error\|error" > gpurun_out/r5_c23_tp8_prefill.log
cat gpurun_out/r5_c23_tp8_prefill.log
timeout 600 python tools/tp_prefill_emulate.py 4 1544 3 rccl 2>&1 | grep "summary\|Error\|error" > gpurun_out/r5_c23_tp4_prefill.log
cat gpurun_out/r5_c23_tp4_prefill.log
