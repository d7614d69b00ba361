#!/bin/bash
# round 5, call 22: tensor-parallel prefill in two row halves (all-reduces on a second stream): tests (1-rank comm block at the true
# width, TP = 8 / 2 shards; two rank processes sharing the GPU) and the per-rank prefill cost of both schedules for a TP = 8 shard
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py -x -q > gpurun_out/r5_c22_tests.log 2>&1
tail -n 25 gpurun_out/r5_c22_tests.log
timeout 600 python tools/tp_prefill_emulate.py 8 1544 4 rccl 2>&1 | grep "tp=\|summary\|Error\|error" > gpurun_out/r5_c22_tp8_prefill.log
cat gpurun_out/r5_c22_tp8_prefill.log
