#!/bin/bash
# round 5, call 16: mode 3 (tail all-reduce): shard tests, 8-rank true-width run, per-rank cost of a TP = 8 shard
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_decode_fused.py -x -q -k "tp_shard" > gpurun_out/r5_c16_tests.log 2>&1
tail -n 4 gpurun_out/r5_c16_tests.log
timeout 1200 python -m pytest tests/test_gpu_tp_multiproc.py -x -q -k "tp8_true_width" >> gpurun_out/r5_c16_tests.log 2>&1
tail -n 4 gpurun_out/r5_c16_tests.log
timeout 600 python tools/tp_emulate.py 8 32 p2p 0,3,2 2>&1 | grep hipGraph > gpurun_out/r5_c16_tp8.log
cat gpurun_out/r5_c16_tp8.log
