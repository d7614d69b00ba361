#!/bin/bash
# round 5, call 1: fused decode layers -- bit-identity tests, TP = 8 shard emulation (modes 0/1/2), TP = 1 A/B
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode_fused.py -x -q > gpurun_out/r5_c1_tests.log 2>&1
tail -n 15 gpurun_out/r5_c1_tests.log
timeout 600 python tools/tp_emulate.py 8 32 p2p 0,1,2 > gpurun_out/r5_c1_tp8.log 2>&1
tail -n 8 gpurun_out/r5_c1_tp8.log
timeout 900 python tools/tp_emulate.py 1 24 none 0,1 > gpurun_out/r5_c1_tp1.log 2>&1
tail -n 6 gpurun_out/r5_c1_tp1.log
