#!/bin/bash
# round 5, call 11: eight ranks sharing the GPU at the true LLaMA-33B width (7-head shards), launches and fused layers
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_tp_multiproc.py -x -q -k "tp8_true_width or sharing_one_gpu" > gpurun_out/r5_c11_tests.log 2>&1
tail -n 30 gpurun_out/r5_c11_tests.log
