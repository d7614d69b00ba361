#!/bin/bash
# round 5, call 26: two-lane tensor-parallel prefill tests on the final bounds (fp32 oracle yardstick), TP multi-process suite
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py -q -s > gpurun_out/r5_c26_tests.log 2>&1
grep -n "Fatal\|passed\|failed\|oracle:" gpurun_out/r5_c26_tests.log | head -30
timeout 1500 python -m pytest tests/test_gpu_tp_multiproc.py -q > gpurun_out/r5_c26_tests_tp.log 2>&1
tail -n 3 gpurun_out/r5_c26_tests_tp.log
