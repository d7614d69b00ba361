#!/bin/bash
# round 5, call 41: the two-lane prefill with the unconditional join: its tests, the TP multi-process suite
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py tests/test_gpu_tp_multiproc.py -q > gpurun_out/r5_c41_tests.log 2>&1
grep -n "passed\|failed\|Fatal" gpurun_out/r5_c41_tests.log | tail -3
