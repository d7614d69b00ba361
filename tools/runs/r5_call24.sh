#!/bin/bash
# round 5, call 24 (run twice: one event recorded on both lanes, then one event per lane): two-lane tensor-parallel prefill, peer-to-peer all-reduces chained by events instead of a third stream: tests
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_tp_overlap.py -q > gpurun_out/r5_c24_tests.log 2>&1
grep -n "Fatal\|passed\|failed\|Error\|assert" gpurun_out/r5_c24_tests.log | head -20
tail -n 12 gpurun_out/r5_c24_tests.log | cut -c1-300
