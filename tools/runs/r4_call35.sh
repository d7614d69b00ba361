#!/bin/bash
# round 4, call 35: K-slice sum + RMSNorm in one row-wise launch: bit-identity (the prefill fusion test), model tests, A/B
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -n 6 > gpurun_out/r4_c35_tests.log
cat gpurun_out/r4_c35_tests.log
for i in 1 2; do
timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
EMU_PREFILL_FUSION=0 timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
done > gpurun_out/r4_c35_ab.log
cat gpurun_out/r4_c35_ab.log
