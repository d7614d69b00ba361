#!/bin/bash
# round 4, call 34: causal attention, longest query blocks first: parity (attention + model tests), A/B of the prefill
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q 2>&1 | tail -n 6 > gpurun_out/r4_c34_tests.log
cat gpurun_out/r4_c34_tests.log
for i in 1 2; do
timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
EMU_TUNE=32768 timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
done > gpurun_out/r4_c34_ab.log
timeout 300 python tools/prefill_time.py 1544 4 2>&1 | grep "prefill S" >> gpurun_out/r4_c34_ab.log
EMU_TUNE=32768 timeout 300 python tools/prefill_time.py 1544 4 2>&1 | grep "prefill S" >> gpurun_out/r4_c34_ab.log
cat gpurun_out/r4_c34_ab.log
