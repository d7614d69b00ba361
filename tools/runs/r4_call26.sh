#!/bin/bash
# round 4, call 26: V^T / cross-attention epilogues with fp8 operands: tests, same-run A/B of the W8A8 step with / without them; 8192^3
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_unet_truewidth.py tests/test_gpu_fp8.py -x -q 2>&1 | tail -n 8 > gpurun_out/r4_c26_tests.log
cat gpurun_out/r4_c26_tests.log
timeout 600 python tools/unet_ab.py 20 7,fp8,fp8m0 2 > gpurun_out/r4_c26_unet_ab_fp8.log 2>&1
tail -n 8 gpurun_out/r4_c26_unet_ab_fp8.log
timeout 300 python tools/fp8_gemm_time.py --filter square --cfgs 0,P > gpurun_out/r4_c26_square.log 2>&1
tail -n 2 gpurun_out/r4_c26_square.log
