#!/bin/bash
# round 4, call 18: fp8 form of the lock-step tiles + remainder rows with fp8 operands + W8A8 ViT blocks: parity, then timing
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | tail -n 15 > gpurun_out/r4_c18_fp8_tests.log
timeout 600 python -m pytest tests/test_gpu_gemm_cfgs.py tests/test_gpu_fused_ln.py -x -q 2>&1 | tail -n 6 > gpurun_out/r4_c18_gemm_tests.log
timeout 300 python tools/fp8_gemm_time.py --filter unet --cfgs 0,K,B,C,S,P --cold-mb 800 > gpurun_out/r4_c18_fp8_unet_cold.log 2>&1
timeout 300 python tools/fp8_gemm_time.py --filter vit --cfgs 0,K,B,C,S,P > gpurun_out/r4_c18_fp8_vit.log 2>&1
timeout 300 python tools/fp8_gemm_time.py --filter prefill --cfgs 0 > gpurun_out/r4_c18_fp8_prefill.log 2>&1
cat gpurun_out/r4_c18_fp8_tests.log gpurun_out/r4_c18_gemm_tests.log
cat gpurun_out/r4_c18_fp8_unet_cold.log gpurun_out/r4_c18_fp8_vit.log gpurun_out/r4_c18_fp8_prefill.log
