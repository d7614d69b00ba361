#!/bin/bash
# round 4, call 27: remainder rows on 16x16x32 MFMAs: GEMM parity at the true shapes, then prefill / ViT timing
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm_cfgs.py tests/test_gpu_fp8.py tests/test_gpu_ops.py tests/test_gpu_fused_ln.py -x -q 2>&1 | tail -n 8 > gpurun_out/r4_c27_tests.log
cat gpurun_out/r4_c27_tests.log
timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S" > gpurun_out/r4_c27_time.log
timeout 300 python tools/prefill_time.py 1544 4 2>&1 | grep "prefill S" >> gpurun_out/r4_c27_time.log
timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode" >> gpurun_out/r4_c27_time.log
timeout 300 python tools/vit_time.py 8 --fp8 2>&1 | grep "vit encode" >> gpurun_out/r4_c27_time.log
cat gpurun_out/r4_c27_time.log
timeout 300 python tools/fp8_gemm_time.py --filter prefill --cfgs 0 > gpurun_out/r4_c27_prefill_gemms.log 2>&1
cat gpurun_out/r4_c27_prefill_gemms.log | tail -n 6
timeout 300 python tools/fp8_gemm_time.py --filter "vit f" --cfgs 0 2>&1 | tail -n 2
