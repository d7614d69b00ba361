#!/bin/bash
# round 5, call 7: successor-weight prefetch from inside the GEMM kernels: UNet tests, same-run A/B (tune bit 16 = off)
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_truewidth.py tests/test_gpu_fused_ln.py -x -q > gpurun_out/r5_c7_tests.log 2>&1
tail -n 5 gpurun_out/r5_c7_tests.log
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k finished_hypotheses > gpurun_out/r5_c7_tests_b.log 2>&1
tail -n 3 gpurun_out/r5_c7_tests_b.log
timeout 900 python tools/unet_ab.py 20 7,7t65536 3 > gpurun_out/r5_c7_unet_ab.log 2>&1
tail -n 8 gpurun_out/r5_c7_unet_ab.log
