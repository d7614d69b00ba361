#!/bin/bash
# round 5, call 36: 2-D tile blocks per XCD in the 256x256 kernel (unsliced plain GEMMs): same-run A/B (tune 16 = the column-major strips),
# GEMM configuration tests
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/gemm_ab.py --cfgs 0,0t16,0,0t16 --no-check --iters 20 --shapes "8192,8192,8192,0;4096,4096,4096,0;8192,5120,640,5;8192,1920,640,0;8192,640,2560,1;4100,15360,1792,4;4100,6144,1792,0;2048,10240,1280,5;32768,320,960,0" > gpurun_out/r5_c36_gemm_ab.log 2>&1
cat gpurun_out/r5_c36_gemm_ab.log | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_gemm_cfgs.py -x -q > gpurun_out/r5_c36_tests.log 2>&1
tail -n 3 gpurun_out/r5_c36_tests.log
