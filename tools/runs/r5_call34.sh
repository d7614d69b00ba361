#!/bin/bash
# round 5, call 34: this repo's bf16 GEMM kernels beside the vendor library (torch linear -> hipBLASLt) on the hot path's own shapes
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/gemm_vs_library.py > gpurun_out/r5_c34_gemm_vs_library.log 2>&1
cat gpurun_out/r5_c34_gemm_vs_library.log | cut -c1-200
