#!/bin/bash
# round 4, call 33: ViT V^T epilogue: bit-identity, timing A/B; whole model test file
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_emu1.py tests/test_gpu_gemm_cfgs.py tests/test_gpu_fused_ln.py -x -q 2>&1 | tail -n 10 > gpurun_out/r4_c33_tests.log
cat gpurun_out/r4_c33_tests.log
for i in 1 2; do
timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode"
EMU_VIT_FUSION=0 timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode"
done > gpurun_out/r4_c33_ab.log
cat gpurun_out/r4_c33_ab.log
