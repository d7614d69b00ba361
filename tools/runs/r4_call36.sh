#!/bin/bash
# round 4, call 36: ViT fc2 K-slice sum + LayerNorm + residual in one row-wise launch: bit-identity, A/B per fusion bit
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_emu1.py -x -q 2>&1 | tail -n 6 > gpurun_out/r4_c36_tests.log
cat gpurun_out/r4_c36_tests.log
for i in 1 2; do
for f in 3 1 0; do EMU_VIT_FUSION=$f timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode"; done
done > gpurun_out/r4_c36_ab.log
cat gpurun_out/r4_c36_ab.log
