#!/bin/bash
# round 4, call 29: same-box A/B of the previous build (32x32 remainder rows in phase B) and the new one (16x16 in phase A): ViT bf16 / W8A8, prefill
cd /root/repo
mkdir -p gpurun_out
P=/root/repo/emu_amd/csrc/libemu_hip_prev.so
for i in 1 2; do
echo "--- new"; timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode"; timeout 300 python tools/vit_time.py 8 --fp8 2>&1 | grep "vit encode"
echo "--- prev"; EMU_HIP_TOOLS=1 EMU_HIP_LIB=$P timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode"; EMU_HIP_TOOLS=1 EMU_HIP_LIB=$P timeout 300 python tools/vit_time.py 8 --fp8 2>&1 | grep "vit encode"
done > gpurun_out/r4_c29_ab.log 2>&1
echo "--- new"; timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S" >> gpurun_out/r4_c29_ab.log
echo "--- prev"; EMU_HIP_TOOLS=1 EMU_HIP_LIB=$P timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S" >> gpurun_out/r4_c29_ab.log
cat gpurun_out/r4_c29_ab.log
