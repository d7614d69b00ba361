#!/bin/bash
# round 4, call 20: isolate the NaN of the W8A8 UNet step under graph replay; kernel statistics of the W8A8 ViT / UNet legs
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/debug/unet_fp8_graph.py > gpurun_out/r4_c20_debug.log 2>&1
cat gpurun_out/r4_c20_debug.log | tail -n 14
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -- python /root/repo/tools/vit_time.py --fp8 > /root/repo/gpurun_out/r4_c20_vit_time.log 2>&1
tail -n 5 /root/repo/gpurun_out/r4_c20_vit_time.log
python /root/repo/tools/kernel_stats.py /tmp/prof_vit > /root/repo/gpurun_out/r4_c20_vit_fp8_kernel_stats.csv 2>&1
head -n 30 /root/repo/gpurun_out/r4_c20_vit_fp8_kernel_stats.csv
