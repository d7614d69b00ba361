#!/bin/bash
# round 4, call 28: why is the W8A8 ViT encode 14.8 ms after the remainder-row change (12.3 before)?  kernel statistics
cd /tmp && export TMPDIR=/tmp
R=/root/repo; mkdir -p $R/gpurun_out
timeout 300 python $R/tools/vit_time.py 8 --fp8 2>&1 | grep "vit encode"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -- python $R/tools/vit_time.py 8 --fp8 > $R/gpurun_out/r4_c28_vit_time.log 2>&1
grep "vit encode" $R/gpurun_out/r4_c28_vit_time.log
python $R/tools/kernel_stats.py /tmp/prof_vit 16 > $R/gpurun_out/r4_c28_vit_fp8_kernel_stats.csv 2>&1
cut -c1-150 $R/gpurun_out/r4_c28_vit_fp8_kernel_stats.csv
