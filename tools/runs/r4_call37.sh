#!/bin/bash
# round 4, call 37: kernel statistics of the bf16 ViT encode on the final sources
cd /tmp && export TMPDIR=/tmp
R=/root/repo; mkdir -p $R/gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -- python $R/tools/vit_time.py 8 > $R/gpurun_out/r4_c37_vit_time.log 2>&1
grep "vit encode" $R/gpurun_out/r4_c37_vit_time.log
python $R/tools/kernel_stats.py /tmp/prof_vit 24 > $R/gpurun_out/r4_c37_vit_kernel_stats.csv 2>&1
cut -c1-150 $R/gpurun_out/r4_c37_vit_kernel_stats.csv
