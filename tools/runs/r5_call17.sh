#!/bin/bash
# round 5, call 17: the batched (4-image, M = 4100) ViT encode: tile configurations of its four GEMMs on cold weights, kernel statistics
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/gemm_ab.py --cfgs 0,B,C,K,S,P,Q,H --no-check --cold-mb 700 --shapes "4100,6144,1792,0;4100,1792,2048,1;4100,15360,1792,4;4100,1792,15360,1;4100,1792,15360,0" > gpurun_out/r5_c17_gemm_ab_vit_b4.log 2>&1
tail -n 8 gpurun_out/r5_c17_gemm_ab_vit_b4.log
cd /tmp && export TMPDIR=/tmp
R=/root/repo
EMU_VIT_BATCH=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vit -- python $R/tools/vit_time.py 4 > $R/gpurun_out/r5_c17_vit_b4.log 2>&1
python $R/tools/kernel_stats.py /tmp/prof_vit 20 > $R/gpurun_out/r5_c17_vit_b4_kernel_stats.csv
head -n 14 $R/gpurun_out/r5_c17_vit_b4_kernel_stats.csv | cut -c1-150
