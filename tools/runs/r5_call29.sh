#!/bin/bash
# round 5, call 29: rocprofv3 kernel statistics of the VAE decode (128^2 latents -> 1024^2): how much of the 19 ms is the mid-block attention
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_vae -- python $R/tools/vae_time.py > $O/r5_c29_vae.log 2>&1
python $R/tools/kernel_stats.py /tmp/prof_vae 24 > $O/r5_c29_vae_kernel_stats.csv
cut -c1-210 $O/r5_c29_vae_kernel_stats.csv | head -22
grep "vae decode" $O/r5_c29_vae.log
