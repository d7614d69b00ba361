#!/bin/bash
# round 5, call 14: which launch of the split GEGLU GEMM names the successor (head vs remainder): same-run A/B
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/unet_ab.py 20 7,7t131072,7t65536 3 > gpurun_out/r5_c14_unet_ab.log 2>&1
tail -n 10 gpurun_out/r5_c14_unet_ab.log
