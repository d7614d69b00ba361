#!/bin/bash
# round 5, call 10: final form of the successor prefetch (UNet transformer chain, linear, two lines per thread): A/B + the GPU suite
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/unet_ab.py 20 7,7t65536 2 > gpurun_out/r5_c10_unet_ab.log 2>&1
tail -n 5 gpurun_out/r5_c10_unet_ab.log
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_c10_tests.log 2>&1
tail -n 5 gpurun_out/r5_c10_tests.log
