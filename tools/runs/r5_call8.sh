#!/bin/bash
# round 5, call 8: successor prefetch across the UNet's modules and through the ViT's blocks: tests + same-run A/B
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_truewidth.py -x -q > gpurun_out/r5_c8_tests.log 2>&1
tail -n 3 gpurun_out/r5_c8_tests.log
timeout 900 python tools/unet_ab.py 20 7,7t65536 2 > gpurun_out/r5_c8_unet_ab.log 2>&1
tail -n 5 gpurun_out/r5_c8_unet_ab.log
EMU_TUNES=0,65536,0,65536 timeout 600 python tools/vit_time.py 8 --graph > gpurun_out/r5_c8_vit.log 2>&1
EMU_VIT_BATCH=4 EMU_TUNES=0,65536,0,65536 timeout 600 python tools/vit_time.py 6 --graph >> gpurun_out/r5_c8_vit.log 2>&1
grep "vit encode" gpurun_out/r5_c8_vit.log
