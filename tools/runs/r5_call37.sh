#!/bin/bash
# round 5, call 37: 2-D tile blocks per XCD in the 256x256 kernel, end to end: same-run A/B of the denoise step, the 4-image ViT encode and
# the S = 1544 prefill (tune bit 17 = column-major strips as before, bit 18 = blocks only for launches of more than one round)
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/unet_ab.py 20 7,7t131072,7t262144 3 > gpurun_out/r5_c37_unet_ab.log 2>&1
tail -n 6 gpurun_out/r5_c37_unet_ab.log
EMU_VIT_BATCH=4 EMU_TUNES=0,131072,262144,0,131072,262144 timeout 600 python tools/vit_time.py 6 --graph > gpurun_out/r5_c37_vit.log 2>&1
grep "vit encode" gpurun_out/r5_c37_vit.log
EMU_TUNES=0,131072,0,131072 timeout 600 python tools/vit_time.py 8 --graph >> gpurun_out/r5_c37_vit.log 2>&1
grep "vit encode" gpurun_out/r5_c37_vit.log | tail -4
EMU_TUNES=0,131072,0,131072 timeout 900 python tools/prefill_time.py 1544 5 --graph > gpurun_out/r5_c37_prefill.log 2>&1
grep "prefill S" gpurun_out/r5_c37_prefill.log
