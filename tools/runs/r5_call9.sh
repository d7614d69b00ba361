#!/bin/bash
# round 5, call 9: k-major, two-lines-per-thread successor prefetch: UNet / ViT / prefill same-run A/B (tune 65536 = off), tests
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/unet_ab.py 20 7,7t65536 2 > gpurun_out/r5_c9_unet_ab.log 2>&1
tail -n 5 gpurun_out/r5_c9_unet_ab.log
EMU_TUNES=0,65536,0,65536 timeout 600 python tools/vit_time.py 8 --graph > gpurun_out/r5_c9_vit.log 2>&1
EMU_VIT_BATCH=4 EMU_TUNES=0,65536,0,65536 timeout 600 python tools/vit_time.py 6 --graph >> gpurun_out/r5_c9_vit.log 2>&1
grep "vit encode" gpurun_out/r5_c9_vit.log
EMU_TUNES=0,65536,0,65536 timeout 900 python tools/prefill_time.py 770 6 --graph > gpurun_out/r5_c9_prefill.log 2>&1
grep "prefill S" gpurun_out/r5_c9_prefill.log
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_model.py tests/test_gpu_gemm_cfgs.py -x -q > gpurun_out/r5_c9_tests.log 2>&1
tail -n 3 gpurun_out/r5_c9_tests.log
