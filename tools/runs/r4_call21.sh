#!/bin/bash
# round 4, call 21: W8A8 blocks after the single-pass quantiser: tests, kernel statistics of the fp8 denoise leg, ViT timing
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_unet_truewidth.py -x -q 2>&1 | tail -n 8 > gpurun_out/r4_c21_tests.log
cat gpurun_out/r4_c21_tests.log
timeout 300 python tools/vit_time.py 8 > gpurun_out/r4_c21_vit.log 2>&1; timeout 300 python tools/vit_time.py 8 --fp8 >> gpurun_out/r4_c21_vit.log 2>&1
grep "vit encode" gpurun_out/r4_c21_vit.log
timeout 900 python bench.py --only-denoise --denoise-steps 20 > gpurun_out/r4_c21_denoise.json 2> gpurun_out/r4_c21_denoise.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c21_denoise.json').read().strip().splitlines()[-1])
x=d.get('denoise',d)
f=x.get('fp8_transformer_blocks') or {}
print('bf16 ms/step',x.get('ms_per_step'),'fp8 ms/step',f.get('ms_per_step'),'rel',f.get('rel_l2_of_final_latents_vs_bf16_run'),f.get('note') if f.get('ms_per_step') is None else '')
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d8 -- python /root/repo/tools/unet_ab.py 6 fp8 1 > /root/repo/gpurun_out/r4_c21_unet_ab_fp8.log 2>&1
tail -n 4 /root/repo/gpurun_out/r4_c21_unet_ab_fp8.log
python /root/repo/tools/kernel_stats.py /tmp/prof_d8 30 > /root/repo/gpurun_out/r4_c21_denoise_fp8_kernel_stats.csv 2>&1
head -n 24 /root/repo/gpurun_out/r4_c21_denoise_fp8_kernel_stats.csv | cut -c1-150
