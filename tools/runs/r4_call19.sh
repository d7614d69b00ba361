#!/bin/bash
# round 4, call 19: W8A8 ViT / UNet blocks: parity tests, then the bench legs that report them
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_unet_truewidth.py -x -q 2>&1 | tail -n 15 > gpurun_out/r4_c19_tests.log
cat gpurun_out/r4_c19_tests.log
timeout 900 python bench.py --only-denoise --denoise-steps 20 > gpurun_out/r4_c19_denoise.json 2> gpurun_out/r4_c19_denoise.err
tail -n 3 gpurun_out/r4_c19_denoise.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c19_denoise.json').read().strip().splitlines()[-1])
x=d.get('denoise',d)
print('bf16 ms/step',x.get('ms_per_step'),'fp8',x.get('fp8_transformer_blocks'))
PY
timeout 1200 python bench.py --no-cpu-baseline --no-beam --no-denoise --no-legs > gpurun_out/r4_c19_bench_fp8.json 2> gpurun_out/r4_c19_bench.err
tail -n 3 gpurun_out/r4_c19_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c19_bench_fp8.json').read().strip().splitlines()[-1])
print('value',d['value'],'vit',d['extra']['vit_encode_ms'],'prefill',d['extra']['prefill_ms'])
f=d.get('decode_fp8_weights',{})
print({k:f.get(k) for k in ('value','prefill_ms','prefill_rel_l2_vs_bf16_hidden','vit_encode_fp8','note')})
PY
