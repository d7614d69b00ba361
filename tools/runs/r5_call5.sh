#!/bin/bash
# round 5, call 5: the default bench line with the new CPU baselines of the legs and the fp8 roofline objects
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r5_c5_bench.json 2> gpurun_out/r5_c5_bench.err
tail -n 4 gpurun_out/r5_c5_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_c5_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
print('denoise', d['denoise']['ms_per_step'], d['denoise'].get('cpu_baseline'))
print('extra', d['extra']['vit_encode_ms'], d['extra']['prefill_ms'], d['extra'].get('vit_cpu_baseline'), d['extra'].get('prefill_cpu_baseline'), d['extra'].get('legs_cpu_baseline_note'))
print('fp8', d['decode_fp8_weights'].get('roofline'), d['decode_fp8_weights'].get('prefill_roofline'))
print('unet fp8', d['denoise']['fp8_transformer_blocks'].get('roofline'))
PY
