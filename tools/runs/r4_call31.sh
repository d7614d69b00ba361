#!/bin/bash
# round 4, call 31: the default bench line once more (box lottery: the pool's boxes differ by +-5 %)
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r4_c31_bench.json 2> gpurun_out/r4_c31_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c31_bench.json').read().strip().splitlines()[-1])
e=d['extra']
print('value',d['value'],'vit',e['vit_encode_ms'],'prefill',e['prefill_ms'],e['prefill_roofline']['frac'],e['prefill_roofline']['traffic_source'][:70])
x=d['denoise']; print('denoise',x['ms_per_step'],x['roofline']['frac'],'fp8',(x.get('fp8_transformer_blocks') or {}).get('ms_per_step'),'beam',d['beam_search_5']['ms_per_beam_step'])
l=d['legs']; print('S1544',l['prefill_fewshot_S1544']['prefill_ms'],l['prefill_fewshot_S1544']['mfma_frac'],'e2e',l['any_to_image_e2e']['ms'],l['any_to_image_e2e_fp8']['ms'])
PY
