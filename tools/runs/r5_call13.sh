#!/bin/bash
# round 5, call 13: the default bench line on the final sources
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r5_c13_bench.json 2> gpurun_out/r5_c13_bench.err
tail -n 3 gpurun_out/r5_c13_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_c13_bench.json').read().strip().splitlines()[-1])
e=d['extra']
print('value',d['value'],d['ms_per_step'],'roofline',d['roofline']['frac'],'token',d['roofline']['token_level_frac'],'traffic',d['roofline']['traffic'], d['roofline']['traffic_source'][:40])
print('vit',e['vit_encode_ms'],e['vit_roofline']['frac'],'prefill',e['prefill_ms'],e['prefill_roofline']['frac'],e['prefill_roofline']['traffic_source'][:60])
print('beam',d['beam_search_5']['ms_per_beam_step'])
x=d['denoise']; print('denoise',x['ms_per_step'],x['roofline']['frac'],'fp8',(x.get('fp8_transformer_blocks') or {}).get('ms_per_step'), x.get('cpu_baseline',{}).get('value'))
f=d['decode_fp8_weights']; print('fp8 decode',f['value'],'prefill',f['prefill_ms'],'vit8',(f.get('vit_encode_fp8') or {}).get('ms'))
l=d['legs']
for k,v in l.items(): print(k,{kk:vv for kk,vv in v.items() if kk in ('ms','prefill_ms','ms_per_step','mfma_frac','finite','vit_encode_4_images_ms')})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['kind'],d['cpu_baseline']['cores'], e.get('vit_cpu_baseline',{}).get('value'), e.get('prefill_cpu_baseline',{}).get('value'))
PY
