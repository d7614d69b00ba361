#!/bin/bash
# round 4, call 24: the default bench line on the final sources
cd /root/repo
mkdir -p gpurun_out
( time timeout 1500 python bench.py > gpurun_out/r4_c24_bench.json 2> gpurun_out/r4_c24_bench.err ) 2> gpurun_out/r4_c24_time.log
tail -n 4 gpurun_out/r4_c24_bench.err; cat gpurun_out/r4_c24_time.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c24_bench.json').read().strip().splitlines()[-1])
e=d['extra']
print('value',d['value'],'roofline',d['roofline']['frac'],'traffic',d['roofline']['traffic'])
print('vit',e['vit_encode_ms'],'prefill',e['prefill_ms'],e['prefill_roofline']['frac'],e['prefill_roofline']['traffic_source'][:60])
print('beam',d['beam_search_5']['ms_per_beam_step'])
x=d['denoise']; print('denoise',x['ms_per_step'],x['roofline']['frac'],'fp8',(x.get('fp8_transformer_blocks') or {}).get('ms_per_step'))
f=d['decode_fp8_weights']; print('fp8 decode',f['value'],'prefill',f['prefill_ms'],'vit8',f.get('vit_encode_fp8'))
l=d['legs']
for k,v in l.items(): print(k,{kk:vv for kk,vv in v.items() if kk in ('ms','prefill_ms','ms_per_step','mfma_frac','finite')})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['kind'])
PY
