#!/bin/bash
# round 5, call 18: the whole GPU suite, smoke and the default bench line on the final sources
cd /root/repo
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -x -q > gpurun_out/r5_c18_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r5_c18_tests.log | tail -n 2
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r5_c18_smoke.log 2>&1
tail -n 1 gpurun_out/r5_c18_smoke.log
timeout 1500 python bench.py > gpurun_out/r5_c18_bench.json 2> gpurun_out/r5_c18_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_c18_bench.json').read().strip().splitlines()[-1])
e=d['extra']
print('value',d['value'],d['ms_per_step'],'roofline',d['roofline']['frac'],'token',d['roofline']['token_level_frac'])
print('vit',e['vit_encode_ms'],'prefill',e['prefill_ms'],e['prefill_roofline']['frac'])
x=d['denoise']; print('denoise',x['ms_per_step'],x['roofline']['frac'])
print(d['legs']['prefill_fewshot_S1544'])
PY
