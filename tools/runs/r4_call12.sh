#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu -k "generate_image or beam" > $O/r4_tests12.log 2>&1; echo "rc $?" >> $O/r4_tests12.log )
tail -n 6 $O/r4_tests12.log
timeout 900 python bench.py > $O/r4_bench3.json 2> $O/r4_bench3.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r4_bench3.json'))
print('decode', d['value'], 'denoise', d['denoise']['ms_per_step'])
print(json.dumps(d['denoise']['kernels'], indent=1)[:3500])
for k,v in d['legs'].items(): print(k, {kk:round(vv,3) for kk,vv in v.items() if isinstance(vv,(int,float))})
b=d['beam_search_5']; print({k:v for k,v in b.items() if k!='note'})
PY
tail -n 3 $O/r4_bench3.err
