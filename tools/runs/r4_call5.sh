#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_unet.py tests/test_gpu_unet_truewidth.py tests/test_gpu_emu1.py -x -q -m gpu > $O/r4_tests5.log 2>&1; echo "rc $?" >> $O/r4_tests5.log )
tail -n 4 $O/r4_tests5.log
timeout 900 python tools/unet_ab.py 20 7,7t32,7t64,7t128,7t16,7t8,7t248 3 > $O/r4_unet_ab1.log 2>&1
tail -n 2 $O/r4_unet_ab1.log
timeout 600 python bench.py > $O/r4_bench2.json 2> $O/r4_bench2.err
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r4_bench2.json'))
print('decode', d['value'], 'denoise', d['denoise']['ms_per_step'])
ex=d['extra']; print('vit', ex['vit_encode_ms'], 'prefill', ex['prefill_ms'], ex['prefill_mfma_frac'])
for k,v in d['legs'].items(): print(k, {kk:round(vv,3) for kk,vv in v.items() if isinstance(vv,(int,float))})
print('beam', d['beam_search_5']['ms_per_beam_step'])
PY
