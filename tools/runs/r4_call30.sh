#!/bin/bash
# round 4, call 30: default bench line + FETCH_SIZE of the prefill GEMMs + denoise kernel statistics on the final sources
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python bench.py > gpurun_out/r4_c30_bench.json 2> gpurun_out/r4_c30_bench.err
tail -n 2 gpurun_out/r4_c30_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c30_bench.json').read().strip().splitlines()[-1])
e=d['extra']
print('value',d['value'],'roofline',d['roofline']['frac'],'traffic',d['roofline']['traffic'])
print('vit',e['vit_encode_ms'],e['vit_roofline']['frac'],'prefill',e['prefill_ms'],e['prefill_roofline']['frac'],e['prefill_roofline']['traffic_source'][:60])
print('beam',d['beam_search_5']['ms_per_beam_step'])
x=d['denoise']; print('denoise',x['ms_per_step'],x['roofline']['frac'],'fp8',(x.get('fp8_transformer_blocks') or {}).get('ms_per_step'))
f=d['decode_fp8_weights']; print('fp8 decode',f['value'],'prefill',f['prefill_ms'],'vit8',(f.get('vit_encode_fp8') or {}).get('ms'))
l=d['legs']
for k,v in l.items(): print(k,{kk:vv for kk,vv in v.items() if kk in ('ms','prefill_ms','ms_per_step','mfma_frac','finite','vit_encode_4_images_ms')})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['kind'])
PY
cd /tmp && export TMPDIR=/tmp
R=/root/repo
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pf -- python $R/bench.py --pmc-prefill 2 > $R/gpurun_out/r4_c30_pmc_prefill_bench.json 2> $R/gpurun_out/r4_c30_pmc_prefill.err
cd $R && python tools/pmc_gemm_traffic.py /tmp/prof_pf gpurun_out/r4_c30_pmc_prefill_bench.json > gpurun_out/r04_prefill_gemm_pmc_traffic.json
tail -n 4 gpurun_out/r04_prefill_gemm_pmc_traffic.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dn -- python $R/bench.py --only-denoise --denoise-steps 12 --no-fp8 > $R/gpurun_out/r4_c30_denoise_bench.json 2> $R/gpurun_out/r4_c30_denoise.err
python $R/tools/kernel_stats.py /tmp/prof_dn 60 > $R/gpurun_out/r4_c30_denoise_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $R/bench.py --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline > $R/gpurun_out/r4_c30_decode_bench.json 2> $R/gpurun_out/r4_c30_decode.err
python $R/tools/kernel_stats.py /tmp/prof_dec 60 > $R/gpurun_out/r4_c30_decode_kernel_stats.csv
head -n 14 $R/gpurun_out/r4_c30_decode_kernel_stats.csv | cut -c1-130
