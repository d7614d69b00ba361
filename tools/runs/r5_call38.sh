#!/bin/bash
# round 5, call 38: final sources -- the whole GPU suite, smoke(), the driver-style bench line, two-lane prefill parity at 60 layers (TP = 8 shard)
cd /root/repo
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_c38_tests.log 2>&1
tail -n 3 gpurun_out/r5_c38_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_tp1_final_v3.json 2> gpurun_out/r05_bench_tp1_final_v3.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_tp1_final_v3.json') if l.startswith('{')][-1])
print("decode", round(d['value'],2), "tok/s", round(d['ms_per_step'],3), "ms; roofline", round(d['roofline']['frac'],3), "; denoise", round(d['denoise']['ms_per_step'],2), "ms;",
      {k: {kk: round(vv,2) for kk,vv in v.items() if isinstance(vv,(int,float))} for k,v in d['legs'].items() if isinstance(v,dict)})
print({k: v for k, v in d['extra'].items() if isinstance(v,(int,float))} if isinstance(d.get('extra'),dict) else '')
PY
timeout 600 python tools/tp_prefill_emulate.py 8 1544 3 rccl 2>&1 | grep "parity\|summary" > gpurun_out/r5_c38_tp8_prefill_parity.log
cat gpurun_out/r5_c38_tp8_prefill_parity.log
