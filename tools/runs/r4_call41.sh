#!/bin/bash
# round 4, call 41: TP = 2 bench with both ranks on the one GPU (validation of the data path + the new per-rank breakdown)
cd /root/repo
mkdir -p gpurun_out
EMU_TP_SHARED_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --no-legs --no-denoise --no-beam --no-fp8 --no-cpu-baseline --steps 16 --warmup 4 > gpurun_out/r4_c41_tp2.json 2> gpurun_out/r4_c41_tp2.err
tail -n 3 gpurun_out/r4_c41_tp2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_c41_tp2.json').read().strip().splitlines()[-1])
print(d['value'], d['n_gpus'], d['config']['parallelism'], d['config']['allreduce'], d['config']['tp'], d['extra']['first_tokens'])
PY
