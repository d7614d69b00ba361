#!/bin/bash
# round 4, call 40: decode attention with the in-kernel last-arriver merge: bit-identity stress, model tests, decode A/B
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py tests/test_gpu_beam.py -x -q 2>&1 | tail -n 6 > gpurun_out/r4_c40_tests.log
cat gpurun_out/r4_c40_tests.log
for i in 1 2; do
timeout 600 python bench.py --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline --steps 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tail merge  ', d['value'], d['ms_per_step'])"
EMU_DECODE_TAIL=0 timeout 600 python bench.py --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline --steps 128 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two launches', d['value'], d['ms_per_step'])"
done > gpurun_out/r4_c40_ab.log
cat gpurun_out/r4_c40_ab.log
