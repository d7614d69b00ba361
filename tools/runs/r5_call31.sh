#!/bin/bash
# round 5, call 31: bench.py --gpus 2 with both ranks on this GPU and the opt-in prefill-schedule leg
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_tp_multiproc.py -q -k bench_two_ranks > gpurun_out/r5_c31_tests.log 2>&1
tail -n 12 gpurun_out/r5_c31_tests.log | cut -c1-400
