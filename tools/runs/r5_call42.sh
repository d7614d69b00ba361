#!/bin/bash
# round 5, call 42: o_proj with the decode attention's split merge in its prologue (TP shards): bit-identity tests, per-rank cost of a TP = 8 shard
# with (default) and without (tune bit 19) it
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_decode_fused.py -x -q > gpurun_out/r5_c42_tests.log 2>&1
tail -n 4 gpurun_out/r5_c42_tests.log | cut -c1-300
for t in 0 524288 0 524288; do EMU_HIP_TOOLS=1 EMU_GEMM_TUNE=$t timeout 600 python tools/tp_emulate.py 8 48 p2p 0 2>&1 | grep hipGraph; done > gpurun_out/r5_c42_tp8.log
cat gpurun_out/r5_c42_tp8.log
