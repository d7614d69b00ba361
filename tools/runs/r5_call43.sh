#!/bin/bash
# round 5, call 43: merged o_proj, second version (one merge per workgroup through LDS): bit-identity tests, TP = 8 shard with / without (bit 19)
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_decode_fused.py -x -q > gpurun_out/r5_c43_tests.log 2>&1
tail -n 2 gpurun_out/r5_c43_tests.log | cut -c1-300
for t in 0 524288; do EMU_HIP_TOOLS=1 EMU_GEMM_TUNE=$t timeout 600 python tools/tp_emulate.py 8 64 p2p 0 2>&1 | grep hipGraph; done > gpurun_out/r5_c43_tp8.log
cat gpurun_out/r5_c43_tp8.log
