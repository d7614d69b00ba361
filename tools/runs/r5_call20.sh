#!/bin/bash
# round 5, call 20: the stand-alone P2P all-reduce without cache-wide fences: latency with 1 / 2 / 4 rank processes on the GPU, the
# multi-rank engine tests (2 / 4 / 8 ranks), the TP = 8 shard's per-rank cost
cd /root/repo
mkdir -p gpurun_out
for n in 1 2 4; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) tools/p2p_time.py 2>&1 | grep -i "all-reduce\|us per" | head -4
done > gpurun_out/r5_c20_p2p_time.log
cat gpurun_out/r5_c20_p2p_time.log
timeout 1500 python -m pytest tests/test_gpu_tp_multiproc.py tests/test_gpu_decode_fused.py -x -q > gpurun_out/r5_c20_tests.log 2>&1
tail -n 3 gpurun_out/r5_c20_tests.log
timeout 600 python tools/tp_emulate.py 8 32 p2p 0,3 2>&1 | grep hipGraph > gpurun_out/r5_c20_tp8.log
cat gpurun_out/r5_c20_tp8.log
