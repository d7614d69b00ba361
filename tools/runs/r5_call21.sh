#!/bin/bash
# round 5, call 21: TP = 8 shard with / without the in-kernel split merge of the decode attention (one launch less per layer)
cd /root/repo
mkdir -p gpurun_out
for t in 0 1 0 1; do EMU_DECODE_TAIL=$t timeout 600 python tools/tp_emulate.py 8 48 p2p 0 2>&1 | grep hipGraph; done > gpurun_out/r5_c21_tp8_tail.log
cat gpurun_out/r5_c21_tp8_tail.log
for t in 0 1; do EMU_DECODE_TAIL=$t timeout 600 python tools/tp_emulate.py 4 48 p2p 0 2>&1 | grep hipGraph; done >> gpurun_out/r5_c21_tp8_tail.log
tail -n 2 gpurun_out/r5_c21_tp8_tail.log
