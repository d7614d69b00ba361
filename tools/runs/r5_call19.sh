#!/bin/bash
# round 5, call 19: the RCCL all-reduce path (1-rank communicator in the loop) after this round's engine changes: TP = 8 shard, modes 0 / 1
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/tp_emulate.py 8 32 rccl 0,1 2>&1 | grep "ms/token" > gpurun_out/r5_c19_tp8_rccl.log
cat gpurun_out/r5_c19_tp8_rccl.log
