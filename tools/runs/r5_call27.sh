#!/bin/bash
# round 5, call 27: what the two-lane cut costs at S = 770 (BASELINE configs[1]'s prompt; lanes of 512 + 258 rows) for TP = 8 / 2 shards
cd /root/repo
mkdir -p gpurun_out
timeout 600 python tools/tp_prefill_emulate.py 8 770 4 rccl 512 2>&1 | grep "summary\|Error\|error" > gpurun_out/r5_c27_prefill_770.log
timeout 900 python tools/tp_prefill_emulate.py 2 770 3 rccl 512 2>&1 | grep "summary\|Error\|error" >> gpurun_out/r5_c27_prefill_770.log
cat gpurun_out/r5_c27_prefill_770.log
