#!/bin/bash
# round 5, call 39: two-lane vs serial prefill layer by layer at full depth on a common input (TP = 8 shard)
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/tp_overlap_layerwise.py 8 1544 > gpurun_out/r5_c39_layerwise.log 2>&1
tail -n 16 gpurun_out/r5_c39_layerwise.log
