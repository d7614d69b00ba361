#!/bin/bash
# round 4, call 22: do the eagerly launched ViT encode / prefill pay for their launches?  (same process order: eager, graph, eager, graph)
cd /root/repo
mkdir -p gpurun_out
for i in 1 2; do
timeout 300 python tools/vit_time.py 8 2>&1 | grep "vit encode"
timeout 300 python tools/vit_time.py 8 --graph 2>&1 | grep "vit encode"
done > gpurun_out/r4_c22_vit_graph.log
cat gpurun_out/r4_c22_vit_graph.log
for i in 1 2; do
timeout 300 python tools/prefill_time.py 770 6 2>&1 | grep "prefill S"
timeout 300 python tools/prefill_time.py 770 6 --graph 2>&1 | tail -n 2
done > gpurun_out/r4_c22_prefill_graph.log
cat gpurun_out/r4_c22_prefill_graph.log
