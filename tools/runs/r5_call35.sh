#!/bin/bash
# round 5, call 35: every tile configuration on the shapes where the vendor library leads (call 34): is the dispatch's choice the best of ours?
cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/gemm_ab.py --cfgs 0,B,C,K,S,P,Q,H --no-check --iters 20 --shapes "1025,6144,1792,0;2048,3840,1280,0;2048,1280,5120,1;8192,5120,640,5;4100,15360,1792,4;4100,1792,15360,1;1544,35840,6656,2;1544,6656,17920,1" > gpurun_out/r5_c35_gemm_ab.log 2>&1
cat gpurun_out/r5_c35_gemm_ab.log | cut -c1-220
