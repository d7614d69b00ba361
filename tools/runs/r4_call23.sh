#!/bin/bash
# round 4, call 23: the whole GPU suite on the final sources
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -n 12 > gpurun_out/r4_c23_gpu_suite.log
cat gpurun_out/r4_c23_gpu_suite.log
