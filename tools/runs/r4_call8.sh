#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python tools/mall_prefetch_probe.py > gpurun_out/r4_mall_probe.log 2>&1
cat gpurun_out/r4_mall_probe.log
