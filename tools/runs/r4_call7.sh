#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 600 python tools/gemm_ab.py --cfgs 0,B,C,K,S,Q --filter unet --no-check > $O/r4_ab_unet_warm.log 2>&1
cat $O/r4_ab_unet_warm.log
timeout 900 python tools/gemm_ab.py --cfgs 0,B,C,K,S,Q --filter unet --no-check --cold-mb 800 > $O/r4_ab_unet_cold.log 2>&1
cat $O/r4_ab_unet_cold.log
