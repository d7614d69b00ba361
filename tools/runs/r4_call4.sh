#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
timeout 600 python tools/gemm_ab.py --cfgs 0,B,C,K,S,P,Q,H --filter unet --no-check > $O/r4_ab_unet.log 2>&1
cat $O/r4_ab_unet.log
timeout 600 python tools/gemm_ab.py --cfgs 0,B,C,K,S,P,Q,H --filter vit --no-check > $O/r4_ab_vit.log 2>&1
cat $O/r4_ab_vit.log
timeout 600 python tools/gemm_ab.py --cfgs 0,B,C,S,P,Q,H --filter prefill --no-check > $O/r4_ab_prefill.log 2>&1
cat $O/r4_ab_prefill.log
timeout 600 python tools/gemm_ab.py --cfgs 0,B,C,K,S,P,Q --filter conv --no-check > $O/r4_ab_conv.log 2>&1
cat $O/r4_ab_conv.log
