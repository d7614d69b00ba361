#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu -k "flash or attn or unet" > $O/r4_tests14.log 2>&1; echo "rc $?" >> $O/r4_tests14.log )
tail -n 4 $O/r4_tests14.log
timeout 900 python tools/unet_ab.py 20 7,7t4096,7t8192,7t4288 3 > $O/r4_unet_ab3.log 2>&1
tail -n 2 $O/r4_unet_ab3.log
timeout 300 python tools/kbench.py --filter flash > $O/r4_kbench_flash.log 2>&1; tail -n 12 $O/r4_kbench_flash.log
