#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_unet.py tests/test_gpu_unet_truewidth.py -x -q -m gpu > $O/r4_tests9.log 2>&1; echo "rc $?" >> $O/r4_tests9.log )
tail -n 4 $O/r4_tests9.log
timeout 900 python tools/unet_ab.py 20 7,7p0,7p2,7t32,7t32p0,7t248p0 3 > $O/r4_unet_ab2.log 2>&1
tail -n 3 $O/r4_unet_ab2.log
