#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 1800 python -m pytest tests -x -q -m gpu > $O/r4_pytest_gpu.log 2>&1; echo "rc $?" >> $O/r4_pytest_gpu.log )
tail -n 8 $O/r4_pytest_gpu.log
