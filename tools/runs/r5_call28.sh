#!/bin/bash
# round 5, call 28: rocprofv3 kernel statistics of a TP = 8 shard's decode steps (launches, p2p all-reduce): where the 52.7 us per layer go
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tp8 -- python $R/tools/tp_emulate.py 8 32 p2p 0 > $O/r5_c28_tp8.log 2>&1
python $R/tools/kernel_stats.py /tmp/prof_tp8 24 > $O/r5_c28_tp8_kernel_stats.csv
cut -c1-200 $O/r5_c28_tp8_kernel_stats.csv | head -26
grep "ms/token" $O/r5_c28_tp8.log
