#!/bin/bash
# round 5, call 40: kernel trace of serial vs two-lane prefills of a TP = 8 shard: do the lanes' kernels execute side by side?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_lanes -- python $R/tools/lane_overlap.py run 8 1544 > $O/r5_c40_run.log 2>&1
python $R/tools/lane_overlap.py read /tmp/prof_lanes > $O/r5_c40_lane_overlap.log 2>&1
cat $O/r5_c40_lane_overlap.log; tail -2 $O/r5_c40_run.log
