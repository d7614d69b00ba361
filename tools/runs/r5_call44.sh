#!/bin/bash
# round 5, call 44: the whole GPU suite on the final sources (gemv_merge.hip opt-in, 2-D tile blocks, two-lane prefill), smoke()
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_c44_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r5_c44_tests.log | tail -2
timeout 200 python __graft_entry__.py --smoke 2>&1 | tail -1
