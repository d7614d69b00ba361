#!/bin/bash
# round 5, call 33: the whole GPU suite on the round's final sources, then smoke()
cd /root/repo
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r5_c33_tests.log 2>&1
tail -n 6 gpurun_out/r5_c33_tests.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
