#!/bin/bash
# round 5, call 45: the clean-built library (every object recompiled): smoke() and the operator tests
cd /root/repo
mkdir -p gpurun_out
timeout 100 python __graft_entry__.py --smoke 2>&1 | tail -1
timeout 100 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -1
