#!/bin/bash
# round 5, call 6: CFG-pair split tests, finished-hypothesis beam fixture on the GPU, decode fused tests again (epoch keys), model suite
cd /root/repo
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_model.py -x -q -k finished_hypotheses > gpurun_out/r5_c6_tests_a.log 2>&1
tail -n 12 gpurun_out/r5_c6_tests_a.log
