#!/bin/bash
# round 5, call 15: the whole GPU suite + smoke on the current sources
cd /root/repo
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -x -q > gpurun_out/r5_c15_tests.log 2>&1
tail -n 4 gpurun_out/r5_c15_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/r5_c15_smoke.log 2>&1
tail -n 2 gpurun_out/r5_c15_smoke.log
