#!/bin/bash
# round 5, call 4: fused decode layers, wave-level waits + all-up-front q / gate-up roles: tests, timing, timelines
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode_fused.py -x -q > gpurun_out/r5_c4_tests.log 2>&1
tail -n 5 gpurun_out/r5_c4_tests.log
timeout 600 python tools/tp_emulate.py 8 32 p2p 0,1,2 2>&1 | grep hipGraph > gpurun_out/r5_c4_tp8.log
cat gpurun_out/r5_c4_tp8.log
timeout 900 python tools/tp_emulate.py 1 24 none 0,1 2>&1 | grep hipGraph > gpurun_out/r5_c4_tp1.log
cat gpurun_out/r5_c4_tp1.log
export EMU_HIP_TOOLS=1 EMU_HIP_LIB=emu_amd/csrc/libemu_hip_trace.so
timeout 300 python tools/decode_trace.py 1 3 1 > gpurun_out/r5_c4_trace_tp1.log 2>&1
tail -n 11 gpurun_out/r5_c4_trace_tp1.log
timeout 300 python tools/decode_trace.py 8 3 2 > gpurun_out/r5_c4_trace_tp8.log 2>&1
tail -n 11 gpurun_out/r5_c4_trace_tp8.log
