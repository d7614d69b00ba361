#!/bin/bash
# round 5, call 2: per-workgroup timelines of the fused decode launch (trace library), tp = 1 and a tp = 8 shard
cd /root/repo
mkdir -p gpurun_out
export EMU_HIP_TOOLS=1 EMU_HIP_LIB=emu_amd/csrc/libemu_hip_trace.so
timeout 300 python tools/decode_trace.py 1 4 1 > gpurun_out/r5_c2_trace_tp1.log 2>&1
tail -n 24 gpurun_out/r5_c2_trace_tp1.log
timeout 300 python tools/decode_trace.py 8 4 2 > gpurun_out/r5_c2_trace_tp8.log 2>&1
tail -n 24 gpurun_out/r5_c2_trace_tp8.log
