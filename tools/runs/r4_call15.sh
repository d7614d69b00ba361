#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so timeout 900 python tools/unet_trace.py > $O/r4_unet_trace.log 2>&1
cat $O/r4_unet_trace.log | grep -v amdgpu
