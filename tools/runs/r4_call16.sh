#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so
for fx in "" ln lnvt; do
echo "---- fx=$fx"
timeout 300 python tools/gemm_trace.py --shapes unet --cfgs 0,C --fx "$fx" --only "qkv 32" 2>&1 | grep -A5 "qkv 32" | grep "==\|first k\|main loop\|epilogue" | cut -c1-150
done
