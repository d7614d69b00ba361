#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 900 python -m pytest tests/test_gpu_fused_ln.py tests/test_gpu_unet.py tests/test_gpu_unet_truewidth.py -x -q -m gpu > $O/r4_tests17.log 2>&1; echo "rc $?" >> $O/r4_tests17.log )
tail -n 4 $O/r4_tests17.log
export EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so
timeout 300 python tools/gemm_trace.py --shapes unet --cfgs 0 --fx lnvt --only "qkv 32" 2>&1 | grep "==\|first k\|main loop\|epilogue" | cut -c1-150
timeout 300 python tools/gemm_trace.py --shapes unet --cfgs 0 --fx lnvt --only "qkv 32" --tune 16384 2>&1 | grep "==\|first k\|main loop\|epilogue" | cut -c1-150
unset EMU_HIP_TOOLS EMU_HIP_LIB
timeout 900 python tools/unet_ab.py 20 7,7t16384 3 > $O/r4_unet_ab4.log 2>&1
tail -n 1 $O/r4_unet_ab4.log
