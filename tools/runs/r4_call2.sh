#!/bin/bash
# round 4, GPU call 2: staged epilogue -- bit-identity + parity tests, traces on/off, denoise / full bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_gemm_cfgs.py -x -q -m gpu -k "staged" > $O/r4_staged.log 2>&1; echo "rc $?" >> $O/r4_staged.log )
tail -n 4 $O/r4_staged.log
( timeout 1200 python -m pytest tests/test_gpu_gemm_cfgs.py tests/test_gpu_fused_ln.py tests/test_gpu_beam.py tests/test_gpu_unet.py tests/test_gpu_unet_truewidth.py tests/test_gpu_fp8.py -x -q -m gpu > $O/r4_gemm_tests.log 2>&1; echo "rc $?" >> $O/r4_gemm_tests.log )
tail -n 4 $O/r4_gemm_tests.log
export EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so
timeout 300 python tools/gemm_trace.py --shapes all --cfgs 0 > $O/r4_trace_staged.log 2>&1
timeout 300 python tools/gemm_trace.py --shapes unet --cfgs K,C,B,Q > $O/r4_trace_staged_cfgs.log 2>&1
timeout 300 python tools/gemm_trace.py --shapes unet --cfgs 0,Q --tune 8 > $O/r4_trace_direct.log 2>&1
unset EMU_HIP_TOOLS EMU_HIP_LIB
grep "==" $O/r4_trace_staged.log | cut -c1-120
timeout 300 python bench.py --only-denoise --denoise-steps 20 > $O/r4_dn1.json 2> $O/r4_dn1.err
python -c "import json;d=json.load(open('$O/r4_dn1.json'));print('denoise ms/step',d['ms_per_step'])"
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu > $O/r4_full.log 2>&1; echo "fullsize rc $?" >> $O/r4_full.log ) 
tail -n 3 $O/r4_full.log
timeout 600 python bench.py > $O/r4_bench1.json 2> $O/r4_bench1.err
tail -c 300 $O/r4_bench1.json
