#!/bin/bash
# round 4, GPU call 3: new tile configs (D, E, F, G), 2-D XCD blocks on/off, kernel stats of the denoise leg
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_gemm_cfgs.py -x -q -m gpu -k "staged or unsplit" > $O/r4_staged.log 2>&1; echo "rc $?" >> $O/r4_staged.log )
tail -n 4 $O/r4_staged.log
export EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so
timeout 400 python tools/gemm_trace.py --shapes unet --cfgs 0,K,F,D,B,G,C,E > $O/r4_trace_c3.log 2>&1
timeout 300 python tools/gemm_trace.py --shapes unet --cfgs K,F,C --tune 16 > $O/r4_trace_c3_no2d.log 2>&1
unset EMU_HIP_TOOLS EMU_HIP_LIB
grep "==" $O/r4_trace_c3.log | cut -c1-110
echo "--- no 2-D blocks"
grep "==" $O/r4_trace_c3_no2d.log | cut -c1-110
timeout 300 python bench.py --only-denoise --denoise-steps 20 > $O/r4_dn2.json 2> $O/r4_dn2.err
python -c "import json;d=json.load(open('$O/r4_dn2.json'));print('denoise ms/step',d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_dn
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_dn -o dn -- python $R/bench.py --only-denoise --denoise-steps 12 > $O/prof_dn.json 2> $O/prof_dn.err
cd $R
python tools/kernel_stats.py $O/prof_dn > $O/r04_denoise_kernel_stats_v1.csv 2>/dev/null || ls $O/prof_dn
head -40 $O/r04_denoise_kernel_stats_v1.csv | cut -c1-200
rm -rf $O/prof_dn/*.db
( timeout 900 python -m pytest tests/test_gpu_gemm_cfgs.py tests/test_gpu_fused_ln.py -x -q -m gpu > $O/r4_gemm_tests.log 2>&1; echo "rc $?" >> $O/r4_gemm_tests.log )
tail -n 4 $O/r4_gemm_tests.log
