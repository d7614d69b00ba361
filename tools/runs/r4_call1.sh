#!/bin/bash
# round 4, GPU call 1: new parity tests, GEMM workgroup timelines, PMC passes over the denoise leg, box calibration
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -s -m gpu > $O/r4_full.log 2>&1; echo "fullsize rc $?" >> $O/r4_full.log ) 
tail -3 $O/r4_full.log
( timeout 600 python -m pytest tests/test_gpu_beam.py tests/test_gpu_fused_ln.py tests/test_gpu_model.py -x -q -m gpu -k "beam or row_stats" > $O/r4_beam.log 2>&1; echo "rc $?" >> $O/r4_beam.log )
tail -3 $O/r4_beam.log
EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so timeout 300 python tools/gemm_trace.py --shapes all --cfgs 0 > $O/r4_trace_0.log 2>&1
EMU_HIP_TOOLS=1 EMU_HIP_LIB=$R/emu_amd/csrc/libemu_hip_trace.so timeout 300 python tools/gemm_trace.py --shapes unet --cfgs K,C,B,Q,S > $O/r4_trace_cfgs.log 2>&1
tail -5 $O/r4_trace_0.log
timeout 300 python bench.py --only-denoise --denoise-steps 20 > $O/r4_dn0.json 2> $O/r4_dn0.err
tail -c 600 $O/r4_dn0.json
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_a $O/pmc_b
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a -o a -- python $R/bench.py --only-denoise --denoise-steps 3 > $O/pmc_a.json 2> $O/pmc_a.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b -o b -- python $R/bench.py --only-denoise --denoise-steps 3 > $O/pmc_b.json 2> $O/pmc_b.err
cd $R
python tools/pmc_kernels.py "round 4, denoise leg (bench.py --only-denoise --denoise-steps 3), two PMC passes" $O/pmc_a $O/pmc_b > $O/r04_denoise_pmc_kernels.json 2> $O/r04_denoise_pmc_kernels.txt
head -30 $O/r04_denoise_pmc_kernels.txt
tail -3 $O/pmc_a.err $O/pmc_b.err
rm -rf $O/pmc_a $O/pmc_b
