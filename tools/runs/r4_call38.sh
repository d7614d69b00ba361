#!/bin/bash
# round 4, call 38: the two PMC passes over the denoise leg again, on the final sources (cf. r04_denoise_pmc_kernels_baseline.*)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_a $O/pmc_b
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a -o a -- python $R/bench.py --only-denoise --denoise-steps 3 --no-fp8 > $O/pmc_a.json 2> $O/pmc_a.err
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_b -o b -- python $R/bench.py --only-denoise --denoise-steps 3 --no-fp8 > $O/pmc_b.json 2> $O/pmc_b.err
cd $R
python tools/pmc_kernels.py "round 4 final sources, denoise leg (bench.py --only-denoise --denoise-steps 3 --no-fp8), two PMC passes" $O/pmc_a $O/pmc_b > $O/r04_denoise_pmc_kernels_final.json 2> $O/r04_denoise_pmc_kernels_final.txt
head -24 $O/r04_denoise_pmc_kernels_final.txt | cut -c1-200
tail -2 $O/pmc_a.err $O/pmc_b.err
rm -rf $O/pmc_a $O/pmc_b
