#!/bin/bash
# round 5, call 30: PMC tables on the round-5 sources -- the denoise leg (two passes), the 256x256 ping-pong GEMM at 8192^3 and at the
# prefill shapes (two passes) -- and the driver-style bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
PB="FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
timeout 600 rocprofv3 --kernel-trace --pmc $PA --output-format csv -d $O/pmc_a -o a -- python $R/bench.py --only-denoise --denoise-steps 3 --no-fp8 > $O/pmc_a.json 2> $O/pmc_a.err
timeout 600 rocprofv3 --kernel-trace --pmc $PB --output-format csv -d $O/pmc_b -o b -- python $R/bench.py --only-denoise --denoise-steps 3 --no-fp8 > $O/pmc_b.json 2> $O/pmc_b.err
for f in square8192 prefill; do
  timeout 300 rocprofv3 --kernel-trace --pmc $PA --output-format csv -d $O/pmc_c -o c_$f -- python $R/tools/fp8_gemm_time.py --filter $f --iters 4 > $O/pmc_c_$f.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc $PB --output-format csv -d $O/pmc_d -o d_$f -- python $R/tools/fp8_gemm_time.py --filter $f --iters 4 > $O/pmc_d_$f.log 2>&1
done
cd $R
python tools/pmc_kernels.py "round 5 final sources, denoise leg (bench.py --only-denoise --denoise-steps 3 --no-fp8), two PMC passes" $O/pmc_a $O/pmc_b > $O/r05_denoise_pmc_kernels.json 2> $O/r05_denoise_pmc_kernels.txt
head -24 $O/r05_denoise_pmc_kernels.txt | cut -c1-220
python tools/pmc_kernels.py "round 5, tools/fp8_gemm_time.py --filter {square8192, prefill} --iters 4 (bf16 and fp8 GEMMs at 8192^3 and at the S = 770 / 1544 prefill shapes), two PMC passes" $O/pmc_c $O/pmc_d > $O/r05_gemm256_pmc.json 2> $O/r05_gemm256_pmc.txt
head -24 $O/r05_gemm256_pmc.txt | cut -c1-220
tail -2 $O/pmc_a.err $O/pmc_c_square8192.log
rm -rf $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_bench_tp1_final_v2.json 2> $O/r05_bench_tp1_final_v2.err
tail -c 600 $O/r05_bench_tp1_final_v2.json; tail -3 $O/r05_bench_tp1_final_v2.err
