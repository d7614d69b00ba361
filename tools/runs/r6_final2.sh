# round 6, second half (four-wave GEMM tile): final-source records -- GPU suite, PMC traffic passes, kernel statistics of the decode,
# denoise and prefill legs, the vendor-library yardstick on the same box, the bench line
set -x
R=$(pwd)
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | grep -E 'passed|failed|error' | tail -3 > gpurun_out/r06_gpu_suite_final.log; cat gpurun_out/r06_gpu_suite_final.log
bash tools/pmc_traffic.sh > gpurun_out/r6_final_pmc_gemv.log 2>&1; tail -n 3 gpurun_out/r6_final_pmc_gemv.log
cp gpurun_out/r06_gemv_pmc_traffic.json profiles/ 2>/dev/null
bash tools/pmc_prefill_traffic.sh > gpurun_out/r6_final_pmc_prefill.log 2>&1; tail -n 3 gpurun_out/r6_final_pmc_prefill.log
cp gpurun_out/r06_prefill_gemm_pmc_traffic.json profiles/ 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dn -- python $R/bench.py --only-denoise --denoise-steps 12 --no-fp8 > $R/gpurun_out/r6_final_denoise_bench.json 2> $R/gpurun_out/r6_final_denoise.err
python $R/tools/kernel_stats.py /tmp/prof_dn 60 > $R/gpurun_out/r06_denoise_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $R/bench.py --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline > $R/gpurun_out/r6_final_decode_bench.json 2> $R/gpurun_out/r6_final_decode.err
python $R/tools/kernel_stats.py /tmp/prof_dec 60 > $R/gpurun_out/r06_bench_decode_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pf2 -- python $R/tools/prefill_time.py > $R/gpurun_out/r6_final_prefill_time.log 2>&1
python $R/tools/kernel_stats.py /tmp/prof_pf2 30 > $R/gpurun_out/r06_prefill_kernel_stats.csv
cd $R
python tools/gemm_vs_library.py --filter "prefill" > gpurun_out/r06_gemm_vs_vendor_library.log 2>&1
python tools/gemm_vs_library.py --filter "square" >> gpurun_out/r06_gemm_vs_vendor_library.log 2>&1
tail -20 gpurun_out/r06_gemm_vs_vendor_library.log
EMU_TUNES=0,2097152,0,2097152 python tools/prefill_time.py 2>&1 | tail -4 > gpurun_out/r06_prefill_ab_final.log
EMU_TUNES=0,2097152 python tools/prefill_time.py 1544 2>&1 | tail -2 >> gpurun_out/r06_prefill_ab_final.log; cat gpurun_out/r06_prefill_ab_final.log
python bench.py > gpurun_out/r06_bench_tp1.json 2> gpurun_out/r06_bench_tp1.err
tail -c 1500 gpurun_out/r06_bench_tp1.json
EMU_TP_SHARED_GPU=1 python bench.py --gpus 2 --steps 16 --warmup 4 --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline > gpurun_out/r06_bench_tp2_shared_gpu_validation.json 2> gpurun_out/r06_bench_tp2.err; tail -c 600 gpurun_out/r06_bench_tp2_shared_gpu_validation.json
