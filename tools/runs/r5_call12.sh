#!/bin/bash
# round 5, call 12: PMC traffic passes (decode GEMVs, prefill GEMMs) and rocprofv3 kernel statistics of the decode / denoise legs
cd /root/repo
mkdir -p gpurun_out
R=/root/repo
bash tools/pmc_traffic.sh > gpurun_out/r5_c12_pmc_gemv.log 2>&1
tail -n 6 gpurun_out/r5_c12_pmc_gemv.log
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pf -- python $R/bench.py --pmc-prefill 2 > $R/gpurun_out/r5_c12_pmc_prefill_bench.json 2> $R/gpurun_out/r5_c12_pmc_prefill.err
cd $R && python tools/pmc_gemm_traffic.py /tmp/prof_pf gpurun_out/r5_c12_pmc_prefill_bench.json > gpurun_out/r05_prefill_gemm_pmc_traffic.json
tail -n 4 gpurun_out/r05_prefill_gemm_pmc_traffic.json
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dn -- python $R/bench.py --only-denoise --denoise-steps 12 --no-fp8 > $R/gpurun_out/r5_c12_denoise_bench.json 2> $R/gpurun_out/r5_c12_denoise.err
python $R/tools/kernel_stats.py /tmp/prof_dn 60 > $R/gpurun_out/r05_denoise_kernel_stats.csv
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $R/bench.py --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline > $R/gpurun_out/r5_c12_decode_bench.json 2> $R/gpurun_out/r5_c12_decode.err
python $R/tools/kernel_stats.py /tmp/prof_dec 60 > $R/gpurun_out/r05_bench_decode_kernel_stats.csv
head -n 12 $R/gpurun_out/r05_bench_decode_kernel_stats.csv | cut -c1-130
head -n 12 $R/gpurun_out/r05_denoise_kernel_stats.csv | cut -c1-130
