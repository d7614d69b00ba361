#!/bin/bash
# round 4, call 25: rocprofv3 kernel statistics of the denoise leg and of the decode loop (final sources); FETCH_SIZE of the prefill GEMMs
cd /tmp && export TMPDIR=/tmp
R=/root/repo; mkdir -p $R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dn -- python $R/bench.py --only-denoise --denoise-steps 12 --no-fp8 > $R/gpurun_out/r4_c25_denoise_bench.json 2> $R/gpurun_out/r4_c25_denoise.err
python $R/tools/kernel_stats.py /tmp/prof_dn 60 > $R/gpurun_out/r4_c25_denoise_kernel_stats.csv
head -n 12 $R/gpurun_out/r4_c25_denoise_kernel_stats.csv | cut -c1-140
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -- python $R/bench.py --no-denoise --no-legs --no-beam --no-fp8 --no-cpu-baseline > $R/gpurun_out/r4_c25_decode_bench.json 2> $R/gpurun_out/r4_c25_decode.err
python $R/tools/kernel_stats.py /tmp/prof_dec 60 > $R/gpurun_out/r4_c25_decode_kernel_stats.csv
head -n 10 $R/gpurun_out/r4_c25_decode_kernel_stats.csv | cut -c1-140
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pf -- python $R/bench.py --pmc-prefill 2 > $R/gpurun_out/r4_c25_pmc_prefill_bench.json 2> $R/gpurun_out/r4_c25_pmc_prefill.err
cd $R && python tools/pmc_gemm_traffic.py /tmp/prof_pf gpurun_out/r4_c25_pmc_prefill_bench.json > gpurun_out/r04_prefill_gemm_pmc_traffic.json
tail -n 12 gpurun_out/r04_prefill_gemm_pmc_traffic.json
