#!/bin/bash
# HBM traffic of the prefill GEMMs from PMC counters (separate --pmc pass, kernel trace only, gfx950 FETCH_SIZE correction), tied to the
# GEMM sources by hash.  Run on the MI355X box from the repo root; writes gpurun_out/r06_prefill_gemm_pmc_traffic.json (copy it to profiles/).
set -e
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_pf
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_pf -- python $R/bench.py --pmc-prefill 2 > $R/gpurun_out/pmc_prefill_bench.json 2> $R/gpurun_out/pmc_prefill.err
cd $R
python tools/pmc_gemm_traffic.py /tmp/prof_pf gpurun_out/pmc_prefill_bench.json > gpurun_out/r06_prefill_gemm_pmc_traffic.json
tail -n 4 gpurun_out/r06_prefill_gemm_pmc_traffic.json
