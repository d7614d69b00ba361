#!/bin/bash
# HBM traffic of the decode GEMVs from PMC counters, as MI355X_MICROARCH.md prescribes (separate --pmc passes, kernel trace
# only, gfx950 FETCH_SIZE correction).  Run on the MI355X box from the repo root; writes gpurun_out/r06_gemv_pmc_traffic.json (copy it to profiles/).
set -e
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --pmc-mode 6 > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --pmc-mode 6 > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch.json > gpurun_out/r06_gemv_pmc_traffic.json
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
tail -12 gpurun_out/r06_gemv_pmc_traffic.json
