#!/bin/bash
# HBM traffic of the decode GEMVs from PMC counters, as MI355X_MICROARCH.md prescribes (separate --pmc passes, kernel trace
# only, gfx950 FETCH_SIZE correction).  Run on the MI355X box from the repo root; writes profiles/r02_gemv_pmc_traffic.json.
set -e
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --pmc-mode 6 > $R/gpurun_out/pmc_fetch.json 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --pmc-mode 6 > $R/gpurun_out/pmc_write.json 2> $R/gpurun_out/pmc_write.err
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_fetch.json > profiles/r02_gemv_pmc_traffic.json
cat profiles/r02_gemv_pmc_traffic.json
