#!/bin/bash
# same-box kernel stats of the denoise leg: round-3 behaviour (tune 248) vs now (tune 0)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for T in 0 248; do
  rm -rf $O/prof_t$T
  EMU_HIP_TOOLS=1 EMU_GEMM_TUNE=$T timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_t$T -o dn -- python $R/bench.py --only-denoise --denoise-steps 12 > $O/prof_t$T.json 2> $O/prof_t$T.err
  python $R/tools/kernel_stats.py $O/prof_t$T 60 > $O/r04_denoise_kernel_stats_tune$T.csv
  python -c "import json;d=json.load(open('$O/prof_t$T.json'));print('tune $T denoise ms/step (profiled run)',d['ms_per_step'])"
  rm -rf $O/prof_t$T
done
cd $R
( timeout 600 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "share" > $O/r4_tests6.log 2>&1; echo "rc $?" >> $O/r4_tests6.log ); tail -n 3 $O/r4_tests6.log
