#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_beam.py tests/test_gpu_emu1.py tests/test_gpu_tp_multiproc.py -x -q -m gpu > $O/r4_tests10.log 2>&1; echo "rc $?" >> $O/r4_tests10.log )
tail -n 12 $O/r4_tests10.log
timeout 600 python bench.py --no-denoise --no-legs --no-fp8 --no-cpu-baseline > $O/r4_bench_beam.json 2> $O/r4_bench_beam.err
python -c "
import json;d=json.load(open('$O/r4_bench_beam.json'));print('decode',d['value'],'beam',d.get('beam_search_5'))"
tail -n 5 $O/r4_bench_beam.err
