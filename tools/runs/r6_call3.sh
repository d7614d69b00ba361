# round 6, call 3: row-streaming GEMV A/B, attention / combine / all-reduce chain shortening in the decode step
python tools/gemv_rows_ab.py 40 > gpurun_out/r06_gemv_rows_ab.log 2>&1
python tools/tp_emulate.py 8 32 p2p 0 2>&1 | grep hipGraph > gpurun_out/r06_tp_emulate_chain.log
python tools/tp_emulate.py 1 32 none 0 2>&1 | grep hipGraph >> gpurun_out/r06_tp_emulate_chain.log
EMU_GEMM_TUNE=1048576 EMU_HIP_TOOLS=1 python tools/tp_emulate.py 8 32 p2p 0 2>&1 | grep hipGraph >> gpurun_out/r06_tp_emulate_chain.log
EMU_GEMM_TUNE=6291456 EMU_HIP_TOOLS=1 python tools/tp_emulate.py 1 32 none 0 2>&1 | grep hipGraph >> gpurun_out/r06_tp_emulate_chain.log
EMU_GEMM_TUNE=2097152 EMU_HIP_TOOLS=1 python tools/tp_emulate.py 1 32 none 0 2>&1 | grep hipGraph >> gpurun_out/r06_tp_emulate_chain.log
EMU_GEMM_TUNE=6291456 EMU_HIP_TOOLS=1 python tools/tp_emulate.py 8 32 p2p 0 2>&1 | grep hipGraph >> gpurun_out/r06_tp_emulate_chain.log
python -m pytest tests/test_gpu_model.py tests/test_gpu_ops.py -x -q 2>&1 | tail -5
cat gpurun_out/r06_gemv_rows_ab.log gpurun_out/r06_tp_emulate_chain.log
