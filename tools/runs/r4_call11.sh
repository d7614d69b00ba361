#!/bin/bash
# TP = 8 dry run with all 8 rank processes on the one GPU (validation of rendezvous, IPC handle exchange for 8 ranks, 56-head
# padding, 7-head shard kernels, P2P all-reduce with 8 participants); not a measurement
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0 EMU_TP_SHARED_GPU=1
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 8 --warmup 2 --no-legs --no-denoise --no-fp8 --no-beam --no-cpu-baseline > $O/r4_tp8_shared.json 2> $O/r4_tp8_shared.err
echo "rc $?"
tail -c 1500 $O/r4_tp8_shared.json
tail -n 15 $O/r4_tp8_shared.err
timeout 600 python tools/tp_emulate.py 8 32 p2p > $O/r4_tp_emulate8.log 2>&1; tail -n 5 $O/r4_tp_emulate8.log
