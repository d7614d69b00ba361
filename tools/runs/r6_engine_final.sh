# round 6: the persistent engine's record -- single ops, chains, the whole-layer timeline, the shard emulation with mode 4
export EMU_ENGINE_TIMEOUT_MS=50
for tp in 8 4; do timeout 300 python tools/engine_probe.py $tp 30 2>&1 | grep "^tp"; done
timeout 200 python tools/engine_trace.py 8 layer 2>&1 | tail -22
timeout 600 python tools/tp_emulate.py 8 32 p2p 0,4 2>&1 | grep "hipGraph"
timeout 600 python tools/tp_emulate.py 4 32 p2p 0,4 2>&1 | grep "hipGraph"
