# round 6: what the hand-offs of a whole-layer chain cost against the loaders' depth (fills in flight = queueing at the memory side)
export EMU_ENGINE_TIMEOUT_MS=50
for nl in 2 3 4; do for dbg in 0 4096 4; do echo "== loaders $nl dbg $dbg (4096 = one fill in flight per loader always; 4 = never thinned)"; EMU_ENGINE_LOADERS=$nl EMU_ENGINE_DBG=$dbg timeout 300 python tools/engine_probe.py ${TP:-8} 30 2>&1 | grep "^tp" | tail -1; done; done
EMU_ENGINE_LOADERS=3 EMU_ENGINE_DBG=4096 timeout 200 python tools/engine_trace.py 8 layer 2>&1 | tail -13
