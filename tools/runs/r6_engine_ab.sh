# round 6: persistent engine A/B (loaders 1 / 2, with / without the consumers' arithmetic), then per-CU timelines
export EMU_ENGINE_TIMEOUT_MS=20
for nl in 1 2; do for dbg in 0 1; do echo "== loaders $nl dbg $dbg"; EMU_ENGINE_LOADERS=$nl EMU_ENGINE_DBG=$dbg timeout 300 python tools/engine_probe.py ${TP:-8} 30 2>&1 | grep "^tp"; done; done
for c in gu mlp; do echo "== trace $c loaders 1"; EMU_ENGINE_LOADERS=1 timeout 200 python tools/engine_trace.py ${TP:-8} $c 2>&1 | tail -9; done
