export EMU_ENGINE_TIMEOUT_MS=20
for nl in 1 2; do for dbg in 0 4; do echo "== loaders $nl dbg $dbg"; EMU_ENGINE_LOADERS=$nl EMU_ENGINE_DBG=$dbg timeout 300 python tools/engine_probe.py 8 30 2>&1 | grep "^tp" | tail -1; done; done
echo "== trace mlp loaders 2 nothin"; EMU_ENGINE_LOADERS=2 EMU_ENGINE_DBG=4 timeout 200 python tools/engine_trace.py 8 mlp 2>&1 | tail -9
