export EMU_ENGINE_TIMEOUT_MS=20
for nl in 2 3 4 5; do for dbg in 1 0; do echo "== 8 waves, loaders $nl dbg $dbg (1 = no math)"; EMU_ENGINE_LOADERS=$nl EMU_ENGINE_DBG=$dbg timeout 300 python tools/engine_probe.py 8 30 2>&1 | grep "^tp" | sed -n '1p;3p;5p'; done; done
