export EMU_ENGINE_TIMEOUT_MS=20
for nl in 1 2; do for dbg in 1 257 513 1025 1281; do echo "== loaders $nl dbg $dbg (1 = no math; +256 lag 4; +512 lag 2; +1024 default policy)"; EMU_ENGINE_LOADERS=$nl EMU_ENGINE_DBG=$dbg timeout 300 python tools/engine_probe.py 8 30 2>&1 | grep "^tp" | sed -n '3p;5p'; done; done
