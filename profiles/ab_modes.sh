for m in 1 3; do echo "== GSD_WGRAD_MODE=$m"; GSD_WGRAD_MODE=$m bash profiles/run_prof.sh abm$m --steps 2 --warmup 1 | grep -E "wgrad3x3|frames" | cut -c1-150; done
