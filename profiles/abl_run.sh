#!/bin/bash
# A/B the tuning builds under build/abl/*.so against the production library (per-kernel averages)
for f in prod build/abl/*.so; do
  echo "== $f"
  if [ "$f" != prod ]; then export GSD_LIB_PATH=$PWD/$f; else unset GSD_LIB_PATH; fi
  bash profiles/run_prof.sh abl_$(basename $f .so) --steps 2 --warmup 1 | grep -E "wgrad3x3|frames" | cut -c1-150
done
