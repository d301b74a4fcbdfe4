#!/bin/bash
# usage: bash profiles/ab_env.sh VAR   -- per-kernel averages without / with env VAR=1
for v in 0 1; do
  if [ $v = 1 ]; then export $1=1; fi
  echo "== $1=$v"; bash profiles/run_prof.sh ab$v --steps 2 --warmup 1 | grep -E "wgrad3x3|frames" | cut -c1-150
done
