#!/bin/bash
# A/B of the fp32 step time under environment settings (GPU box): bash profiles/ab_step.sh "A=1 B=2" "C=3" ...   ("-" = no setting)
# each setting: batch 32 (10 steps) and batch 8 (10 steps), twice, interleaved
for rep in 1 2; do
  for s in "$@"; do
    [ "$s" = "-" ] && e="" || e="$s"
    b32=$(env $e python bench.py --no-cpu-baseline --no-parity --no-extra --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().split('\n')[-1])['ms_per_step'])")
    b8=$(env $e python bench.py --no-cpu-baseline --no-parity --no-extra --batch 8 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read().strip().split('\n')[-1])['ms_per_step'])")
    echo "[$s] batch 32: $b32 ms   batch 8: $b8 ms"
  done
done
