#!/bin/bash
# A/B the tuning builds under build/abl/*.so against the production library, same process conditions
echo "prod"; bash profiles/quick_bench.sh --steps 4 --warmup 2
for f in build/abl/*.so; do echo $f; GSD_LIB_PATH=$PWD/$f bash profiles/quick_bench.sh --steps 4 --warmup 2; done
