#!/bin/bash
# usage: bash profiles/quick_bench.sh [bench args]  -- prints ms/step and roofline numbers
python bench.py --no-cpu-baseline "$@" 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('frames/s',d['value'],'ms/step',d['ms_per_step'],'dom TF',r['achieved'],'allconv TF',r['all_conv3x3_tflops'],'share',r['conv3x3_share_of_step'])"
