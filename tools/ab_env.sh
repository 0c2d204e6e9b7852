#!/bin/bash
# A/B of one environment knob on one box: bash tools/ab_env.sh VAR "v1 v2 ..." <bench.py args>   (three rounds, alternating)
VAR=$1; VALS=$2; shift 2
for round in 1 2 3; do for v in $VALS; do
  env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$VAR=$v  ms/step %.3f  kernel %.4f ms x%d' % (d['ms_per_step'], r['kernel_ms'], r['launches_per_step']))"
done; done
