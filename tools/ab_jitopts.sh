#!/bin/bash
# A/B of SRACK_JIT_OPTS (the specialised kernels' extra compiler options) on one box: bash tools/ab_jitopts.sh "<opts>" <bench.py args>   (three rounds, alternating with none)
OPTS=$1; shift
for round in 1 2 3; do for which in none opts; do
  if [ $which = none ]; then unset SRACK_JIT_OPTS; else export SRACK_JIT_OPTS="$OPTS"; fi
  python bench.py --steps 10 --warmup 3 --no-cpu --no-side-configs "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$which  ms/step %.3f  kernel %.4f ms x%d' % (d['ms_per_step'], r['kernel_ms'], r['launches_per_step']))"
done; done
