#!/bin/bash
# A/B of two builds of the library on one box: bash tools/ab_lib.sh libA.so libB.so <bench.py args>  (three rounds, alternating;
# both files under s-rack_amd/, the first is restored as libsrack_hip.so at the end)
A=$1; B=$2; shift 2
D=s-rack_amd
cp $D/libsrack_hip.so /tmp/lib_orig.so
for round in 1 2 3; do for v in $A $B; do
  cp $D/$v /tmp/lib_ab.so; cp /tmp/lib_ab.so $D/libsrack_hip.so
  python bench.py --steps 10 --warmup 3 --no-cpu "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v  ms/step %.3f  kernel %.4f ms x%d' % (d['ms_per_step'], r['kernel_ms'], r['launches_per_step']))"
done; done
cp /tmp/lib_orig.so $D/libsrack_hip.so
