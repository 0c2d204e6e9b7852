for v in 1 64 1024; do for w in cfg3 p3; do for b in 0 1024; do
extra=""; [ $w = p3 ] && extra="--no-frames"
timeout 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-side-configs --workload $w $extra --voices $v --block $b 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$w V=$v block=$b: %.3f ms per second of audio; kernel %s x%d' % (d['ms_per_step'], r['kernel'], r['launches_per_step']))"
done; done; done
