P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), round(d["roofline"]["kernel_ms"]*d["roofline"]["launches_per_step"],2), round(d["roofline"]["voice_samples_per_s_kernel"]/1e9,1), round(d["roofline"]["frac"],3), d["config"]["program"][-40:])'
for args in "$@"; do
  echo "== $args"; timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu $args 2>&1 | python -c "$P"
done
