set -u
mkdir -p gpurun_out/r4
for round in 1 2 3; do
  for v in base ilp; do
    if [ $v = ilp ]; then export SRACK_JIT_OPTS="-mllvm -amdgpu-sched-strategy=max-ilp"; else unset SRACK_JIT_OPTS; fi
    for w in cfg3_poly cfg4 p3 p4; do
      timeout 600 python bench.py --no-cpu --no-side-configs --workload $w --steps 10 --warmup 2 > gpurun_out/r4/ilp_${v}_${w}_$round.json 2>gpurun_out/r4/ilp.err
      python - gpurun_out/r4/ilp_${v}_${w}_$round.json $v $w <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "ms/step %.3f kernel_ms %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"]), d["config"]["program"][-50:])
PY
    done
  done
done
