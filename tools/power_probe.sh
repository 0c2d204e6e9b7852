# tools/power_probe.sh "<bench args>" — socket power and clocks (rocm-smi) sampled while bench.py runs; is the kernel at the power limit?
ARGS="$1"
python bench.py --steps ${PP_STEPS:-1500} --warmup 2 --no-cpu $ARGS > /tmp/pp_bench.log 2>&1 &
BP=$!
for i in $(seq 1 40); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|mclk" | sed -e 's/.*: //' | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
tail -1 /tmp/pp_bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step', round(d['ms_per_step'],2))"
rocm-smi --showmaxpower 2>/dev/null | grep -i "Max Graphics"
