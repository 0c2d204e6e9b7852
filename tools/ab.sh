# tools/ab.sh "<bench args>" libA.so libB.so ... — alternate prebuilt variants of libsrack_hip.so on ONE box (boxes differ
# by a few percent, so variants are only comparable within a call); three rounds, ms/step and kernel ms per variant.
ARGS="$1"; shift
cp s-rack_amd/libsrack_hip.so /tmp/_orig.so
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],2), round(d["roofline"]["kernel_ms"]*d["roofline"]["launches_per_step"],2))'
for round in 1 2 3; do
  for lib in "$@"; do
    cp "$lib" s-rack_amd/libsrack_hip.so
    echo -n "$(basename $lib)  "; timeout 200 python bench.py --steps 5 --warmup 1 --no-cpu $ARGS 2>&1 | python -c "$P"
  done
done
cp /tmp/_orig.so s-rack_amd/libsrack_hip.so
