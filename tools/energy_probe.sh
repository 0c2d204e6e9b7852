#!/bin/bash
# tools/energy_probe.sh — package power and sclk (rocm-smi, 4 samples per second) while each variant of tools/energybench runs for 5 s
# -> gpurun_out/profiles/r04_energy.txt: per variant the bench's JSON line and the median power / clock of the samples taken while it ran
OUT=gpurun_out/profiles; mkdir -p $OUT
[ -n "${ENERGY_VARIANTS:-}" ] && OUTF=$OUT/r04_energy_extra.txt || OUTF=$OUT/r04_energy.txt
echo "idle: $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power \(W\)|sclk' | sed -e 's/.*: //' | tr '\n' ' ')" > $OUTF
for v in ${ENERGY_VARIANTS:-13 0 12 1 2 3 4 5 6 14 7 8 9 10 11 15 16 17}; do
  ./tools/energybench $v 5 > /tmp/eb.json &
  BP=$!
  sleep 1.2
  S=""
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    kill -0 $BP 2>/dev/null || break
    S="$S $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power \(W\)|sclk' | sed -e 's/.*: //' | tr '\n' ',')"
    sleep 0.25
  done
  wait $BP
  echo "$(cat /tmp/eb.json) samples:$S" >> $OUTF
done
cat $OUTF
