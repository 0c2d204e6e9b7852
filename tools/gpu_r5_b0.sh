#!/bin/bash
# Round 5, diagnostic: the seeds pass A's soaks flagged at round 4's HEAD (its workers died writing their JSON, so the merged summary said 0):
# alone and in sequence, under round 4's library (s-rack_amd/libsrack_hip_r4.so) and under this tree's.
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
cp s-rack_amd/libsrack_hip.so /tmp/_new.so
for which in r4 new; do
  if [ $which = r4 ]; then cp s-rack_amd/libsrack_hip_r4.so s-rack_amd/libsrack_hip.so; else cp /tmp/_new.so s-rack_amd/libsrack_hip.so; fi
  ( SOAK_VT=16,48000 DBG_VOICES=1 DBG_FLAGS=0,2,4 timeout 300 python tools/dbg_default.py 51792 51799 51803 51805 55878 ) > $OUT/b0_${which}_single_1s.log 2>&1
  ( SOAK_VT=200,6000 DBG_VOICES=1 DBG_FLAGS=0,2,4 timeout 300 python tools/dbg_default.py 64363 64366 ) > $OUT/b0_${which}_single_v200.log 2>&1
  ( SOAK_VT=16,48000 timeout 400 python tools/fuzz_soak_default.py 51780 51830 ) > $OUT/b0_${which}_range.log 2>&1
  echo "== $which"; grep -h "^seed\|voice \|default modes\|patches" $OUT/b0_${which}_single_1s.log $OUT/b0_${which}_single_v200.log $OUT/b0_${which}_range.log | cut -c1-260 | head -80
done
cp /tmp/_new.so s-rack_amd/libsrack_hip.so
