#!/bin/bash
# Round 5, pass P: the NOISE family (noise / sample player / reverb drawn too) under the bound at soak scale — this round's soaks were all of the
# plain family —, with and without per-voice parameter arrays; the two seeds pinned in pass O's wake through pytest
set -u
OUT=gpurun_out/r5; mkdir -p $OUT
( timeout 200 python -m pytest tests/test_gpu_fuzz.py -q -k soak_finds ) > $OUT/p_pinned.log 2>&1; echo "== pinned rc=$?"; tail -2 $OUT/p_pinned.log
( SOAK_VT=200,6000 SOAK_TIMEOUT=200 timeout 260 python tools/soak_par.py p_noise_v200 120000 124000 16 noise ) > $OUT/p_noise.log 2>&1; echo "== noise family, 200 voices rc=$?"; tail -8 $OUT/p_noise.log | cut -c1-230
( FUZZ_MORE_OV=1 SOAK_VT=200,6000 SOAK_TIMEOUT=200 timeout 260 python tools/soak_par.py p_noise_more_ov_v200 124000 128000 16 noise ) > $OUT/p_noise2.log 2>&1; echo "== noise family, per-voice parameters rc=$?"; tail -8 $OUT/p_noise2.log | cut -c1-230
( SOAK_VT=16,48000 SOAK_TIMEOUT=200 timeout 260 python tools/soak_par.py p_noise_1s 128000 129500 16 noise ) > $OUT/p_noise3.log 2>&1; echo "== noise family, 1 s rc=$?"; tail -8 $OUT/p_noise3.log | cut -c1-230
