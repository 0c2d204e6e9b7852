#!/bin/bash
# Round 5, pass I: the suite, config 4's curves and the bench line after the exact oscillator's last changes (pow's table in LDS, the rounding check below a power of two)
set -u
OUT=gpurun_out/r5; mkdir -p $OUT gpurun_out/profiles
bash tools/gpu_r5_d.sh i
( HORIZON_WORKLOADS=cfg4,cfg4_b1024 HORIZON_FLAGS=0,32,34,1 timeout 600 python tools/horizon.py $OUT/horizon_cfg4_i.json ) > $OUT/i_horizon.log 2>&1; echo "== horizon rc=$?"; cut -c1-200 $OUT/i_horizon.log
