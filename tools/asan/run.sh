#!/bin/bash
# Host code (graph, planner, flattener, kernel generator, .srk reader / writer, C ABI) under AddressSanitizer + UBSan, no GPU needed:
# render.hip is replaced by a stub that flattens and has no device.  Run from the repo root:  bash tools/asan/run.sh
# (swaps s-rack_amd/libsrack_hip.so for the instrumented build while it runs and puts the real one back)
set -u
ROOT=$(pwd); OUT=/tmp/srack_asan; mkdir -p $OUT
( cd s-rack_amd/csrc && g++ -std=c++17 -O1 -g -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer -I. -Ibuild -I$ROOT/tools/asan -I/opt/rocm/include \
    -D__HIP_PLATFORM_AMD__ -include cstring -shared -o $OUT/libsrack_hip.so graph.cpp flatten.cpp srk.cpp capi.cpp dist.cpp jit.cpp $ROOT/tools/asan/stub.cpp \
    -ldl -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib ) || exit 1
cp s-rack_amd/libsrack_hip.so $OUT/real.so && cp $OUT/libsrack_hip.so s-rack_amd/libsrack_hip.so
PRE=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$PRE python tools/asan/drive.py 2>&1 | grep -v "^  0x\|^=>" | head -40
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$PRE python -m pytest tests/test_srk.py tests/test_host.py -q -m "not gpu" -p no:cacheprovider \
    --deselect tests/test_host.py::test_no_gpu_render_fails_loudly 2>&1 | tail -3
cp $OUT/real.so s-rack_amd/libsrack_hip.so
