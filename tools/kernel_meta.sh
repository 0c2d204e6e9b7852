#!/bin/bash
# tools/kernel_meta.sh — registers, LDS and scratch of every kernel in s-rack_amd/libsrack_hip.so (code-object metadata): what bounds the
# occupancy.  One-wave workgroups: waves per SIMD = min(512 / VGPRs (8 at most), 160 KB / LDS per workgroup / 4).
set -eu
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin s-rack_amd/libsrack_hip.so "$T/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$T/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/lib.co"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$T/lib.co" > "$T/notes.txt"
python3 - "$T/notes.txt" <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
rows = []
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    rows.append((name.replace("srack::", "")[:90], int(g("vgpr_count")), int(blk.split()[0]), int(g("sgpr_count")), int(g("group_segment_fixed_size")), int(g("private_segment_fixed_size")), int(g("max_flat_workgroup_size"))))
print("%-90s %5s %5s %5s %7s %8s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "lds", "scratch", "wg"))
for r in sorted(rows):
    print("%-90s %5d %5d %5d %7d %8d %5d" % r)
PY
rm -rf "$T"
