#!/bin/bash
# tools/disasm.sh [out.s] — the gfx950 code object inside s-rack_amd/libsrack_hip.so, disassembled (ISA counts quoted in DESIGN.md).
# tools/disasm.sh out.s '<mangled-name regex>' additionally prints an instruction histogram per loop of that kernel.
set -eu
OUT=${1:-/tmp/srack_lib.s}
T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin s-rack_amd/libsrack_hip.so "$T/fat.bin"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input="$T/fat.bin" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/lib.co"
/opt/rocm/lib/llvm/bin/llvm-objdump -d "$T/lib.co" > "$OUT"
rm -rf "$T"
[ $# -ge 2 ] || exit 0
python3 - "$OUT" "$2" <<'PY'
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*>:$", l) and re.search(sys.argv[2], l))
end = next(i for i in range(start + 1, len(lines)) if lines[i] == "")
body = lines[start + 1:end]
print(lines[start], len(body), "instructions")
addr = lambda l: int(l.split("//")[1].split(":")[0], 16)   # llvm-objdump: "\tinsn ... // 000000330800: ENCODING"
a2i = {addr(l): i for i, l in enumerate(body) if "//" in l}
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+\d+\s.*<.*\+0x([0-9a-f]+)>", l)
    if not m: continue
    base = int(lines[start].split()[0], 16)
    j = a2i.get(base + int(m.group(1), 16))
    if j is None or j >= i: continue   # forward branch
    ops = collections.Counter(x.split()[0] for x in body[j:i + 1])
    print(f"loop [{j}, {i}] {i - j + 1} instructions: " + " ".join(f"{k}:{v}" for k, v in ops.most_common()))
PY
