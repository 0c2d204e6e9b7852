"""glibc's __powf_log2_data (sysdeps/ieee754/flt-32/e_powf_log2_data.c: 16 x {1/c, log2 c} and the degree-5 polynomial of log2(1 + r) / r) as the
HOST's libm holds it, for s-rack_amd/csrc/modules.hip.h (kPowfLog2Tab, powf_libm): found in libm.so.6 by the polynomial's known coefficients, printed
as the bit patterns of its doubles.  Run once per glibc; tests/test_oracle.py holds the device header's copy to the host's powf through a Python
transliteration (tests/libm_powf.py)."""
import glob, struct, sys
path = (glob.glob("/lib/x86_64-linux-gnu/libm.so.6") + glob.glob("/usr/lib/x86_64-linux-gnu/libm.so.6") + glob.glob("/lib64/libm.so.6"))[0]
data = open(path, "rb").read()
a4 = struct.pack("<d", float.fromhex("0x1.71547652ab82bp0"))   # POWF_SCALE_BITS = 0: the last coefficient is log2(e)'s neighbour
at = data.find(a4)
assert at >= 0 and data.find(a4, at + 1) < 0, "the polynomial's last coefficient, once"
poly0 = at - 4 * 8
tab0 = poly0 - 16 * 16
tab = struct.unpack("<32Q", data[tab0:tab0 + 256])
poly = struct.unpack("<5d", data[poly0:poly0 + 40])
assert struct.unpack("<d", struct.pack("<Q", tab[2 * 9]))[0] == 1.0 and tab[2 * 9 + 1] == 0, "entry 9 is c = 1: {1.0, 0.0}"
print("// " + path)
for i in range(16):
    print("    {0x%016x, 0x%016x},  // 1 / c = %s, log2 c = %s" % (tab[2 * i], tab[2 * i + 1], struct.unpack("<d", struct.pack("<Q", tab[2 * i]))[0].hex(), struct.unpack("<d", struct.pack("<Q", tab[2 * i + 1]))[0].hex()))
print("poly:", ", ".join(p.hex() for p in poly))
