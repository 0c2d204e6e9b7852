// powcheck.hip — how often does a device 2^e differ from the host libm's pow(2, e)?  (exact-mode oscillators with a CV compute
// 440 * 2^e / sr per sample on the device; the oracle uses the host libm, as the reference does.)
//   ocml pow:  ~19 % of the results differ in the last bit(s)
//   exp2_cr:   double-double evaluation, correctly rounded => differs only where the host libm itself misrounds
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../s-rack_amd/csrc/modules.hip.h"
__global__ void k(const double* e, double* out, double* out2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = 440.0 * pow(2.0, e[i]) / 48000.0;
        out2[i] = 440.0 * srack::dev::exp2_cr(e[i]) / 48000.0;
    }
}
__global__ void klibm(const double* e, double* out, int n)   // the port of the host libm's own algorithm (exact render mode since round 4)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = srack::dev::exp2_libm(e[i]);
}
__global__ void kraw(const double* e, double* out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = srack::dev::exp2_cr(e[i]);
}
// the Ziv evaluation (exp2_cr) against the reference evaluation it falls back on, bit for bit, over counter-generated arguments:
// any difference would be an error bound that does not hold
__global__ void kziv(unsigned long long first, unsigned long long* differ)
{
    unsigned long long x = first + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x;
    x += 0x9e3779b97f4a7c15ull; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull; x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; x ^= x >> 31;  // splitmix64
    // half the arguments are sums of two f32 (what an oscillator sees), half are full doubles in [-12, 12)
    const double u = (double)(x >> 11) * 0x1p-53;
    double e = u * 24.0 - 12.0;
    if (x & 1) e = (double)(float)(u * 12.0 - 6.0) + (double)(float)((double)((x >> 3) & 0xfffff) * 0x1p-20 * 9.0 - 6.0);
    const double n = __builtin_rint(e), f = e - n;
    const double ref = __builtin_ldexp(srack::dev::exp2_cr_taylor(f), (int)n), got = srack::dev::exp2_cr(e);
    if (__double_as_longlong(ref) != __double_as_longlong(got)) atomicAdd(differ, 1ull);
}
// (the host compiler folds pow(2.0, x) into exp2(x) when it sees the constant base — LLVM's libcall simplifier, fast-math or not —: the
// comparisons below are against the libm's POW, which is what the oracle (gcc) and an unoptimised build of the reference call)
static double host_pow2(double e)
{
    volatile double two = 2.0;
    return std::pow(two, e);
}
int main()
{
    {
        unsigned long long *d, h[2] = {0, 0};
        hipMalloc(&d, 16); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
        const unsigned long long per = 1ull << 28;
        for (int r = 0; r < 4; r++) hipLaunchKernelGGL(kziv, dim3((unsigned)(per / 256)), dim3(256), 0, 0, per * r, d);
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("exp2_cr (Ziv) against its double-double reference evaluation over %llu arguments: %llu differ\n", 4 * per, h[0]);
    }
    const int n = 1 << 22;
    std::vector<double> e(n), got(n), got2(n);
    uint64_t s = 12345;
    for (int i = 0; i < n; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const float cv = (float)((double)(s >> 11) / 9007199254740992.0 * 12.0 - 6.0);  // an f32 CV
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const float val = (float)((double)(s >> 11) / 9007199254740992.0 * 9.0 - 6.0);  // an f32 `val`
        e[i] = (double)cv + (double)val;
    }
    double *de, *dout, *dout2;
    hipMalloc(&de, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&dout2, n * 8);
    hipMemcpy(de, e.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, de, dout, dout2, n);
    hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(got2.data(), dout2, n * 8, hipMemcpyDeviceToHost);
    int bad = 0, bad2 = 0, badp = 0, bade = 0, cr_vs_l = 0, libm_vs_l = 0;
    std::vector<double> raw(n);
    {   // the bare power, against the x87 long-double exp2l rounded to double (11 extra bits: decides all but ~0.05 % of the roundings)
        double* dr; hipMalloc(&dr, n * 8);
        hipLaunchKernelGGL(kraw, dim3(n / 256), dim3(256), 0, 0, de, dr, n);
        hipMemcpy(raw.data(), dr, n * 8, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; i++) {
            const double l = (double)exp2l((long double)e[i]), m = host_pow2(e[i]);
            if (memcmp(&l, &raw[i], 8)) cr_vs_l++;
            if (memcmp(&l, &m, 8)) libm_vs_l++;
        }
    }
    for (int i = 0; i < n; i++) {
        const double want = 440.0 * host_pow2(e[i]) / 48000.0, want2 = 440.0 * std::exp2(e[i]) / 48000.0;
        if (memcmp(&want, &got[i], 8)) bad++;
        if (memcmp(&want, &got2[i], 8)) bad2++;
        if (memcmp(&want2, &got2[i], 8)) bade++;
        if (memcmp(&want, &want2, 8)) badp++;
    }
    {   // exp2_libm against the host's pow, bit for bit: the oscillator arguments above, then wide and tiny ones (every branch of the port)
        double* dr; hipMalloc(&dr, n * 8);
        std::vector<double> lm(n);
        int differ = 0;
        for (int pass = 0; pass < 4; pass++) {
            std::vector<double> arg(e);
            if (pass == 1) for (int i = 0; i < n; i++) arg[i] = e[i] * 50.0;              // up to +- 600: every scale the table reaches, the scaled-arithmetic range beyond 512 / ln 2
            if (pass == 2) for (int i = 0; i < n; i++) arg[i] = std::ldexp(e[i], -40 - (i & 63));  // tiny: the 1 + x exits
            if (pass == 3) for (int i = 0; i < n; i++) arg[i] = e[i] * 200.0;             // overflow / underflow / subnormal results
            hipMemcpy(de, arg.data(), n * 8, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(klibm, dim3(n / 256), dim3(256), 0, 0, de, dr, n);
            hipMemcpy(lm.data(), dr, n * 8, hipMemcpyDeviceToHost);
            int d = 0;
            for (int i = 0; i < n; i++) {
                const double m = host_pow2(arg[i]);
                if (memcmp(&m, &lm[i], 8)) { if (d < 3) printf("   pass %d: e = %a  libm %a  device %a\n", pass, arg[i], m, lm[i]); d++; }
            }
            printf("exp2_libm != host pow(2, e), pass %d (%s): %d of %d\n", pass, pass == 0 ? "oscillator arguments" : pass == 1 ? "x 50" : pass == 2 ? "tiny" : "x 200", d, n);
            differ += d;
        }
        hipMemcpy(de, e.data(), n * 8, hipMemcpyHostToDevice);
        printf("exp2_libm against the host libm: %d differ in all\n", differ);
    }
    printf("bare 2^e against RN(exp2l): exp2_cr differs in %d, libm pow in %d\n", cr_vs_l, libm_vs_l);
    printf("of %d: ocml pow != libm pow %d;  exp2_cr != libm pow %d;  exp2_cr != libm exp2 %d;  libm pow != libm exp2 %d\n", n, bad, bad2, bade, badp);
    return 0;
}
