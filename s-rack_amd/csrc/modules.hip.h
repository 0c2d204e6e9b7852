// modules.hip.h — one per-sample step function per module type, for one voice (= one lane).
//
// These are the arithmetic bodies of the reference's calc() loops (file:line cited per function),
// written once and inlined both into the tile interpreter (wires in LDS tiles) and into the fused
// chain kernels (wires in VGPRs).  `flags` is wave-uniform; in the fused kernels it is a
// compile-time constant and every `if (flags & ...)` folds away.
//
// Numerics contract (build: -ffp-contract=off, f32 denormals on, correctly rounded f32 divide):
//   * ADSR / VCA / mixer / math / sequencers: the same IEEE f32 operations in the same order as the
//     reference => bit-identical to the CPU tick given identical inputs.  The ladder filter likewise in
//     the exact mode; in the default mode its multiply-subtract pairs are fma-contracted and its clamps
//     are v_med3 (vcf_step<kFast>): ~1e-7 relative, inside the 1e-5 contract.
//   * oscillator phase: f64 accumulate + exact wrap => bit-identical `pos` whenever delta is
//     (constant pitch: delta is computed on the host with glibc pow, like the reference).
//   * oscillator outputs, default mode: PolyBLEP evaluated in f32 from f64-exact phase differences (abs
//     error ~1e-7, inside the 1e-5 contract); the sine in f64 with one rounding to f32 (sine_fast); OSC_EXACT mode: f64 with true
//     division / ocml sin / pow exactly as oscillator.rs spells them (saw and square bit-identical).
#pragma once
#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#endif

#ifdef __HIPCC_RTC__
#include "srack_hip.h"
#else
#include "../../include/srack_hip.h"
#endif
#include "program.hpp"

namespace srack {
namespace dev {

#define SRK_DEV __device__ __forceinline__

// Makes x opaque to the optimiser at this point: both arms of a following select are then already
// computed, so it stays a v_cndmask instead of being turned into an exec-masked branch.
SRK_DEV float keep(float x)
{
    asm("" : "+v"(x));
    return x;
}

// The 0.0 an unconnected input reads (`None => 0.0` in every calc()).  Where a kernel is specialised for a program the port
// flags are compile-time constants and that zero becomes a literal — and the AMDGPU backend folds `(0.0 - a) - b` into
// `(-a) - b`, which turns +0.0 into -0.0 when a and b are both +0.0 (seen on a filter without audio input: a channel of -0.0
// where the reference has +0.0).  Those kernels define SRK_OPAQUE_ZERO: the zero then comes out of an empty asm statement the
// optimiser cannot see through, and the arithmetic stays the reference's.
SRK_DEV float zero_f32()
{
#ifdef SRK_OPAQUE_ZERO
    float z = 0.0f;
    asm("" : "+v"(z));
    return z;
#else
    return 0.0f;
#endif
}

// TransitionDetector::is_transition, synth.rs:292-297
constexpr int kTileRows = 32;  // samples per tile of the tile-wise forms below (wave.hip.h: kMixRows, checked there)

SRK_DEV bool rising_edge(bool& last, float val)
{
    bool above = val > 0.0f;
    bool t = above && !last;
    last = above;
    return t;
}

// ---------------------------------------------------------------------------------------------
// Oscillator — oscillator.rs:43-67 (get_freq_in_hz, poly_blep), :124-153 (per-sample body)
// ---------------------------------------------------------------------------------------------
struct OscRegs {
    double pos;      // phase in [0,1)
    bool sync_last;  // sync TransitionDetector.last
    // CV path only, not module state: the last CV seen and the increment computed for it.  2^x is a pure function
    // of (cv, val), and sequencer-driven CVs are constant for thousands of samples, so it is re-evaluated only
    // when some lane's CV changed (NaN != NaN: the first sample always evaluates).
    float seen_cv = __builtin_nanf("");
    double seen_delta = 0.0;
};

struct OscConst {    // per-voice constants, set up once per kernel (or tile)
    double delta;    // phase increment when CV is unconnected
    double val;      // f64(val) when CV is connected
    double sr;       // f64(sample_rate)
    float inv_dt;    // 1 / f32(delta) for the f32 PolyBLEP (constant pitch)
    double scale = 0.0;  // 440 / sr * 2^val, only where a kernel passes OSC_VAL_FOLDED
};

// 1 / f32(dt) for the f32 PolyBLEP of a CONSTANT pitch, rounded ONCE: the f64 quotient, then one conversion (set-up time: once per voice and
// launch).  `1.0f / (float)delta` rounds twice, and the carried-phase forms (cosc_saw, fosc_saw: t' + 1 = next phase / dt) square the quotient —
// measured on the GPU in round 6 (tools/emu_vs_gpu.py): a raw constant-pitch saw or square was up to 1.0e-6 off the reference where the
// error bound (csrc/approx.cpp, kEpsBlep) had counted 2.4e-7.
SRK_DEV float inv_dt_f32(double delta) { return (float)(1.0 / delta); }

// 2^x in f64 for the CV path of the default mode: x = n + f, |f| <= 1/2, a degree-8 polynomial for 2^f (Chebyshev
// interpolant: max relative error 1.1e-12 on the interval), scaled by 2^n with ldexp.  The phase increment needs ~1e-10
// relative accuracy to keep the accumulated phase error of a 1 s render below 1e-7 cycles; this leaves two orders of
// magnitude.  Evaluated by Estrin's scheme — 8 fma + 3 mul with a dependency depth of 4 instead of Horner's 10: the FM
// kernels run ONE wave per SIMD, where the length of the per-sample dependency chain, not the instruction count, sets the
// pace (8 fma + 2 mul since the degree-8 coefficient joined the f^4 group).  Overflow / NaN propagate through ldexp like pow's.
// (Measured and dropped: the libm-style form — a 32-entry table of 2^(k/32) from constant memory plus a degree-4 polynomial, 9
// f64-rate instructions instead of 15.  The table load sits inside the modulator's feedback recurrence and its latency cannot be
// hidden at one wave per SIMD: config 4 went from 9.4 to 16.5 ms per step.)
template <bool kReduce = true>
SRK_DEV double exp2_fast(double x)
{
    // kReduce == false: the caller has proved |x| <= 1/2 (OSC_CV_SMALL), x is its own reduced argument: no rndne / subtract /
    // cvt_i32 / ldexp, four of the 19 f64-rate instructions of an oscillator's increment.
    const double n = kReduce ? __builtin_rint(x) : 0.0;
    const double f = kReduce ? x - n : x;
    const double f2 = f * f;
    const double a01 = __builtin_fma(0.6931471805459332, f, 1.0000000000000004);
    const double a23 = __builtin_fma(0.055504109412108156, f, 0.24022650695814518);
    const double a45 = __builtin_fma(0.0013333450563173552, f, 0.009618129159053034);
    const double a67 = __builtin_fma(1.5310080611926545e-05, f, 0.00015403456082633648);
    const double f4 = f2 * f2;
    const double b0 = __builtin_fma(a23, f2, a01);
    const double b1 = __builtin_fma(a67, f2, a45);
    const double b2 = __builtin_fma(1.3255179556479267e-06, f4, b1);  // the degree-8 term rides on f4: no f^8, same depth of 4
    const double p = __builtin_fma(b2, f4, b0);
    return kReduce ? __builtin_ldexp(p, (int)n) : p;
}

// The same with the interpolant of degree 10 (tools/exp2_coeffs.py: 3.1e-16 with the coefficients rounded to f64; 10 fma + 3 mul, depth 5)
// — for every oscillator OUTSIDE the proved FM loops (osc_delta_fast).  1.1e-12 is three orders below what ONE oscillator's phase needs,
// but it is a smooth function of the CV, not noise: a CV that is flagged as sweeping and in fact sits still (a filter that renders
// silence, a square between its edges) makes the phase drift one way, 1e-9 cycles in a second; the f32 roundings of the saw it produces
// then flip one way, and an oscillator that takes that saw as its pitch integrates the flips — 4e-5 on ITS saw after a second (the
// one-second soak's seeds 30111, 31051).  At 3e-16 the chain ends at 1e-8.  The proved loops keep degree 8: their CVs are sines through
// gains, which do sweep.
template <bool kReduce = true>
SRK_DEV double exp2_fast10(double x)
{
    const double n = kReduce ? __builtin_rint(x) : 0.0;
    const double f = kReduce ? x - n : x;
    const double f2 = f * f;
    const double a01 = __builtin_fma(0x1.62e42fefa3a19p-1, f, 1.0);
    const double a23 = __builtin_fma(0x1.c6b08d703ce49p-5, f, 0x1.ebfbdff82c598p-3);
    const double a45 = __builtin_fma(0x1.5d87fe9d7a584p-10, f, 0x1.3b2ab6fba1ddap-7);
    const double a67 = __builtin_fma(0x1.ffcb54062e698p-17, f, 0x1.430913096fd9fp-13);
    const double a89 = __builtin_fma(0x1.b675bca4eeebbp-24, f, 0x1.62bfd47773353p-20);
    const double f4 = f2 * f2;
    const double b0 = __builtin_fma(a23, f2, a01);
    const double b1 = __builtin_fma(a67, f2, a45);
    const double b2 = __builtin_fma(0x1.e6063f7217bc6p-28, f2, a89);
    const double f8 = f4 * f4;
    const double lo = __builtin_fma(b1, f4, b0);
    const double p = __builtin_fma(b2, f8, lo);
    return kReduce ? __builtin_ldexp(p, (int)n) : p;
}

// Between the two, degree 9 (tools/exp2_coeffs.py 9: 1.9e-14; 9 fma + 2 mul, depth 4 like degree 8) — for the PROVED classes (OSC_CV_SERIES9:
// render_fm_pair_x / render_fm_pair_block_x's carriers, the specialised kernels' bounded-CV oscillators).  Their CVs are sines through gains
// and do sweep, but a sine does not average a smooth error away: the degree-8 error, weighted by 2^cv, leaves 1.3e-13 relative — 1.2e-8 cycles
// of phase per minute at config 4's carrier, and the time-parallel kernel's error crept from 2.4e-7 to 3.6e-7 over a minute
// (profiles/r06_horizon.json; the same sum on the CPU: -1.2e-8 / +1.2e-8 / -1.0e-8 cycles for index 0.5 / 1 / 1.5).  Degree 9 leaves 1e-10
// cycles per minute for one instruction more (degree 10: 3e-12 for three: config 4 at buffer_size 1024 13.75 -> 14.35 ms, measured).
template <bool kReduce = true>
SRK_DEV double exp2_fast9(double x)
{
    const double n = kReduce ? __builtin_rint(x) : 0.0;
    const double f = kReduce ? x - n : x;
    const double f2 = f * f;
    const double a01 = __builtin_fma(0x1.62e42fefa39f7p-1, f, 0x1.000000000003dp+0);
    const double a23 = __builtin_fma(0x1.c6b08d7044119p-5, f, 0x1.ebfbdff8149f2p-3);
    const double a45 = __builtin_fma(0x1.5d87fe908f88ap-10, f, 0x1.3b2ab72b175eep-7);
    const double a67 = __builtin_fma(0x1.ffcb76789860fp-17, f, 0x1.43088e257f341p-13);
    const double a89 = __builtin_fma(0x1.b6571de2f2351p-24, f, 0x1.63ef969a64d3cp-20);
    const double f4 = f2 * f2;
    const double b0 = __builtin_fma(a23, f2, a01);
    const double b1 = __builtin_fma(a67, f2, a45);
    const double b2 = __builtin_fma(a89, f4, b1);
    const double p = __builtin_fma(b2, f4, b0);
    return kReduce ? __builtin_ldexp(p, (int)n) : p;
}

// (SRK_OSC_NOCOLD, below: a TIMING experiment only — tools/ab_survey_opts.sh passes it to the kernel generator's compiles through SRACK_JIT_OPTS.
// It leaves the out-of-line repairs of the exact forms out — the results then differ from the reference's in a few samples per million — to
// measure what the per-sample "could some lane not decide" branches cost the general path: notes/r06.md R6.17.)
// 2^e, correctly rounded (exact render mode).  The reference evaluates `2.0_f64.powf(e)` with the host's libm, whose pow is
// within 0.52 ulp of the true value, i.e. the correctly rounded double except for a fraction of a percent of the arguments.
// ocml's pow / exp2 are ~1 ulp functions: 19 % of their results differ from the host's in the last bit (tools/powcheck.hip),
// and although an oscillator's f32 output shows that only rarely, a patch that feeds it back (chaotic FM loops, a clock
// derived from the saw) then parts from the oracle for good.  So the power is evaluated here in double-double arithmetic
// and rounded once: e = n + f, |f| <= 1/2; y = f ln2 (ln2 as hi + lo); the reference evaluation is exp(y) = (exp(y / 8))^8 with
// a degree-13 Taylor polynomial, every operation double-double (|y / 8| < 0.0434: truncation 2^-99, ~2^-100 overall) — and an
// evaluation five times cheaper decides the rounding whenever it provably can (exp2_cr below).  Finite results only; the rest
// goes to ocml's pow.
struct DD {
    double hi, lo;
};
SRK_DEV DD dd_fast_two_sum(double a, double b)  // |a| >= |b|
{
    const double s = a + b;
    return DD{s, b - (s - a)};
}
SRK_DEV DD dd_two_sum(double a, double b)
{
    const double s = a + b, bb = s - a;
    return DD{s, (a - (s - bb)) + (b - bb)};
}
SRK_DEV DD dd_mul(DD a, DD b)
{
    const double p = a.hi * b.hi;
    const double e = __builtin_fma(a.hi, b.hi, -p) + (a.hi * b.lo + a.lo * b.hi);
    return dd_fast_two_sum(p, e);
}
SRK_DEV DD dd_add(DD a, DD b)
{
    DD s = dd_two_sum(a.hi, b.hi);
    s.lo += a.lo + b.lo;
    return dd_fast_two_sum(s.hi, s.lo);
}
SRK_DEV DD dd_mul_d(DD a, double b)
{
    const double p = a.hi * b;
    const double e = __builtin_fma(a.hi, b, -p) + a.lo * b;
    return dd_fast_two_sum(p, e);
}
SRK_DEV DD dd_sqr(DD a)
{
    const double p = a.hi * a.hi;
    const double e = __builtin_fma(a.hi + a.hi, a.lo, __builtin_fma(a.hi, a.hi, -p));
    return dd_fast_two_sum(p, e);
}
// 2^f for |f| <= 1/2 as hi + lo, ~2^-100: the reference evaluation every result can fall back on (410 f64-rate instructions)
SRK_DEV double exp2_cr_taylor(double f)
{
    // y = f * ln2 / 8
    const DD ln2_8 = {0x1.62e42fefa39efp-4, 0x1.abc9e3b39803fp-59};  // ln 2 / 8 = hi + lo
    const DD y = dd_mul_d(ln2_8, f);
    // exp(y) = 1 + y (1 + y/2 (1 + y/3 ( ... (1 + y/13))))
    DD acc = {1.0, 0.0};
#pragma unroll
    for (int k = 13; k >= 1; k--) {
        DD t = dd_mul(acc, y);
        // t / k: k is a small integer; 1/k as a double-double
        const double rk = 1.0 / (double)k;
        const double rk_lo = __builtin_fma(-rk, (double)k, 1.0) / (double)k;  // 1/k = rk + rk_lo
        t = dd_mul(t, DD{rk, rk_lo});
        acc = dd_add(DD{1.0, 0.0}, t);
    }
    acc = dd_mul(acc, acc);
    acc = dd_mul(acc, acc);
    acc = dd_mul(acc, acc);
    return acc.hi + acc.lo;
}
// Ziv's strategy: a cheap evaluation with a proved error bound decides the rounding whenever the result is not within that bound of
// a rounding boundary, which is all but 2^-13.5 of the arguments; only a wave with such a lane (one in 180) pays for the one above.
// Cheap = exp(y)^64 with y = f ln2 / 64, |y| <= 0.0055:  1 + y + y^2/2 in double-double, y^3 (1/6 + ... + y^5/8!) in plain f64 — that
// part is below 2^-25, so its five roundings (2^-50.7 relative) and the final sum's cost 2^-76 of the result; truncation (y^9/9!) is
// 2^-86 — then six double-double squarings, each of which doubles the relative error: 2^-69.5 at the end.  The test uses 2^-67.
SRK_DEV double exp2_cr(double e)
{
    if (!(e > -1000.0 && e < 1000.0)) return pow(2.0, e);  // overflow / gradual underflow / NaN: the library's special cases
    const double n = __builtin_rint(e);
    const double f = e - n;  // exact
    const DD ln2_64 = {0x1.62e42fefa39efp-7, 0x1.abc9e3b39803fp-62};  // ln 2 / 64 = hi + lo
    const DD y = dd_mul_d(ln2_64, f);
    const double yh = y.hi;
    double q = 1.0 / 40320.0;
    q = __builtin_fma(q, yh, 1.0 / 5040.0);
    q = __builtin_fma(q, yh, 1.0 / 720.0);
    q = __builtin_fma(q, yh, 1.0 / 120.0);
    q = __builtin_fma(q, yh, 1.0 / 24.0);
    q = __builtin_fma(q, yh, 1.0 / 6.0);
    const double y2 = yh * yh;
    const double c3 = (y2 * yh) * q;
    const double y2l = __builtin_fma(yh + yh, y.lo, __builtin_fma(yh, yh, -y2));  // y^2 = y2 + y2l (y.lo^2 dropped: 2^-120)
    const DD t = dd_fast_two_sum(1.0, yh);
    const DD v = dd_fast_two_sum(t.hi, 0.5 * y2);
    const double rest = ((t.lo + v.lo) + (y.lo + 0.5 * y2l)) + c3;
    DD a = dd_fast_two_sum(v.hi, rest);
#pragma unroll
    for (int k = 0; k < 6; k++) a = dd_sqr(a);
    const double d = a.hi * 0x1p-67;
    const double up = a.hi + (a.lo + d), down = a.hi + (a.lo - d);
    double r = up;
    if (__builtin_amdgcn_ballot_w64(up != down) != 0) {
        if (up != down) r = exp2_cr_taylor(f);
    }
    return __builtin_ldexp(r, (int)n);
}

// 2^e as the HOST's libm computes `pow(2.0, e)` — what the reference's `2.0_f64.powf(e)` is (oscillator.rs:45; Rust's f64::powf is the
// platform libm's pow).  glibc's pow (2.28 and later: Szabolcs Nagy's algorithm, sysdeps/ieee754/dbl-64/e_pow.c) is within 0.52 ulp of the
// true value: NOT always the correctly rounded double — one argument in 1300 comes out one ulp off (tools/pow_misround.py), and a patch
// that iterates its phases (a loop through a pitch or sync input) grows that last bit into different samples.  Until round 3 the exact
// mode evaluated the power correctly rounded (exp2_cr below, double-double): bit-identical to the reference except on those arguments,
// and so not on the fuzzer's seeds 725 and 1473.  This is the algorithm itself, operation for operation as the x86-64 FMA build of glibc
// 2.35 executes it (disassembled: which products are contracted into fused multiply-adds is the compiler's choice, and part of the
// result): log(2) = lhi + llo falls out of its log step as constants (x = 2: table entry 75, r = 0), then
//     ehi = e * lhi,  elo = fma(e, llo, fma(lhi, e, -ehi)),  kd = fma(ehi, N / ln 2, 0x1.8p52) - 0x1.8p52  (N = 128),
//     r = fma(kd, -ln2lo / N, fma(kd, -ln2hi / N, ehi)) + elo,  2^(k / N) = (1 + tail) * scale from its 128-entry table,
//     result = fma(tail + r + r^2 (C2 + r C3) + r^4 (C4 + r C5), scale, scale)   (the sums as the fmas written below).
// tools/powcheck.hip compares it with the host's pow over 2^22 oscillator arguments on the GPU box (0 differ), tests/test_oracle.py runs
// a Python transliteration (exact fused multiply-adds) against the host's pow.  |e ln 2| >= 512 (a pitch of 2^738 cycles per sample)
// and NaNs go to ocml's pow: inf, 0 and NaN are the same there, and nothing in between is a pitch.
__device__ const uint64_t kLibmExpTab[256] = {  // glibc's __exp_data.tab (N = 128): {bits(tail_k), bits(2^(k/N)) - (k << 52) / N}
    0x0000000000000000ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull,
    0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull, 0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull,
    0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
    0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull,
    0x3c979aa65d837b6dull, 0x3fefb5586cf9890full, 0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull,
    0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
    0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull,
    0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull, 0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull,
    0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
    0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull,
    0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull, 0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull,
    0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
    0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full,
    0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull, 0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull,
    0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
    0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull,
    0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull, 0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull,
    0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
    0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull,
    0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull, 0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull,
    0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
    0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull,
    0x3c4363ed60c2ac11ull, 0x3feece086061892dull, 0x3c9666093b0664efull, 0x3feeca41ed1d0057ull,
    0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
    0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull,
    0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull, 0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull,
    0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
    0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull,
    0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull, 0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull,
    0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
    0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull,
    0xbc93cedd78565858ull, 0x3feea23882552225ull, 0x3c5710aa807e1964ull, 0x3feea155d44ca973ull,
    0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
    0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull,
    0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull, 0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull,
    0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
    0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull,
    0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull, 0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull,
    0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
    0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull,
    0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull, 0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull,
    0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
    0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full,
    0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull, 0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull,
    0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
    0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull,
    0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull, 0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull,
    0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
    0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull,
    0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull, 0xbc90a40e3da6f640ull, 0x3feef9728de5593aull,
    0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
    0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull,
    0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull, 0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull,
    0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
    0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull,
    0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull, 0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull,
    0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
    0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull,
    0xbc82d52107b43e1full, 0x3fef7c97337b9b5full, 0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull,
    0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
    0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull,
    0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull, 0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull,
    0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
    0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull,
};

// A kernel specialised at run time for a program with an exact oscillator whose pitch moves keeps the table in LDS (jit.cpp defines
// SRK_LIBM_TAB_LDS and fills it at kernel entry): the lookup sits inside the oscillator's recurrence, at one wave per SIMD where nothing hides
// a trip to the vector cache.
#if defined(SRK_LIBM_TAB_LDS)
static __shared__ uint64_t srk_libm_tab_lds[256];
#define SRK_LIBM_TAB(i) srk_libm_tab_lds[(i)]
SRK_DEV void libm_tab_fill(int lane)
{
    for (int k = lane; k < 256; k += 64) srk_libm_tab_lds[k] = kLibmExpTab[k];
    __syncthreads();
}
#else
#define SRK_LIBM_TAB(i) kLibmExpTab[(i)]
#endif

// (the library's pow and fmod behind the cold branches below: out of line — inlined, each of their copies per oscillator was a few hundred
// instructions and several dozen registers the hot path never uses)
__device__ __attribute__((noinline)) double pow2_cold(double e) { return pow(2.0, e); }
__device__ __attribute__((noinline)) double fmod1_cold(double x) { return fmod(x, 1.0); }

// (the arithmetic of the plain range, branch-free; `cold` is set where the argument is outside it and the value returned is not pow's;
// `tab(i)`: word i of the table above, from wherever the caller keeps it)
template <class Tab>
SRK_DEV double exp2_libm_plain_t(double e, bool& cold, const Tab& tab)
{
    constexpr double lhi = 0x1.62e42fefa39efp-1, llo = 0x1.abc9e3b398000p-56;  // log(2.0) as glibc's log_inline returns it
    const double ehi = e * lhi;
    const double elo = __builtin_fma(e, llo, __builtin_fma(lhi, e, -ehi));
    const uint32_t abstop = ((uint32_t)__double2hiint(ehi) >> 20) & 0x7ffu;
    cold = cold || !(abstop - 0x3c9u <= 0x3eu);  // plain: 2^-54 <= |e ln 2| < 512
    const double kds = __builtin_fma(ehi, 0x1.71547652b82fep+7, 0x1.8p52);
    const uint64_t ki = (uint64_t)__double_as_longlong(kds);
    const double kd = kds - 0x1.8p52;
    double r = __builtin_fma(kd, -0x1.cf79abc9e3b3ap-47, __builtin_fma(kd, -0x1.62e42fefa0000p-8, ehi));
    r = elo + r;
    const uint32_t idx = 2u * ((uint32_t)ki & 127u);
    const double tail = __longlong_as_double((long long)tab(idx));
    const uint64_t sbits = tab(idx + 1u) + (ki << 45);
    const double r2 = r * r;
    const double a = __builtin_fma(r, 0x1.555555555543cp-3, 0x1.ffffffffffdbdp-2);
    const double b = r + tail;
    const double c = __builtin_fma(r, 0x1.1111167a4d017p-7, 0x1.55555cf172b91p-5);
    double tmp = __builtin_fma(a, r2, b);
    tmp = __builtin_fma(c, r2 * r2, tmp);
    const double scale = __longlong_as_double((long long)sbits);
    return __builtin_fma(tmp, scale, scale);
}
struct LibmTabDefault {
    SRK_DEV uint64_t operator()(uint32_t i) const { return SRK_LIBM_TAB(i); }
};
SRK_DEV double exp2_libm_plain(double e, bool& cold) { return exp2_libm_plain_t(e, cold, LibmTabDefault{}); }
// (what pow does outside the plain range)
SRK_DEV double exp2_libm_special(double e)
{
    constexpr double lhi = 0x1.62e42fefa39efp-1;
    const double ehi = e * lhi;
    const uint32_t abstop = ((uint32_t)__double2hiint(ehi) >> 20) & 0x7ffu;
    const uint32_t topy = ((uint32_t)__double2hiint(e) >> 20) & 0x7ffu;
    if (topy < 0x3beu) return 1.0 + e;                        // |e| < 2^-65: pow's own early exit for x > 1
    if (topy < 0x43eu && abstop < 0x3c9u) return 1.0 + ehi;   // |e ln 2| < 2^-54: exp_inline's
    return pow2_cold(e);                                       // overflow, underflow, NaN, and the scaled arithmetic of 512 <= |e ln 2| < 1024
}
SRK_DEV double exp2_libm(double e)
{
    bool cold = false;
    double y = exp2_libm_plain(e, cold);
#ifndef SRK_OSC_NOCOLD
    if (__builtin_amdgcn_ballot_w64(cold) != 0) {
        if (cold) y = exp2_libm_special(e);
    }
#endif
    return y;
}

// The default mode's phase increment, 440 / sr * 2^e with the polynomial above — up to ONE CYCLE PER SAMPLE.  From there on (a pitch CV of
// +7 and more at 48 kHz: a gain close to 1 on a feedback cycle gets a patch there, the soak's seed 16340) the polynomial's error (1e-12
// when the seed was found, 3e-16 since) is no longer a relative error of something small but an absolute phase error per sample — at 2^60 cycles per sample either is —, which a saw or a square shows within a few hundred
// samples; what the reference renders up there is aliasing noise, but it is ITS noise: such an increment is evaluated as the reference
// spells it, with the host libm's own 2^e (out of line: the hot path pays a compare of the upper word and a branch).
// The same function serves every CV that holds its values (osc_step's recompute-on-change path, steposc_step): there the polynomial's
// 1e-12 is a CONSTANT error for as long as a note lasts, the phase drifts one way (8e-10 cycles after a second), the f32 roundings of the
// saw it produces flip one way too, and an oscillator that takes that saw as its pitch integrates the flips: 3.5e-5 on its square's
// edges after one second (the one-second soak's seed 30111).  A call per note is free; an audio-rate CV sweeps the polynomial's
// error through both signs and keeps the polynomial.
__device__ __attribute__((noinline)) double osc_delta_cold(double e, double sr) { return 440.0 * exp2_libm(e) / sr; }
template <bool kReduce = true>
SRK_DEV double osc_delta_fast(double e, double sr)
{
    double d = (440.0 / sr) * exp2_fast10<kReduce>(e);
    if (__builtin_expect((uint32_t)__double2hiint(d) >= 0x3ff00000u, 0)) d = osc_delta_cold(e, sr);  // d >= 1.0, a NaN, a negative rate
    return d;
}
// ... which the PROVED loops (below: a bound on |cv| for the whole launch) rule out beforehand: 440 / sr * 2^(val + bound) < 1, the
// exponent rounded up to an integer.  (Callers have checked bound + |val| < 1000.)
SRK_DEV bool osc_below_rate(float bound, double val, double sr)
{
    return __builtin_ldexp(440.0 / sr, (int)__builtin_ceil((double)__builtin_fabsf(bound) + val)) < 1.0;
}

// poly_blep, f64, literally (oscillator.rs:50-67)
SRK_DEV double poly_blep_exact(double t, double dt)
{
    if (dt == 0.0) return 0.0;
    if (t < dt) {
        t /= dt;
        return t + t - t * t - 1.0;
    } else if (t > 1.0 - dt) {
        t = (t - 1.0) / dt;
        return t * t + t + t + 1.0;
    }
    return 0.0;
}

// poly_blep in f32 from the f64-exact pair (t, t - 1).  `t < dt` <=> t/dt < 1 and
// `t > 1 - dt` <=> (t-1)/dt > -1; the polynomial is continuous (= 0) at both borders, so a
// border decided differently from the f64 compare costs nothing.  dt == 0 => inv_dt = inf =>
// ta = inf or NaN, tb = -inf: both tests false => 0, as the reference's early return.
SRK_DEV float poly_blep_fast(float t, float tm1, float inv_dt)
{
    float ta = t * inv_dt;
    float tb = tm1 * inv_dt;
    float fa = keep(__builtin_fmaf(ta, 2.0f - ta, -1.0f));  // 2t - t^2 - 1
    float fb = keep(__builtin_fmaf(tb, tb + 2.0f, 1.0f));   // t^2 + 2t + 1
    const float hi = tb > -1.0f ? fb : 0.0f;
    return ta < 1.0f ? fa : hi;
}
// The same polynomials with the reference's own f64 branch decisions (`t < dt`, else `t > 1.0 - dt`).  Needed whenever the
// increment may exceed 1/2: the two windows then OVERLAP, the branches no longer meet at 0 at a border, and a border decided
// by the rounded f32 quotient picks the wrong polynomial — which happens systematically, not rarely: an oscillator that starts at
// phase 0 sits exactly ON the border t == dt after its first step (found by tools/fv_soak.py at sample rate 1000: errors of 0.7).
SRK_DEV float poly_blep_sel(float t, float tm1, float inv_dt, bool first, bool second)
{
    float ta = t * inv_dt;
    float tb = tm1 * inv_dt;
    float fa = keep(__builtin_fmaf(ta, 2.0f - ta, -1.0f));
    float fb = keep(__builtin_fmaf(tb, tb + 2.0f, 1.0f));
    return first ? fa : (second ? fb : 0.0f);
}

// sin(2*pi*pos), pos in [0,1), as the reference's `(pos * PI * 2.0).sin() as f32` (oscillator.rs:133) up to the final
// rounding: folded to a quarter wave exactly (f64 subtractions of values in [-0.5, 0.5]), an odd polynomial of degree 13 in
// f64 (Chebyshev interpolant of sin(2 pi x) / x in x^2: max error 8e-14 on |x| <= 1/4; Estrin's scheme, dependency depth 3),
// ONE rounding to f32.  The result is the correctly rounded f32 sine except within ~1e-13 of a rounding boundary — i.e. it
// has the reference's own, unbiased, half-ulp error.  That matters because a sine that feeds a pitch CV (FM, vibrato) is
// INTEGRATED by the next oscillator's phase: an f32 evaluation (6e-8, biased) let config 4 drift to 5e-5 after one second;
// with this one default mode stays at f32 rounding level.
// The fold: qn = 1/2 - pos in (-1/2, 1/2] has sin(2 pi pos) = sin(2 pi qn) = sign(qn) sin(2 pi xa) with xa = 1/4 - ||qn| - 1/4| in [0, 1/4].
// Two f64 additions with |.| source modifiers instead of a reflection, an f64 compare and two selects per lane (14 issue cycles of a
// sine's ~70: the FM kernels are bound by exactly these).  Both additions are EXACT for every pos in [0, 1): |qn| > 1/4 is Sterbenz's
// case, and |qn| < 1/8 only happens for pos in (3/8, 5/8), where pos — and with it qn — is a multiple of 2^-54, which 1/4 - |qn| in
// (1/8, 1/4] can represent.  So xa == |qn| or 1/2 - |qn| exactly, as the reflection gave it.
SRK_DEV double sine_fold(double pos, uint32_t& sign)  // sign: bit 31 set where the sine is negative (for a pos in [0, 1))
{
    const double qn = 0.5 - pos;
    sign = (uint32_t)__double2hiint(qn) & 0x80000000u;
    const double t = __builtin_fabs(qn) - 0.25;
    return 0.25 - __builtin_fabs(t);
}
SRK_DEV float sine_fast(double pos)
{
    uint32_t sign;
    const double x = sine_fold(pos, sign);
    const double z = x * x;
    const double a01 = __builtin_fma(-41.34170223990684, z, 6.283185307179272);
    const double a23 = __builtin_fma(-76.70584757807868, z, 81.60524914955879);
    const double a45 = __builtin_fma(-15.081496425342264, z, 42.05813586028645);
    const double z2 = z * z;
    const double b0 = __builtin_fma(a23, z2, a01);
    const double b1 = __builtin_fma(3.6659216216293173, z2, a45);
    const double z4 = z2 * z2;
    const double p = __builtin_fma(b1, z4, b0);
    return __uint_as_float(__float_as_uint((float)(p * x)) ^ sign);  // (xor, not copysign: a phase outside [0, 1) — only a host can store one — folds to a negative x)
}
// The reference's own sine, `(pos * PI * 2.0).sin() as f32` (oscillator.rs:133), bit for bit — through the polynomial wherever that is decidable.
// The polynomial's value y is within 8e-14 y of sin(2 pi pos) (tests/test_oracle.py evaluates it against mpmath); the reference's f64 sine is
// within 1.6e-15 + 1.2e-16 y of it (one rounding of pos * PI, the libm's sub-ulp error).  Both round to the SAME f32 unless y lies within the
// sum of those of a rounding boundary: that is tested here (the two ends of that interval, converted), and only the lanes
// that fail — 3 in a million, the neighbourhoods of the sine's zeros, where the reference's value is its argument's rounding error —
// evaluate the reference's expression itself (ocml's sin, as the exact flavour did for every sample until round 5: 27.5 -> ms per step on
// config 4).  A phase outside [0, 1) — only a host can store one — takes that way too.
template <bool kRange = true>
SRK_DEV float sine_exact_plain(double pos, bool& cold)  // (branch-free; `cold` is set where the rounding is not decided here)
{
    uint32_t sign;
    const double x = sine_fold(pos, sign);
    const double z = x * x;
    const double a01 = __builtin_fma(-41.34170223990684, z, 6.283185307179272);
    const double a23 = __builtin_fma(-76.70584757807868, z, 81.60524914955879);
    const double a45 = __builtin_fma(-15.081496425342264, z, 42.05813586028645);
    const double z2 = z * z;
    const double b0 = __builtin_fma(a23, z2, a01);
    const double b1 = __builtin_fma(3.6659216216293173, z2, a45);
    const double z4 = z2 * z2;
    const double y = __builtin_fma(b1, z4, b0) * x;        // >= 0 for a pos in [0, 1)
    // Every value within d = 1e-13 y + 2e-15 of y — the polynomial's 8e-14 y, the reference's 1.6e-15 + 1.2e-16 y — rounds to the same f32 iff the
    // two ends of that interval do (rounding is monotone): two conversions and a compare, where round 5 measured y's distance to the rounding
    // boundary on its side (a conversion back, the half-ulp's exponent, its halving below a power of two, three compares: twice the
    // instructions).  tools/sine_check.c restates this on the CPU: 4.8e8 phases against glibc — uniform, around the quarter points, dyadic —, no
    // decided value differs; 3.4e-6 of uniformly distributed phases stay undecided (the zeros' neighbourhoods, where y - d < 0 <= y + d, among them).
    const double d = __builtin_fma(1.0e-13, y, 2.0e-15);
    const float r = (float)(y - d), r2 = (float)(y + d);
    bool sure = __float_as_uint(r) == __float_as_uint(r2);
    if (kRange) sure = sure && pos >= 0.0 && pos < 1.0;    // (a phase outside [0, 1) — only a host can store one — or a NaN: the reference's expression; kRange == false: the caller has proved it)
    cold = cold || !sure;
    return __uint_as_float(__float_as_uint(r) ^ sign);
}
SRK_DEV float sine_exact(double pos)
{
    bool cold = false;
    float r = sine_exact_plain(pos, cold);
#ifndef SRK_OSC_NOCOLD
    if (__builtin_amdgcn_ballot_w64(cold) != 0) {
        if (cold) r = (float)sin(pos * 3.14159265358979323846 * 2.0);
    }
#endif
    return r;
}

// The same sine for a port whose value cannot reach a pitch input (host-proved, OSC_SINE_LOOSE): nothing integrates its error,
// so f32 arithmetic after the exact f64 fold is inside the 1e-5 contract (max error 2e-7: a degree-9 polynomial in f32).
SRK_DEV float sine_loose(double pos)
{
    uint32_t sign;
    const float x = (float)sine_fold(pos, sign);
    const float z = x * x;
    const float a01 = __builtin_fmaf(-41.34168243408203f, z, 6.2831854820251465f);
    const float a23 = __builtin_fmaf(-76.58116912841797f, z, 81.60247802734375f);
    const float z2 = z * z;
    const float p = __builtin_fmaf(__builtin_fmaf(39.75982666015625f, z2, a23), z2, a01);
    return __uint_as_float(__float_as_uint(p * x) ^ sign);
}

SRK_DEV double wrap01(double x)
{
    // `pos %= 1.0` (fmod) for x >= 0: x - floor(x) is exact in f64; NaN/inf propagate as fmod's do
    return x - __builtin_floor(x);
}

// fmod(x, 1.0) bit for bit, without the library's bit-serial loop: for x >= 0 the remainder is x - floor(x), exact in f64
// (floor(x) shares x's exponent or lies below it; x >= 2^52 is an integer: 0, as fmod's; +inf and NaN give NaN both ways).
// A negative x — only reachable from a negative phase a host stored itself — keeps the dividend's sign in fmod: the library.
SRK_DEV double fmod1(double x)
{
    double r = x - __builtin_floor(x);
    if (__builtin_amdgcn_ballot_w64(x < 0.0) != 0) {
        if (x < 0.0) r = fmod1_cold(x);
    }
    return r;
}

// A square is what gates are made of (an LFO into an envelope, a clock into a sequencer): the consumer looks at `value > 0.0` only.  The
// f32 PolyBLEP is within 1e-7 of the reference's value — but just after the rising edge the reference's value is a tiny POSITIVE number
// (2x - x^2 with x = (pos - 0.5) / dt) which f32 rounds to exactly 0.0 for x < 1.5e-8: the gate would open a sample late, and an
// envelope a sample late is an error of one increment (1e-2), not 1e-7.  One edge in 7e7 — but the metric's configuration has millions of
// edges per second of audio.  So a value this close to zero (any lane: a wave-uniform branch, taken about once per 1e5 in-window samples)
// is re-evaluated with the reference's own f64 operations (oscillator.rs:50-67,135-142): same sign, same bits.
SRK_DEV float square_sign_safe(float sq, double pos, double delta)
{
    const bool tiny = __builtin_fabsf(sq) < 2.0e-6f;
    if (__builtin_amdgcn_ballot_w64(tiny) != 0) {
        if (tiny) {
            sq = (pos < 0.5 ? -1.0f : 1.0f) - (float)(poly_blep_exact(pos, delta) - poly_blep_exact(fmod1(pos + 0.5), delta));
        }
    }
    return sq;
}

// RN(a / b), the reference's `440.0 * 2^e / sample_rate`, without the division's dozen dependent instructions inside an oscillator's
// recurrence: y = RN(1 / b) is loop-invariant, q = RN(a y) is within an ulp of the quotient, r = a - b q is exact in an fma, and
// RN(q + r y) is the correctly rounded quotient (Markstein's theorem; b's significand — a sample rate — is nowhere near all ones).
// Checked against the division over 1e9 increments and thirteen sample rates on the CPU (notes/r05.md).  Quotients outside the normal
// range — an overflowed or vanishing 2^e — take the division itself.
SRK_DEV double div_rn_plain(double a, double b, bool& cold)  // (branch-free; `cold` is set where the quotient is outside the normal range)
{
    const double y = 1.0 / b;
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    cold = cold || !(__builtin_fabs(q) > 0x1p-900 && __builtin_fabs(q) < 0x1p900);
    return __builtin_fma(r, y, q);
}
// (the same where the caller has proved the quotient's range — a sample rate of 1 ... 65 535 under 440 * 2^e with |e| <= 800 — and holds y = RN(1 / b))
SRK_DEV double div_rn_proved(double a, double b, double y)
{
    const double q = a * y;
    const double r = __builtin_fma(-q, b, a);
    return __builtin_fma(r, y, q);
}
SRK_DEV double div_rn(double a, double b)
{
    bool cold = false;
    double out = div_rn_plain(a, b, cold);
#ifndef SRK_OSC_NOCOLD
    if (__builtin_amdgcn_ballot_w64(cold) != 0) {
        if (cold) out = a / b;
    }
#endif
    return out;
}
// An exact oscillator's sample where one of the branch-free forms could not decide (a 2^e outside pow's plain range, an increment outside
// the normal range, a sine within 1e-13 of an f32 rounding boundary): the reference's expressions themselves, out of line.
__device__ __attribute__((noinline)) void osc_exact_cold(double e, double sr, double pos, bool has_cv, double& delta, float& sine)
{
    if (has_cv) delta = 440.0 * exp2_libm(e) / sr;
    sine = (float)sin(pos * 3.14159265358979323846 * 2.0);
}

// (the increment alone, for a kernel that evaluates increments and sines of different samples side by side: render_fm_pair_block_x)
__device__ __attribute__((noinline)) double osc_delta_exact_cold(double e, double sr) { return 440.0 * exp2_libm(e) / sr; }

SRK_DEV void osc_step(uint32_t flags, OscRegs& s, const OscConst& c, float cv, float sync, float& sine, float& square, float& saw)
{
    if (flags & OSC_HAS_SYNC) {
        if (rising_edge(s.sync_last, sync)) s.pos = 0.0;
    } else {
        // sync_val = 0.0 (oscillator.rs:125-128): never above threshold; `last` still updates
        s.sync_last = false;
    }
    const double pos = s.pos;
    double delta;
    // An exact oscillator whose pitch sweeps (config 4's modulator, inside its feedback recurrence): increment and sine through their
    // branch-free forms, ONE test for "some lane could not decide" per sample — four separate wave-uniform branches were four scheduling
    // barriers in a loop that runs at one wave per SIMD.
    bool cold = false;
    const bool merged = (flags & (OSC_EXACT | OSC_HAS_CV | OSC_CV_AUDIO_RATE)) == (OSC_EXACT | OSC_HAS_CV | OSC_CV_AUDIO_RATE);
    if (merged) {
        const double e = (double)cv + c.val;
        delta = div_rn_plain(440.0 * exp2_libm_plain(e, cold), c.sr, cold);
        float sn = 0.0f;
        if (flags & OSC_OUT_SINE) sn = sine_exact_plain(pos, cold);
#ifndef SRK_OSC_NOCOLD
        if (__builtin_amdgcn_ballot_w64(cold) != 0) {
            if (cold) osc_exact_cold(e, c.sr, pos, true, delta, sn);
        }
#endif
        if (flags & OSC_OUT_SINE) sine = sn;
        s.seen_delta = delta;
        s.seen_cv = cv;
    } else if (flags & OSC_HAS_CV) {
        // 440 * 2^(f64(cv) + f64(val)) / f64(sample_rate), per sample (oscillator.rs:45,132)
        if ((flags & OSC_CV_AUDIO_RATE) || __builtin_amdgcn_ballot_w64(cv != s.seen_cv) != 0) {
            if (!(flags & OSC_EXACT) && (flags & OSC_VAL_FOLDED)) {  // (a kernel that proved a bound on |cv|: see the flags)
                double p;
                const bool d9 = (flags & OSC_CV_SERIES9) != 0;
                if (flags & OSC_CV_QUAD) {
                    p = d9 ? exp2_fast9<false>((double)(cv * 0.25f)) : exp2_fast<false>((double)(cv * 0.25f));
                    p = p * p;
                    p = p * p;
                } else if (flags & OSC_CV_SMALL) {
                    p = d9 ? exp2_fast9<false>((double)cv) : exp2_fast<false>((double)cv);
                } else {
                    p = d9 ? exp2_fast9<true>((double)cv) : exp2_fast<true>((double)cv);
                }
                s.seen_delta = c.scale * p;
            } else {
                const double e = (double)cv + c.val;
                // exact mode: 440 * 2^e / sr as written; default mode: (440 / sr) * 2^e with the series above
                // ... and a CV that HOLDS its values (a sequencer's notes, an envelope's sustain: no OSC_CV_AUDIO_RATE, the increment is only
                // recomputed when the CV changed) gets the reference's own increment too: the polynomial's error is a constant for a
                // constant CV, i.e. a phase that drifts one way for as long as the note lasts (osc_delta_cold)
                s.seen_delta = (flags & OSC_EXACT)           ? div_rn(440.0 * exp2_libm(e), c.sr)
                               : (flags & OSC_CV_SMALL)      ? (440.0 / c.sr) * exp2_fast<false>(e)  // (proved below the rate)
                               : (flags & OSC_CV_AUDIO_RATE) ? osc_delta_fast(e, c.sr)
                                                             : osc_delta_cold(e, c.sr);
            }
            s.seen_cv = cv;
        }
        delta = s.seen_delta;
    } else {
        delta = c.delta;
    }
    if (flags & OSC_EXACT) {
        const bool aa = flags & OSC_AA;
        if ((flags & OSC_OUT_SINE) && !merged) sine = sine_exact(pos);
        if (flags & OSC_OUT_SQUARE)
            square = (pos < 0.5 ? -1.0f : 1.0f) - (aa ? (float)(poly_blep_exact(pos, delta) - poly_blep_exact(fmod1(pos + 0.5), delta)) : 0.0f);
        if (flags & OSC_OUT_SAW) saw = ((float)pos * 2.0f - 1.0f) - (aa ? (float)poly_blep_exact(pos, delta) : 0.0f);
        s.pos = fmod1(pos + delta);
        return;
    }
    float inv_dt = c.inv_dt;
    if (flags & OSC_HAS_CV) inv_dt = 1.0f / (float)delta;
    if (flags & OSC_OUT_SINE) sine = (flags & OSC_SINE_LOOSE) ? sine_loose(pos) : sine_fast(pos);
    if ((flags & OSC_EXACT_BLEP) && (flags & (OSC_OUT_SQUARE | OSC_OUT_SAW))) {
        // saw / square exactly as the reference spells them (they reach a pitch input somewhere: an f32 PolyBLEP's biased 1e-7 would be
        // integrated into a phase); the phase itself, the increment and the sine keep the default arithmetic
        const bool aa = flags & OSC_AA;
        if (flags & OSC_OUT_SQUARE)
            square = (pos < 0.5 ? -1.0f : 1.0f) - (aa ? (float)(poly_blep_exact(pos, delta) - poly_blep_exact(fmod1(pos + 0.5), delta)) : 0.0f);
        if (flags & OSC_OUT_SAW) saw = ((float)pos * 2.0f - 1.0f) - (aa ? (float)poly_blep_exact(pos, delta) : 0.0f);
        s.pos = wrap01(pos + delta);
        return;
    }
    if (flags & (OSC_OUT_SQUARE | OSC_OUT_SAW)) {
        const float p32 = (float)pos;            // `self.pos as f32`
        float blep0 = 0.0f;
        const double upper = 1.0 - delta;        // the reference's `1.0 - dt`, rounded as it rounds it
        if (flags & OSC_AA) blep0 = poly_blep_sel(p32, (float)(pos - 1.0), inv_dt, pos < delta, pos > upper);
        if (flags & OSC_OUT_SAW) saw = __builtin_fmaf(p32, 2.0f, -1.0f) - blep0;  // p32*2 is exact => fma == mul,sub
        if (flags & OSC_OUT_SQUARE) {
            float blep1 = 0.0f;
            if (flags & OSC_AA) {
                double p2 = pos + 0.5;               // (pos + 0.5) % 1.0
                p2 = p2 >= 1.0 ? p2 - 1.0 : p2;
                blep1 = poly_blep_sel((float)p2, (float)(p2 - 1.0), inv_dt, p2 < delta, p2 > upper);
            }
            square = (pos < 0.5 ? -1.0f : 1.0f) - (blep0 - blep1);
            if (flags & OSC_AA) square = square_sign_safe(square, pos, delta);
        }
    }
    // x - floor(x) and v_fract_f64 agree for every finite x >= 0 (the subtraction is exact and below 1; fract's clamp to 1 - 2^-53 never
    // acts); they part at +inf (NaN against the instruction's own answer), which is why the caller has to have proved the precondition
    s.pos = (flags & OSC_PHASE_TAME) ? __builtin_amdgcn_fract(pos + delta) : wrap01(pos + delta);
}

// ---- what a WAVE can prove about an oscillator's pitch CV once per launch (default mode) ---------------------------------------------
// A CV whose magnitude is bounded for the whole launch — a sine through a gain: |cv| <= |gain| — lets the sample loop drop work
// (render_fm_pair in fused.hip.h; the kernels specialised at run time, jit.cpp): every lane votes with its own bound.
//   2: |cv| <= 1/2  2^cv needs no range reduction (OSC_CV_SMALL)   1: |cv| <= 2  2^cv = (2^(cv / 4))^4 (OSC_CV_QUAD)   0: neither
SRK_DEV int fm_gain_class(float bound)
{
    const float g = __builtin_fabsf(bound);  // the margins cover the f32 roundings of the products and sums the bound stands for
    if (__builtin_amdgcn_ballot_w64(!(g <= 0.4999f)) == 0) return 2;
    if (__builtin_amdgcn_ballot_w64(!(g <= 1.9999f)) == 0) return 1;
    return 0;
}
// A compile-time class per versioned oscillator of a specialised kernel: 0 nothing proved (the literal forms), 1 increments finite and
// phase in [0, 1) (one-instruction wrap, val folded into a per-launch scale), 2 = 1 and |cv| <= 2, 3 = 1 and |cv| <= 1/2.
template <uint32_t kValue>
struct UC {
    static constexpr uint32_t value = kValue;
};
constexpr uint32_t fm_class_flags(uint32_t cls)
{
    return cls == 0u ? 0u : (OSC_PHASE_TAME | OSC_VAL_FOLDED | OSC_CV_SERIES9 | (cls == 3u ? OSC_CV_SMALL : cls == 2u ? OSC_CV_QUAD : 0u));
}

// ---------------------------------------------------------------------------------------------
// Constant-pitch, unsynced oscillator — the hot special case (no CV, no sync, delta < 0.25).
// Same phase recurrence as osc_step (bit-identical pos); the per-sample work is cut down by
//   * carrying f32(pos) and t = pos/dt from one sample to the next: the wrapped next phase w
//     gives both next sample's `t < dt` term (w/dt) and this sample's `t > 1-dt` term
//     ((pos-1)/dt = w/dt - 1 when the phase wrapped), so one f64->f32 convert per sample;
//   * testing "is any lane of the wave inside a PolyBLEP window" on the high dword of pos with
//     integer compares; an LFO-rate oscillator is outside its windows for >99% of the samples
//     and then square == -1/+1 exactly, as in the reference (blep terms are exactly 0).
// ---------------------------------------------------------------------------------------------
struct COsc {
    double pos, delta;
    float p32;     // f32(pos)
    float ta;      // p32 * inv_dt  ( = t/dt of poly_blep's first branch)
    float inv_dt;
    // "near an edge" guards on hi32(pos): pos < dt | pos > 1-dt | |pos-0.5| < dt (with margin)
    int hA, hB, hQ0;
    uint32_t hQspan;
};

SRK_DEV void cosc_init(COsc& o, double pos, double delta)
{
    o.pos = pos;
    o.delta = delta;
    o.inv_dt = inv_dt_f32(delta);
    o.p32 = (float)pos;
    o.ta = o.p32 * o.inv_dt;
    // margin 2^-20 over every f64 rounding in the reference's compares; g = 0.25 makes the three
    // windows cover [0,1) (overlapping windows or a NaN delta: always take the full path)
    const double g = __builtin_fmin(delta * (1.0 + 9.5367431640625e-07) + 1e-300, 0.25);
    o.hA = __double2hiint(g);
    o.hB = __double2hiint(1.0 - g);
    o.hQ0 = __double2hiint(0.5 - g);
    o.hQspan = (uint32_t)(__double2hiint(0.5 + g) - o.hQ0);
}

SRK_DEV double cosc_advance(COsc& o, bool& wrapped)
{
    const double np = o.pos + o.delta;              // pos += delta
    wrapped = __double2hiint(np) >= 0x3ff00000;     // np >= 1.0 (np is non-negative)
    return __builtin_amdgcn_fract(np);              // pos %= 1.0: exact for 0 <= np < 2
}

// saw port only.  With t = pos/dt the two PolyBLEP branches are -(1 - t)^2 (first dt after the wrap) and
// (t' + 1)^2 = (w/dt)^2 (last dt before it, w = next phase).  max(1 - t, 0) is zero outside the first
// window, so that branch needs no compare; the second one is gated by the wrap bit of the f64 add.
SRK_DEV float cosc_saw(COsc& o)
{
    bool wrapped;
    const double w = cosc_advance(o, wrapped);
    const float w32 = (float)w;
    const float tn = w32 * o.inv_dt;
    const float base = __builtin_fmaf(o.p32, 2.0f, -1.0f);     // (pos as f32) * 2.0 - 1.0, exact as an fma
    const float u = __builtin_amdgcn_fmed3f(1.0f - o.ta, 0.0f, 1.0f);  // = max(1 - ta, 0) since ta >= 0; written as a [0,1] clamp so it folds into the subtract's output modifier
    const float s1 = __builtin_fmaf(u, u, base);               // base - (2t - t^2 - 1)
    const float s2 = keep(__builtin_fmaf(-tn, tn, s1));        // base - (t'^2 + 2t' + 1)
    const float saw = wrapped ? s2 : s1;
    o.pos = w;
    o.p32 = w32;
    o.ta = tn;
    return saw;
}

// square port only
SRK_DEV float cosc_square(COsc& o)
{
    const int h = __double2hiint(o.pos);
    const uint64_t near = __builtin_amdgcn_ballot_w64(h <= o.hA) | __builtin_amdgcn_ballot_w64(h >= o.hB) |
                          __builtin_amdgcn_ballot_w64((uint32_t)(h - o.hQ0) <= o.hQspan);
    float sq = h < 0x3fe00000 ? -1.0f : 1.0f;  // pos < 0.5
    if (near != 0) {
        const double pos = o.pos;
        const float blep0 = poly_blep_fast((float)pos, (float)(pos - 1.0), o.inv_dt);
        double p2 = pos + 0.5;
        p2 = p2 >= 1.0 ? p2 - 1.0 : p2;
        const float blep1 = poly_blep_fast((float)p2, (float)(p2 - 1.0), o.inv_dt);
        sq = square_sign_safe(sq - (blep0 - blep1), pos, o.delta);
    }
    bool wrapped;
    o.pos = cosc_advance(o, wrapped);
    return sq;
}

// sine port only
SRK_DEV float cosc_sine(COsc& o)
{
    const float s = sine_fast(o.pos);
    bool wrapped;
    o.pos = cosc_advance(o, wrapped);
    return s;
}

template <uint32_t kPort>
SRK_DEV float cosc_step(COsc& o)
{
    if (kPort == OSC_OUT_SAW) return cosc_saw(o);
    if (kPort == OSC_OUT_SQUARE) return cosc_square(o);
    return cosc_sine(o);
}

// ---- quiet groups (kernels specialised at run time, jit.cpp) --------------------------------------------------------------------------
// A gate LFO and the envelope behind it are event machines: between a voice's events — the LFO inside a PolyBLEP window (its only chance
// to change sign), the envelope's phase reaching 1, a gate level or edge that ends the segment — their per-sample work is
// `pos = fract(pos + delta)` and `phase += inc; out = c0 + c1 * (k0 + k1 * phase)`.  The per-sample forms (cosc_square, adsr_seg_step) ask
// "does any lane have an event NOW" every sample: three compares and two ballots for the LFO, two compares, two ballots, six scalar
// operations and a branch for the envelope, and the branch's cold arm costs the hot one a dozen register moves.  With every voice on
// its own clock (real polyphony: nothing is voice-invariant) that questioning is a third of the kernel.  A group of kQuietGroup samples
// asks ONCE, ahead: can any lane have an event within the group?  If not (two groups in three at 64 independent voices), the group
// runs the event-free forms below — the same operations on the same values in the same order as the per-sample forms' hot paths, so
// every bit of state and output is what they would have produced.
#ifndef SRK_QUIET_GROUP
#define SRK_QUIET_GROUP 8
#endif
constexpr int kQuietGroup = SRK_QUIET_GROUP;

struct COscQuiet {
    int hBq, hQ0q;      // guards on hi32(pos) like COsc's hB / hQ0, moved down by the phase a group covers
    uint32_t hQspanq;
};

SRK_DEV void cosc_quiet_init(const COsc& o, COscQuiet& q, int len = kQuietGroup)  // len: samples per group (a group that is not quiet is halved: 8, 4, 2, 1)
{
    const double g = __builtin_fmin(o.delta * (1.0 + 9.5367431640625e-07) + 1e-300, 0.25);  // cosc_init's window half-width
    const double w = g + (double)len * o.delta * (1.0 + 9.5367431640625e-07);                // ... plus the phase the group's samples cover
    if (w < 0.24) {
        q.hBq = __double2hiint(1.0 - w);
        q.hQ0q = __double2hiint(0.5 - w);
        q.hQspanq = (uint32_t)(__double2hiint(0.5 + g) - q.hQ0q);
    } else {  // fast oscillators (and a NaN increment): never quiet
        q.hBq = (int)0x80000000;
        q.hQ0q = 0;
        q.hQspanq = 0u;
    }
}

// this lane's square oscillator stays outside every PolyBLEP window for the next kQuietGroup samples: its value is the level of its
// phase's half, -1.0 or +1.0 exactly (both blep terms are 0.0, as in cosc_square's hot path), and does not change within the group
SRK_DEV bool cosc_square_quiet(const COsc& o, const COscQuiet& q)
{
    const int h = __double2hiint(o.pos);
    return !(h <= o.hA || h >= q.hBq || (uint32_t)(h - q.hQ0q) <= q.hQspanq);
}
SRK_DEV float cosc_square_level(const COsc& o) { return __double2hiint(o.pos) < 0x3fe00000 ? -1.0f : 1.0f; }
// (a quiet group ends below 1 - g: `pos %= 1.0` is the identity on every one of its sums, the wrap instruction is not needed)
SRK_DEV void cosc_quiet_step(COsc& o) { o.pos = o.pos + o.delta; }

// The exact render mode's constant-pitch square / saw behind the same guards (host-proved OSC_CONST_SMALL).  Outside its PolyBLEP
// windows the reference's own value is -1 / +1 (the two blep terms are 0.0 and 0.0 - 0.0 = 0.0) resp. (pos as f32) * 2 - 1, and its
// `pos %= 1.0` is the exact v_fract of a sum below 2; a sample with ANY lane inside a window (or a phase outside [0, 1): the guards
// read it as "near") takes osc_step's literal f64 formulas.  Bit-identical to oscillator.rs:108-158 either way.  Worth it where
// windows are rare: an LFO in any wave (64 lanes x 4 windows of 3.6e-5: 1 % of the samples), any oscillator of a control unit (one voice).
template <uint32_t kPort>
SRK_DEV float cosc_exact_step(COsc& o)
{
    const int h = __double2hiint(o.pos);
    uint64_t near = __builtin_amdgcn_ballot_w64(h <= o.hA) | __builtin_amdgcn_ballot_w64(h >= o.hB);
    if (kPort == OSC_OUT_SQUARE) near |= __builtin_amdgcn_ballot_w64((uint32_t)(h - o.hQ0) <= o.hQspan);
    if (near != 0) {
        OscRegs g;
        g.pos = o.pos;
        g.sync_last = false;
        OscConst k;
        k.delta = o.delta;
        k.val = 0.0;
        k.sr = 0.0;
        k.inv_dt = 0.0f;
        float o3[3] = {0.0f, 0.0f, 0.0f};
        osc_step(OSC_AA | OSC_EXACT | kPort, g, k, 0.0f, 0.0f, o3[0], o3[1], o3[2]);
        o.pos = g.pos;
        return kPort == OSC_OUT_SQUARE ? o3[1] : o3[2];
    }
    const float y = kPort == OSC_OUT_SQUARE ? (h < 0x3fe00000 ? -1.0f : 1.0f) : (float)o.pos * 2.0f - 1.0f;
    o.pos = __builtin_amdgcn_fract(o.pos + o.delta);
    return y;
}

// A constant-pitch oscillator of a CONTROL UNIT (one voice, every lane mirrors it), a tile at a time.  Only the phase is a recurrence —
// `pos = (pos + delta) % 1.0`, two f64 instructions per sample —, the output of sample j is a function of the phase before it: lane j
// keeps that phase, and after the tile's last step every lane evaluates ITS sample with the per-sample formulas (osc_step: the exact
// flavour's are the reference's own, so exact modes stay bit-identical; the default flavour's are the ones a voice kernel uses for a
// general oscillator).  A lone wave pays ~14 cycles per dependent instruction and ~40 per wave-uniform branch: the sample-by-sample form
// cost 35 (saw) ... 71 ns (square, a branch per sample) per sample, this one costs the two-instruction recurrence.
// Returns lane j's sample j (j < n); the caller stores the tile with one coalesced store.
template <bool kExact>
SRK_DEV float cosc_tile(uint32_t flags, COsc& o, int n)
{
    const int lane = (int)(threadIdx.x & 63u);
    double pos = o.pos, mine = o.pos;
    // v_fract_f64 is `% 1.0` for 0 <= x < 2 (cosc_advance); a phase outside [0, 1) — only a host can store one — takes the exact mode
    // through fmod's own path (the default mode has always used the instruction here)
    const bool tame = !kExact || __builtin_amdgcn_ballot_w64(!(pos >= 0.0 && pos < 1.0 && o.delta >= 0.0 && o.delta < 1.0)) == 0;
    if (tame && n == kTileRows) {
        // (not fully unrolled: the 32 `lane == j` masks are loop-invariant, the compiler hoists them out of the tile loop into 64 scalar
        // registers it does not have, and the spill code lands in the kernel that carries this block.  Rolled: 24 instead of 12 ns per
        // sample, behind the filter unit's 65 either way.)
#pragma unroll 8
        for (int j = 0; j < kTileRows; j++) {
            mine = lane == j ? pos : mine;
            pos = __builtin_amdgcn_fract(pos + o.delta);
        }
    } else {
        for (int j = 0; j < n; j++) {
            mine = lane == j ? pos : mine;
            pos = tame ? __builtin_amdgcn_fract(pos + o.delta) : fmod1(pos + o.delta);
        }
    }
    o.pos = pos;
    OscRegs g;
    g.pos = mine;
    g.sync_last = false;
    OscConst k;
    k.delta = o.delta;
    k.val = 0.0;
    k.sr = 0.0;
    k.inv_dt = o.inv_dt;
    float y0 = 0.0f, y1 = 0.0f, y2 = 0.0f;
    osc_step(flags, g, k, 0.0f, 0.0f, y0, y1, y2);
    return (flags & OSC_OUT_SAW) ? y2 : ((flags & OSC_OUT_SQUARE) ? y1 : y0);
}

// ---------------------------------------------------------------------------------------------
// A sequencer-driven pitch (default mode; host-proved OSC_CV_STEPWISE, PolyBLEP on, no sync, one live port): the carried-
// phase oscillator between note changes.  When some lane's CV differs from the one its increment was computed for (a wave-
// uniform test), that increment is recomputed — 440 / sr x 2^(cv + val), as osc_step does — and the carried terms are rebuilt
// from the exact f64 phase.  An increment of 0.25 or more (or NaN) breaks the carried form's "one PolyBLEP window at a
// time": those samples take osc_step.  Shared by the tile interpreter and the kernels specialised at run time.
// ---------------------------------------------------------------------------------------------
struct StepOsc {
    COsc o;
    float seen_cv;
    bool carried;
};

SRK_DEV void steposc_init(StepOsc& s, double pos)
{
    s.o.pos = pos;
    s.o.delta = 0.0;
    s.seen_cv = __builtin_nanf("");
    s.carried = false;
}

template <uint32_t kPort>
SRK_DEV float steposc_step(StepOsc& s, const OscConst& k, float cv)
{
    COsc& o = s.o;
    if (__builtin_amdgcn_ballot_w64(cv != s.seen_cv) != 0) {
        const double delta = osc_delta_cold((double)cv + k.val, k.sr);  // a held CV: the reference's own increment (osc_delta_cold)
        s.seen_cv = cv;
        s.carried = __builtin_amdgcn_ballot_w64(!(delta < 0.25)) == 0;
        cosc_init(o, o.pos, delta);
    }
    if (s.carried) return cosc_step<kPort>(o);
    OscRegs g;
    g.pos = o.pos;
    g.sync_last = false;
    g.seen_cv = s.seen_cv;
    g.seen_delta = o.delta;
    float o3[3] = {0.0f, 0.0f, 0.0f};
    osc_step(OSC_HAS_CV | OSC_CV_STEPWISE | OSC_AA | kPort, g, k, cv, 0.0f, o3[0], o3[1], o3[2]);
    cosc_init(o, g.pos, o.delta);
    return kPort == OSC_OUT_SAW ? o3[2] : (kPort == OSC_OUT_SQUARE ? o3[1] : o3[0]);
}

// ---------------------------------------------------------------------------------------------
// The same constant-pitch saw with the phase in 64-bit FIXED POINT (pos = phase * 2^64) — default mode of the fused
// voice kernels only (OSC_FIXED_PHASE, set by the host).  The reference accumulates the phase in f64 and wraps with
// fmod; on this hardware that is a v_add_f64, a v_fract_f64 and a v_cvt_f32_f64 per sample, each at half rate or less:
// 6 issue slots.  Two 32-bit integer adds and a v_cvt_f32_u32 do the same in 3, and the wrap is the carry.  The two
// accumulators differ by rounding only: f64 rounds every add to 2^-53 (a random walk of ~2e-14 after a second of
// audio), fixed point truncates the increment once to 2^-64 (a drift of < 3e-15 after a second) — both 9 orders of
// magnitude below the 1e-5 contract, and neither is "the" real-number phase.  f32(pos) — the only way the phase reaches
// the output — is the conversion of the upper 32 bits: the same value up to a double rounding that is itself below
// f32 resolution.  The exact mode, the interpreter and every gate-producing oscillator keep the f64 phase.
// ---------------------------------------------------------------------------------------------
struct FOsc {
    uint32_t lo, hi;    // pos * 2^64
    uint32_t dlo, dhi;  // delta * 2^64
    float c32;          // f32(hi) = f32(pos) * 2^32
    float ta;           // f32(pos) / dt — kept up to date in the lanes that need it: those whose last step wrapped
    float inv_s;        // 2^-32 / f32(delta)
    bool first;         // pos < delta: this sample is inside the first PolyBLEP window (<=> the step that led here wrapped)
};

SRK_DEV void fosc_init(FOsc& o, uint32_t lo, uint32_t hi, uint32_t dlo, uint32_t dhi)
{
    o.lo = lo;
    o.hi = hi;
    o.dlo = dlo;
    o.dhi = dhi;
    const double delta = __builtin_fma((double)dhi, 0x1p-32, (double)dlo * 0x1p-64);  // exact: 64 bits fit after the fma's single rounding to 53
    o.inv_s = inv_dt_f32(delta) * 0x1p-32f;
    o.c32 = (float)hi;
    o.ta = o.c32 * o.inv_s;
    o.first = hi < dhi || (hi == dhi && lo < dlo);
}

// saw port: as cosc_saw, with the carried terms derived from the upper phase word — and the two PolyBLEP corrections PREDICATED on the
// lanes that need them.  At the package power cap an instruction's price is its energy, and a lane that is switched off costs next to
// none (tools/energybench: v_fma_f32 0.70 nJ per wave-instruction with 64 lanes enabled, 0.32 with sixteen, 0.09 with one).  A voice is
// inside a window for two samples of its period (2 of 109 at 440 Hz): the first window is the sample after a wrap — `first`, the
// previous step's carry —, the second the sample whose own step wraps — this step's carry.  Neither needs a compare.  Outside them the
// select-free form computed u = max(1 - ta, 0) = 0 and took s1 = fma(0, 0, base) = base: the same bits as not computing them.
SRK_DEV float fosc_saw(FOsc& o)
{
    uint32_t nlo, nhi;
    const bool c0 = __builtin_add_overflow(o.lo, o.dlo, &nlo);
    const bool c1 = __builtin_add_overflow(o.hi, o.dhi, &nhi);
    const bool c2 = __builtin_add_overflow(nhi, (uint32_t)c0, &nhi);
    const bool wrapped = c1 | c2;                                  // pos + delta >= 1: `pos %= 1.0` is the dropped carry
    const float cn = (float)nhi;
    float saw = __builtin_fmaf(o.c32, 0x1p-31f, -1.0f);           // (pos as f32) * 2.0 - 1.0: the power-of-two scale is exact
    if (o.first) {                                                 // 2t - t^2 - 1 = -(1 - t)^2, t = pos / dt < 1 (or, rounded up to 1: u = 0)
        asm volatile("");                                          // (not to be speculated into a select: the point is the exec mask)
        const float u = __builtin_amdgcn_fmed3f(1.0f - o.ta, 0.0f, 1.0f);
        saw = __builtin_fmaf(u, u, saw);
    }
    if (wrapped) {                                                 // t'^2 + 2t' + 1 = (t' + 1)^2 = (next phase / dt)^2
        asm volatile("");
        const float tn = cn * o.inv_s;
        saw = __builtin_fmaf(-tn, tn, saw);
        o.ta = tn;
    }
    o.lo = nlo;
    o.hi = nhi;
    o.c32 = cn;
    o.first = wrapped;
    return saw;
}

// ---------------------------------------------------------------------------------------------
// The exact render mode's constant-pitch saw, a tile at a time (bit-identical to oscillator.rs:50-67,135-152).
// The reference's saw is `((pos as f32) * 2.0 - 1.0) - (poly_blep(pos, delta) as f32)` and poly_blep is 0.0 outside its two
// windows, `t < dt` (the first dt after the phase wrapped) and `t > 1.0 - dt` (the last dt before it wraps).  Inside a window
// it needs a true f64 division — and a wave pays for that whenever ANY of its 64 lanes is inside one: 69 % of the samples at
// 440 Hz, although each single lane is inside a window for 1.8 % of them.  So the tile is produced in two passes:
//   pass 1  every row gets the windowless value (exact there: x - 0.0 == x) while the phase advances; a lane whose step
//           wrapped (RN(pos + delta) >= 1) notes the row and its phase.  Both windows are adjacent to a wrap and nowhere else:
//           `pos > RN(1 - dt)` implies RN(pos + dt) >= 1 (rounding is monotonic), and without a wrap the next phase is
//           RN(pos + dt) >= dt, i.e. outside the first window.  The converse need not hold: the repair pass uses the
//           reference's own comparisons, so a noted row that is not inside a window after all just subtracts 0.0.
//   pass 2  each lane repairs its noted rows: the second-window value of the row that wrapped and the first-window value of
//           the row after it, each with the reference's f64 division and polynomial, subtracted in place in the LDS tile.
//           The divergent part runs once per tile (once more if some lane wraps twice within the tile), not per sample.
// The row after the tile's last row belongs to the next tile: every tile starts by checking its row 0 for the first window
// (which also covers the very first sample of a render, e.g. phase 0).
// Preconditions (checked once per launch, wave-uniform; otherwise the caller takes osc_step): 0 <= pos < 1, 0 <= delta <= 1/2.
// ---------------------------------------------------------------------------------------------
struct XSaw {
    double pos, delta;
    double pend_pos;     // phase of the noted row
    int pend_row;        // the noted row (-1: none)
    float row0_fix;      // first-window value of the tile's row 0 (0.0 outside the window)
};

SRK_DEV bool xsaw_usable(double pos, double delta)
{
    return __builtin_amdgcn_ballot_w64(!(pos >= 0.0 && pos < 1.0 && delta >= 0.0 && delta <= 0.5)) == 0;
}

// the two arms of poly_blep, each alone (a sample is in at most one: `if t < dt {..} else if t > 1.0 - dt {..}`)
SRK_DEV double poly_blep_first(double t, double dt)
{
    if (dt == 0.0 || !(t < dt)) return 0.0;
    t /= dt;
    return t + t - t * t - 1.0;
}
SRK_DEV double poly_blep_second(double t, double dt)
{
    if (dt == 0.0 || t < dt || !(t > 1.0 - dt)) return 0.0;
    t = (t - 1.0) / dt;
    return t * t + t + t + 1.0;
}

SRK_DEV void xsaw_init(XSaw& o, double pos, double delta)
{
    o.pos = pos;
    o.delta = delta;
    o.pend_pos = 0.0;
    o.pend_row = -1;
    o.row0_fix = 0.0f;
}

// `tile`: this lane's column of an LDS tile, `pitch` floats between rows
template <class Ptr>
SRK_DEV void xsaw_repair(XSaw& o, Ptr tile, int pitch, int n)
{
    if (o.pend_row >= 0) {
        const float b = (float)poly_blep_second(o.pend_pos, o.delta);
        tile[o.pend_row * pitch] = tile[o.pend_row * pitch] - b;
        if (o.pend_row + 1 < n) {  // (the tile's last row: its successor is row 0 of the next tile, see xsaw_tile)
            const double p1 = __builtin_amdgcn_fract(o.pend_pos + o.delta);
            const float a = (float)poly_blep_first(p1, o.delta);
            tile[(o.pend_row + 1) * pitch] = tile[(o.pend_row + 1) * pitch] - a;
        }
        o.pend_row = -1;
    }
}

// n rows (n <= 32) of the saw into the tile; on return o.pos is the phase after n samples
template <class Ptr>
SRK_DEV void xsaw_tile(XSaw& o, Ptr tile, int pitch, int n)
{
    float fix0 = 0.0f;
    if (__builtin_amdgcn_ballot_w64(o.pos < o.delta) != 0) fix0 = (float)poly_blep_first(o.pos, o.delta);
    auto row = [&](int i) {
        const double pos = o.pos;
        const float base = __builtin_fmaf((float)pos, 2.0f, -1.0f);  // (pos as f32) * 2.0 - 1.0: the doubling is exact, so the fma rounds once like mul, sub
        tile[i * pitch] = i == 0 ? base - fix0 : base;
        const double np = pos + o.delta;
        const bool wrapped = np >= 1.0;
        o.pos = __builtin_amdgcn_fract(np);  // fmod(np, 1.0) for 0 <= np < 2: exact
        if (__builtin_amdgcn_ballot_w64(wrapped) != 0) {
            if (__builtin_amdgcn_ballot_w64(wrapped && o.pend_row >= 0) != 0) xsaw_repair(o, tile, pitch, n);
            if (wrapped) {
                o.pend_pos = pos;
                o.pend_row = i;
            }
        }
    };
    if (n == 32) {
#pragma unroll 8
        for (int i = 0; i < 32; i++) row(i);
    } else {
        for (int i = 0; i < n; i++) row(i);
    }
    if (__builtin_amdgcn_ballot_w64(o.pend_row >= 0) != 0) xsaw_repair(o, tile, pitch, n);
}

// ---------------------------------------------------------------------------------------------
// Moog ladder — InternalMoogFilterState::calc + clamp_buffers, filter.rs:58-92
// ---------------------------------------------------------------------------------------------
struct VcfRegs {
    float f, p, q;
    float b0, b1, b2, b3, b4;
    float freq, res;  // the (frequency, res) the coefficients were computed for
};

// clamp_buffers: x.min(1.0).max(-1.0) (filter.rs:89).  kMed3: one v_med3_f32 instead of min+max —
// identical for every non-NaN x; a NaN (only reachable when an inf/NaN is fed into the filter)
// becomes -1.0 instead of the reference's +1.0 (v_med3 of a NaN is the minimum of the other two; writing it as -med3(-x, -1, 1)
// does not help: the compiler folds the negations into med3(x, 1, -1)).  The exact render mode uses the literal form.
template <bool kMed3>
SRK_DEV float clamp1(float x)
{
    if (kMed3) return __builtin_amdgcn_fmed3f(x, -1.0f, 1.0f);
    return fmaxf(fminf(x, 1.0f), -1.0f);
}

// filter.rs:61-68, the polynomials.  kFast folds their multiply-adds into fmas (10 instructions instead of 14; matters when an
// envelope sweeps the cutoff and the coefficients change every sample).
template <bool kFast>
SRK_DEV void vcf_polys(float frequency, float res, float& p, float& f, float& qq)
{
    const float q = 1.0f - frequency;
    if (kFast) {
        p = __builtin_fmaf(0.8f * frequency, q, frequency);
        f = __builtin_fmaf(p, 2.0f, -1.0f);
        qq = res * __builtin_fmaf(0.5f * q, __builtin_fmaf(5.6f * q, q, 1.0f - q), 1.0f);
    } else {
        p = frequency + 0.8f * frequency * q;
        f = p * 2.0f - 1.0f;
        qq = res * (1.0f + 0.5f * q * (1.0f - q + 5.6f * q * q));
    }
}

// Lanes whose (frequency, res) pair changed take new coefficients, the others keep theirs (`changed_lanes`: the ballot of `changed`,
// not zero).  A cutoff swept by an envelope changes in every lane at once: that case, wave-uniform and known before the arithmetic,
// computes straight into the state instead of into temporaries behind five selects.
template <bool kFast>
SRK_DEV void vcf_coeffs_update(VcfRegs& s, bool changed, uint64_t changed_lanes, float frequency, float res)
{
    if (changed_lanes == __builtin_amdgcn_ballot_w64(true)) {
        asm volatile("");  // (keeps the two arms apart: merged, they are the selects again, behind a select of their condition)
        s.freq = frequency;
        s.res = res;
        vcf_polys<kFast>(frequency, res, s.p, s.f, s.q);
    } else {
        float p, f, qq;
        vcf_polys<kFast>(frequency, res, p, f, qq);
        s.freq = changed ? frequency : s.freq;
        s.res = changed ? res : s.res;
        s.p = changed ? p : s.p;
        s.f = changed ? f : s.f;
        s.q = changed ? qq : s.q;
    }
}

// filter.rs:61-68 — recompute only when (frequency, res) changed.
template <bool kFast = false>
SRK_DEV void vcf_coeffs(VcfRegs& s, float frequency, float res)
{
    const bool changed = frequency != s.freq || res != s.res;
    const uint64_t changed_lanes = __builtin_amdgcn_ballot_w64(changed);
    if (changed_lanes != 0) vcf_coeffs_update<kFast>(s, changed, changed_lanes, frequency, res);  // wave-uniform skip
}

// The same for a kernel whose `res` cannot change during a launch (a parameter; only the cutoff has a CV): vcf_res_settle, once
// before the first sample, marks the lanes whose stored coefficients belong to another resonance (a NaN in `freq` differs from
// every frequency, and vcf_frequency never yields one), so that the per-sample test is the one compare of the frequency.
SRK_DEV void vcf_res_settle(VcfRegs& s, float res)
{
    if (res != s.res) s.freq = __builtin_nanf("");
}
template <bool kFast = false>
SRK_DEV void vcf_coeffs_freq(VcfRegs& s, float frequency, float res)
{
    const bool changed = frequency != s.freq;
    const uint64_t changed_lanes = __builtin_amdgcn_ballot_w64(changed);
    if (changed_lanes != 0) vcf_coeffs_update<kFast>(s, changed, changed_lanes, frequency, res);
}

// filter.rs:69-82 — returns lowpass; band/highpass through references (caller stores only live ports).
// kFast = false: the reference's f32 operations one by one, each rounded (bit-identical to the CPU tick given identical
// inputs; the exact render mode).  kFast = true (default mode): every `a * b - c * d` / `x - a * b` of the ladder keeps
// one product and folds the other into an fma — one rounding less per stage, i.e. closer to the real-number result than
// the reference itself, 6 of the filter's 27 instructions saved — and the clamps are v_med3.  The ladder is a damped
// recurrence (poles inside the unit circle, states clamped to [-1, 1]), so a 1-ulp difference per stage does not grow:
// the GPU tests hold default-mode renders to 1e-5 of the oracle over full-second renders.
// kMed3 (default: as kFast): the clamps as v_med3_f32.  Identical to min/max for every non-NaN value, so the exact flavour may
// use it too whenever no NaN can reach the filter: its states are clamped to [-1, 1] and its coefficients are finite, so a NaN
// can only come in through the input (see vcf_nan_free).
template <bool kFast = false, bool kMed3 = kFast>
SRK_DEV void vcf_step(VcfRegs& s, float input, float& lowpass, float& bandpass, float& highpass)
{
    if (kFast) {
        input = __builtin_fmaf(-s.q, s.b4, input);
        float t1 = s.b1;
        s.b1 = __builtin_fmaf(input + s.b0, s.p, -(s.b1 * s.f));
        float t2 = s.b2;
        s.b2 = __builtin_fmaf(s.b1 + t1, s.p, -(s.b2 * s.f));
        t1 = s.b3;
        s.b3 = __builtin_fmaf(s.b2 + t2, s.p, -(s.b3 * s.f));
        s.b4 = __builtin_fmaf(s.b3 + t1, s.p, -(s.b4 * s.f));
        s.b4 = __builtin_fmaf(-(s.b4 * s.b4 * s.b4), 0.166667f, s.b4);
    } else {
        input = input - (s.q * s.b4);
        float t1 = s.b1;
        s.b1 = (input + s.b0) * s.p - s.b1 * s.f;
        float t2 = s.b2;
        s.b2 = (s.b1 + t1) * s.p - s.b2 * s.f;
        t1 = s.b3;
        s.b3 = (s.b2 + t2) * s.p - s.b3 * s.f;
        s.b4 = (s.b3 + t1) * s.p - s.b4 * s.f;
        s.b4 = s.b4 - (s.b4 * s.b4 * s.b4) * 0.166667f;
    }
    s.b0 = clamp1<kMed3>(input);
    s.b1 = clamp1<kMed3>(s.b1);
    s.b2 = clamp1<kMed3>(s.b2);
    s.b3 = clamp1<kMed3>(s.b3);
    s.b4 = clamp1<kMed3>(s.b4);
    lowpass = s.b4;
    highpass = input - s.b4;
    bandpass = 3.0f * (s.b3 - s.b4);
}

// No lane of the wave holds a NaN (or an infinity) in the filter's state or coefficients: with a finite input every later
// value is finite too (the states are clamped, the arithmetic is sums and products of bounded numbers).
SRK_DEV bool vcf_nan_free(const VcfRegs& s)
{
    auto fin = [](float x) { return __builtin_fabsf(x) < __builtin_inff(); };
    return __builtin_amdgcn_ballot_w64(!(fin(s.f) && fin(s.p) && fin(s.q) && fin(s.b0) && fin(s.b1) && fin(s.b2) && fin(s.b3) && fin(s.b4))) == 0;
}

// The ladder as a kernel calls it.  kFast: the default mode's contracted form.  Otherwise the literal operations — with the clamps
// as v_med3 (half the instructions of min + max) whenever that cannot be told from the reference: the two differ only on a NaN,
// and a NaN can only come from a NaN / infinite input or from an input so large that the cubic overflows (inf - inf).  `fin` (wave-
// uniform) says the state is finite — vcf_nan_free at load, and true after any step, whose clamps leave finite values behind (a
// NaN clamps to 1.0 in the literal form; the coefficients are products of clamped numbers) — and an input below 1e6 in every lane
// keeps every intermediate finite.  Anything else takes the literal min / max for that sample.
template <bool kFast>
SRK_DEV void vcf_run(VcfRegs& s, bool& fin, float input, float& lowpass, float& bandpass, float& highpass)
{
    if (kFast) {
        vcf_step<true>(s, input, lowpass, bandpass, highpass);
        return;
    }
    if (fin && __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(input) < 1e6f)) == 0) {
        vcf_step<false, true>(s, input, lowpass, bandpass, highpass);
    } else {
        vcf_step<false, false>(s, input, lowpass, bandpass, highpass);
        fin = true;
    }
}

// The literal ladder behind an input that is known to be finite and small (`bounded`, wave-uniform: e.g. the tile-wise exact saw, whose
// preconditions bound it by 2): no per-sample look at the input.
SRK_DEV void vcf_run_bounded(VcfRegs& s, bool& fin, bool bounded, float input, float& lowpass, float& bandpass, float& highpass)
{
    if (fin && bounded)
        vcf_step<false, true>(s, input, lowpass, bandpass, highpass);
    else
        vcf_run<false>(s, fin, input, lowpass, bandpass, highpass);
}

// (self.freq + cv * self.exp_amt).max(0.0).min(0.9), filter.rs:213
SRK_DEV float vcf_frequency(float freq, float cv, float exp_amt) { return fminf(fmaxf(freq + cv * exp_amt, 0.0f), 0.9f); }
SRK_DEV float vcf_resonance(float res) { return fminf(fmaxf(res, 0.0f), 1.0f); }  // filter.rs:214
// default mode: the two clamps as one v_med3_f32 — the same value for every input, NaN included (max(NaN, 0) = 0 = the smaller of the
// other two); only a frequency of -0.0 may come out as either zero
SRK_DEV float vcf_frequency_med3(float freq, float cv, float exp_amt) { return __builtin_amdgcn_fmed3f(freq + cv * exp_amt, 0.0f, 0.9f); }

// ---------------------------------------------------------------------------------------------
// ADSR — adsr.rs:138-214
// ---------------------------------------------------------------------------------------------
struct AdsrRegs {
    float phase, r_val, from_a_val;
    int mode;
    bool gate_last;
};

struct AdsrConst {
    float inc_a, inc_d, inc_r;  // 1.0 / (sample_rate * X_sec): loop-invariant, hoisted (same bits)
    float s_val;
};

SRK_DEV AdsrConst adsr_consts(float a_sec, float d_sec, float s_val, float r_sec, float sample_rate)
{
    AdsrConst c;
    c.inc_a = 1.0f / (sample_rate * a_sec);  // a_sec = 0 => +inf: Attack lasts one sample (adsr.rs:39,152-156)
    c.inc_d = 1.0f / (sample_rate * d_sec);
    c.inc_r = 1.0f / (sample_rate * r_sec);
    c.s_val = s_val;
    return c;
}

SRK_DEV float adsr_step(uint32_t flags, AdsrRegs& s, const AdsrConst& c, float gate)
{
    const bool has_gate = flags & ADSR_HAS_GATE;
    const float g = has_gate ? gate : 0.0f;
    const bool is_transition = rising_edge(s.gate_last, g);
    const bool high = has_gate && gate > 0.0f;
    const bool low = !has_gate || gate <= 0.0f;  // adsr.rs:175 spells Sustain's test `<= 0.0`: a NaN gate is neither high nor low
    switch (s.mode) {
    case SRACK_ADSR_MODE_NONE:
        if (high) {
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_ATTACK;
        }
        break;
    case SRACK_ADSR_MODE_ATTACK:
        s.phase += c.inc_a;
        if (s.phase >= 1.0f) {
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_DECAY;
        } else if (is_transition) {
            s.phase = 0.0f;
            s.r_val = s.from_a_val;
        }
        break;
    case SRACK_ADSR_MODE_DECAY:
        s.phase += c.inc_d;
        if (s.phase >= 1.0f) {
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_SUSTAIN;
        }
        if (is_transition) {
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_ATTACK;
        }
        break;
    case SRACK_ADSR_MODE_SUSTAIN:
        if (low) {  // gate_in_buf.is_none() || gate <= 0.0
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_RELEASE;
        }
        if (is_transition) {
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_ATTACK;
        }
        break;
    default:  // SRACK_ADSR_MODE_RELEASE
        if (high) {
            s.phase = 0.0f;
            s.mode = SRACK_ADSR_MODE_ATTACK;
        }
        s.phase += c.inc_r;  // yes, also right after switching to Attack (adsr.rs:187-199)
        if (s.phase >= 1.0f) {
            s.phase = 0.0f;
            s.r_val = 0.0f;
            s.mode = SRACK_ADSR_MODE_NONE;
        }
        break;
    }
    float out;
    switch (s.mode) {
    case SRACK_ADSR_MODE_NONE: out = 0.0f; break;
    case SRACK_ADSR_MODE_ATTACK: out = s.r_val + (1.0f - s.r_val) * s.phase; break;
    case SRACK_ADSR_MODE_DECAY: out = c.s_val + (1.0f - c.s_val) * (1.0f - s.phase); break;
    case SRACK_ADSR_MODE_SUSTAIN: out = c.s_val; break;
    default: out = c.s_val * (1.0f - s.phase); break;
    }
    if (s.mode != SRACK_ADSR_MODE_ATTACK)
        s.r_val = out;
    else
        s.from_a_val = out;
    return out;
}

// What an envelope can reach, for the kernels that version an oscillator on a bound of its pitch CV (jit.cpp, analyze): every output is a
// convex combination of values already in the hull of {0, 1, s_val, r_val, from_a_val} — Attack r + (1 - r) phase, Decay s + (1 - s)(1 -
// phase), Sustain s, Release s (1 - phase), each with 0 <= phase < 1 — provided the three increments are not negative (a negative time
// constant would run a phase below 0) and the stored phase is a phase.
SRK_DEV float adsr_bound(const AdsrRegs& s, const AdsrConst& c)
{
    return __builtin_fmaxf(__builtin_fmaxf(1.0f, __builtin_fabsf(c.s_val)), __builtin_fmaxf(__builtin_fabsf(s.r_val), __builtin_fabsf(s.from_a_val)));
}
SRK_DEV bool adsr_tame(const AdsrRegs& s, const AdsrConst& c)
{
    return c.inc_a >= 0.0f && c.inc_d >= 0.0f && c.inc_r >= 0.0f && s.phase >= 0.0f && s.phase <= 1.0f && __builtin_fabsf(c.s_val) < 1.0e30f &&
           __builtin_fabsf(s.r_val) < 1.0e30f && __builtin_fabsf(s.from_a_val) < 1.0e30f;
}

// Segmented ADSR: between mode changes the envelope is  phase += inc; out = c0 + c1 * u  with
// u = phase (Attack) or 1 - phase, and (inc, c0, c1) fixed — the same f32 operations adsr_step
// performs for that mode, so the bits are identical.  A sample that may change the mode
// (phase >= 1, gate level or edge, depending on the mode) sends the whole wave through adsr_step.
struct AdsrSeg {
    float inc, c0, c1;
    float k0, k1;    // u = k0 + k1 * phase: (0, 1) in Attack (u = phase), (1, -1) otherwise (u = 1 - phase); both exact
    float held;      // previous sample's output: r_val / from_a_val are materialised from it on demand
    // lane masks (SGPR pairs): the per-sample "does any lane leave its segment" test is scalar-unit work
    uint64_t attack, on_high, on_low, on_edge, last;
};

SRK_DEV bool lane_bit(uint64_t mask) { return (mask >> (threadIdx.x & 63)) & 1u; }

SRK_DEV void adsr_seg_enter(const AdsrRegs& s, const AdsrConst& c, AdsrSeg& g)
{
    const int m = s.mode;
    const bool a = m == SRACK_ADSR_MODE_ATTACK, d = m == SRACK_ADSR_MODE_DECAY, su = m == SRACK_ADSR_MODE_SUSTAIN, r = m == SRACK_ADSR_MODE_RELEASE;
    g.k0 = a ? 0.0f : 1.0f;
    g.k1 = a ? 1.0f : -1.0f;
    g.inc = a ? c.inc_a : (d ? c.inc_d : (r ? c.inc_r : 0.0f));
    g.c0 = a ? s.r_val : ((d || su) ? c.s_val : 0.0f);
    g.c1 = a ? 1.0f - s.r_val : (d ? 1.0f - c.s_val : (r ? c.s_val : 0.0f));
    g.attack = __builtin_amdgcn_ballot_w64(a);
    g.on_high = __builtin_amdgcn_ballot_w64(!(a || d || su));  // None, Release: a high gate starts an attack
    g.on_low = __builtin_amdgcn_ballot_w64(su);                // Sustain: gate low starts the release
    g.on_edge = __builtin_amdgcn_ballot_w64(a || d || su);     // Attack, Decay, Sustain: a rising edge retriggers
    g.last = __builtin_amdgcn_ballot_w64(s.gate_last);
    g.held = a ? s.from_a_val : s.r_val;
}

SRK_DEV void adsr_seg_flush(AdsrRegs& s, const AdsrSeg& g)
{
    if (lane_bit(g.attack))
        s.from_a_val = g.held;
    else
        s.r_val = g.held;
    s.gate_last = lane_bit(g.last);
}

SRK_DEV float adsr_seg_step(AdsrRegs& s, const AdsrConst& c, AdsrSeg& g, float gate)
{
    const float ph = s.phase + g.inc;
    const uint64_t m_over = __builtin_amdgcn_ballot_w64(ph >= 1.0f);
    const uint64_t m_high = __builtin_amdgcn_ballot_w64(gate > 0.0f);
    const uint64_t m_leave = m_over | (m_high & g.on_high) | (~m_high & g.on_low) | (m_high & ~g.last & g.on_edge);
    float out;
    if (m_leave != 0) {
        adsr_seg_flush(s, g);
        out = adsr_step(ADSR_HAS_GATE, s, c, gate);
        adsr_seg_enter(s, c, g);
    } else {
        s.phase = ph;
        g.last = m_high;
        const float u = __builtin_fmaf(g.k1, ph, g.k0);  // (k1 = +-1: the product is exact, the one rounding is the sum's — k0 + k1 * ph bit for bit)
        out = g.c0 + g.c1 * u;
    }
    g.held = out;
    return out;
}

// Quiet groups (see cosc_quiet_init): no lane leaves its segment during the next kQuietGroup samples, given that its gate holds the value
// `gate` for all of them (a group-constant gate: a quiet square oscillator's level).  Wave-uniform part: the gate's level or edge ends no
// lane's segment — adsr_seg_step's own mask, evaluated once; a constant gate has its only possible edge at the group's first sample.
// Per-lane part: the phase stays below 1 — each of the group's `len` rounded additions adds at most inc + 2^-24 (an infinite or NaN increment
// says no).  `m_high` returns the gate's level mask: what adsr_seg_step would leave in g.last after every one of the group's samples.
SRK_DEV bool adsr_seg_quiet(const AdsrRegs& s, const AdsrSeg& g, float gate, uint64_t& m_high, int len = kQuietGroup)
{
    m_high = __builtin_amdgcn_ballot_w64(gate > 0.0f);
    const uint64_t m_leave = (m_high & g.on_high) | (~m_high & g.on_low) | (m_high & ~g.last & g.on_edge);
    return m_leave == 0 && s.phase + (float)len * g.inc <= 0.9999f;
}
// The VCA behind such an envelope asks `cv > 0.0` every sample (vca.rs:132).  Within a quiet group the answer is one per lane: the
// segment's output c0 + c1 u is monotonic in u, u moves one way, and with c0, c1 >= 0 it is positive from the first sample on if c0 > 0
// or c1 u > 0 — OPEN — and exactly +0.0 if both are zero (mode None; Sustain at level 0; a release from 0) — CLOSED.  `decided`: this
// lane is one or the other (a negative level, or a product that could underflow, is neither: the group then takes the per-sample forms).
SRK_DEV bool adsr_seg_open(const AdsrRegs& s, const AdsrSeg& g, bool& decided)
{
    const bool closed = g.c0 == 0.0f && g.c1 == 0.0f;
    // u of the group's samples: phase + inc ... (Attack), 1 - phase - inc ... >= 1e-4 (otherwise; the group's own guard keeps phase <= 0.9999)
    const float u_first = g.k0 + g.k1 * (s.phase + g.inc);
    const bool open = g.c0 >= 0.0f && g.c1 >= 0.0f && (g.c0 > 0.0f || (g.c1 > 1.0e-30f && u_first > 1.0e-6f)) && g.c0 < 1.0e30f && g.c1 < 1.0e30f;
    decided = open || closed;
    return open;
}
// ... and the VCA's sample with that answer (a per-lane bool that is constant over the group lives in an SGPR pair: one v_cndmask, no compare)
SRK_DEV float vca_step_decided(bool open_lane, float audio, float cv)
{
    const float y = audio * cv;
    return open_lane ? y : 0.0f;
}

// one sample of such a group: adsr_seg_step's hot path without its questions (the caller sets g.last = m_high after the group)
SRK_DEV float adsr_seg_quiet_step(AdsrRegs& s, AdsrSeg& g)
{
    s.phase = s.phase + g.inc;
    const float u = __builtin_fmaf(g.k1, s.phase, g.k0);  // (= k0 + k1 * phase bit for bit: k1 = +-1)
    const float out = g.c0 + g.c1 * u;
    g.held = out;
    return out;
}

// The envelope of a CONTROL UNIT (one voice, every lane mirrors it), a tile at a time.  `gate`: lane j holds the gate of sample j.
// The per-sample form above decides "does this sample leave the segment" with two compares, two ballots and a branch per sample; for one
// voice the same question has a closed answer per RUN of samples: the gate bits of the whole tile are one ballot, so the first sample
// whose gate level or edge ends the segment is a count-trailing-zeros away, and the phase cannot reach 1.0 within
// K = (0.9999 - phase) / inc steps (each rounded step adds at most inc + 2^-24: the margin covers a whole tile).  The run up to there is
// the segment's own operations and nothing else — phase += inc; out = c0 + c1 * (k0 + k1 * phase): the same f32 operations in the same
// order as adsr.rs, so the bits are the reference's in every mode — and the sample that ends it goes through adsr_seg_step, which decides
// as before.  ~100 -> ~20 ns per sample.  Returns lane j's sample j (j < n).
SRK_DEV float adsr_seg_tile(AdsrRegs& s, const AdsrConst& c, AdsrSeg& g, float gate, int n)
{
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t tile = (1ull << n) - 1ull;  // n <= 32
    const uint64_t high = __builtin_amdgcn_ballot_w64(gate > 0.0f) & tile;
    const uint64_t before = (high << 1) | (g.last != 0 ? 1ull : 0ull);  // bit j: sample j's predecessor was high
    float keep = 0.0f;
    int i = 0;
    while (i < n) {
        const uint64_t ends = ((g.on_high != 0 ? high : 0ull) | (g.on_low != 0 ? ~high : 0ull) | (g.on_edge != 0 ? (high & ~before) : 0ull)) & tile & (~0ull << i);
        const int by_gate = ends != 0 ? (int)__builtin_ctzll(ends) : n;
        const float room = (0.9999f - s.phase) * __builtin_amdgcn_rcpf(g.inc);  // inc == 0 (Sustain, None): +inf; NaN / negative: no run
        const int by_phase = __builtin_amdgcn_readfirstlane(room > 0.0f ? (int)__builtin_fminf(room, 64.0f) : 0);
        const int run = min(by_gate - i, by_phase);
        if (run > 0) {
            float ph = s.phase, out = g.held;
            for (int k = 0; k < run; k++) {
                ph = ph + g.inc;
                const float u = __builtin_fmaf(g.k1, ph, g.k0);  // (k1 = +-1: the product is exact, the one rounding is the sum's — k0 + k1 * ph bit for bit)
                out = g.c0 + g.c1 * u;
                keep = lane == i + k ? out : keep;
            }
            s.phase = ph;
            g.held = out;
            i += run;
            g.last = ((high >> (i - 1)) & 1ull) ? ~0ull : 0ull;
        }
        if (i < n) {  // the sample that may end the segment (or only the end of a cautious run): decided as it always was
            const float y = adsr_seg_step(s, c, g, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gate), i)));
            keep = lane == i ? y : keep;
            i++;
        }
    }
    return keep;
}

// ---------------------------------------------------------------------------------------------
// VCA vca.rs:127-144, mixer mixer.rs:109-118, math math.rs:46-52,149-156
// ---------------------------------------------------------------------------------------------
SRK_DEV float vca_step(uint32_t flags, bool negative, float audio, float cv)
{
    if ((flags & (VCA_HAS_AUDIO | VCA_HAS_CV)) != (VCA_HAS_AUDIO | VCA_HAS_CV)) return zero_f32();  // output.fill(0.0)
    return (negative || cv > 0.0f) ? audio * cv : 0.0f;
}

// the same with a wave-uniform cv (a control track read through the scalar unit): `cv > 0.0` is decided on the scalar unit from the
// bit pattern — positive, non-zero, not NaN  <=>  0 < bits <= 0x7f800000
SRK_DEV float vca_step_uniform(uint32_t flags, bool negative, float audio, float cv)
{
    if ((flags & (VCA_HAS_AUDIO | VCA_HAS_CV)) != (VCA_HAS_AUDIO | VCA_HAS_CV)) return zero_f32();
    const bool cv_pos = (uint32_t)(__float_as_int(cv) - 1) < 0x7f800000u;
    return (negative || cv_pos) ? audio * cv : 0.0f;
}

SRK_DEV float mixer_step(uint32_t connected, const float in[4], const float gain[4])
{
    float out = zero_f32();  // output.fill(0.0), then one `*dst += src * gain` pass per connected input
#pragma unroll
    for (int k = 0; k < 4; k++)
        if (connected & (1u << k)) out = out + in[k] * gain[k];
    return out;
}

SRK_DEV float math_step(uint32_t flags, float in1, float in2, float constant)
{
    float a = (flags & MATH_HAS_IN1) ? in1 : zero_f32();
    float b = (flags & MATH_HAS_IN2) ? in2 : constant;
    switch ((flags >> MATH_OP_SHIFT) & 3u) {
    case SRACK_MATH_ADD: return a + b;
    case SRACK_MATH_SUBTRACT: return a - b;
    default: return a * b;
    }
}

// ---------------------------------------------------------------------------------------------
// NoiseModule — oscillator.rs:381-387: `(rand::random::<f32>() - 0.5) * 2.0`.  rand 0.8's Standard f32 is
// (next_u32() >> 8) as f32 * 2^-24: 2^24 equally likely values in [0, 1).  The reference's generator is the OS-seeded
// thread-local ChaCha12, so only that distribution is reproducible; the draw is this library's (srack_hip.h): sample n of
// a voice is output n of a splitmix64 generator seeded with the voice's key.  (r - 0.5) * 2 is exact in f32.
// ---------------------------------------------------------------------------------------------
SRK_DEV uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
SRK_DEV uint64_t noise_voice_key(uint64_t base, uint64_t global_voice) { return splitmix64(base ^ global_voice); }
SRK_DEV float noise_sample(uint64_t key, uint64_t n)
{
    const uint32_t k24 = (uint32_t)(splitmix64(key + n * 0x9E3779B97F4A7C15ull) >> 40);
    return ((float)k24 * 0x1p-24f - 0.5f) * 2.0f;
}

// ---------------------------------------------------------------------------------------------
// NonLinearModule — math.rs:203-205: `if a > 0.0 { a.powf(b) } else { -(-a).powf(b) }`
// ---------------------------------------------------------------------------------------------
// `a.powf(b)` is the HOST libm's powf (Rust's f32::powf), and glibc's powf (2.27 and later: Szabolcs Nagy's algorithm, sysdeps/ieee754/flt-32/
// e_powf.c) is within 0.82 ulp of the true power: NOT the correctly rounded float.  Until round 6 the kernels evaluated 2^(b log2 x) in f64 with
// tables of their own and rounded once — the correctly rounded power but for one argument in 1e13, and therefore an f32 ulp away from the
// reference wherever the libm is (P4: 6.0e-8 in every render mode; a waveshaper inside a feedback loop: a different render — the fuzzer's seed
// 405576, 1.97 off).  This is the algorithm itself, operation for operation as the x86-64 FMA build of glibc 2.35 executes it (disassembled: which
// products are contracted into fused multiply-adds is the compiler's choice, and part of the result):
//     log2 x:  x = 2^k z, z in [0x1.66p-1, 0x1.66p0) (OFF = 0x3f330000), the top four bits of z's offset mantissa pick c with 1 / c and log2 c from
//              its 16-entry table, r = fma(z, 1 / c, -1), y0 = log2 c + k,
//              log2 x = fma(fma(r, A0, A1), r^4, fma(r^2, fma(r, A2, A3), fma(r, A4, y0)))
//     2^(y log2 x):  kd = ylogx + 0x1.8p52 / 32 - the same, r = ylogx - kd, 2^(k / 32) from pow2f_libm's table,
//              fma(fma(r, C0, C1), r^2, fma(r, C2, 1)) * 2^(k / 32), ONE rounding to f32 (subnormal results round there)
// with its overflow (ylogx > 0x1.fffffffd1d571p+6) and underflow (ylogx <= -150) answers and its normalisation of a subnormal x.  The table is the
// host libm's own (tools/powf_tables.py reads it out of libm.so.6); tests/libm_powf.py transliterates this function with exact fused multiply-adds
// and tests/test_oracle.py holds it to the host's powf.  Zero, infinite and NaN arguments (and a negative x: no caller has one) go to ocml's powf,
// which answers them as C99 Annex F prescribes.
__device__ const uint64_t kPowfLog2Tab[16][2] = {  // glibc's __powf_log2_data.tab: {bits(1 / c_i), bits(log2 c_i)}
    {0x3ff661ec79f8f3be, 0xbfdefec65b963019}, {0x3ff571ed4aaf883d, 0xbfdb0b6832d4fca4}, {0x3ff49539f0f010b0, 0xbfd7418b0a1fb77b},
    {0x3ff3c995b0b80385, 0xbfd39de91a6dcf7b}, {0x3ff30d190c8864a5, 0xbfd01d9bf3f2b631}, {0x3ff25e227b0b8ea0, 0xbfc97c1d1b3b7af0},
    {0x3ff1bb4a4a1a343f, 0xbfc2f9e393af3c9f}, {0x3ff12358f08ae5ba, 0xbfb960cbbf788d5c}, {0x3ff0953f419900a7, 0xbfaa6f9db6475fce},
    {0x3ff0000000000000, 0x0000000000000000}, {0x3fee608cfd9a47ac, 0x3fb338ca9f24f53d}, {0x3feca4b31f026aa0, 0x3fc476a9543891ba},
    {0x3feb2036576afce6, 0x3fce840b4ac4e4d2}, {0x3fe9c2d163a1aa2d, 0x3fd40645f0c6651c}, {0x3fe886e6037841ed, 0x3fd88e9c2c1b9ff8},
    {0x3fe767dcf5534862, 0x3fdce0a44eb17bcc},
};
// (pow2f_libm's table, below: tab[i] = bits(2^(i/32)) - (i << 47))
__device__ const uint64_t kExp2fTab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa,
    0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d,
    0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74,
    0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d,
    0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540,
};
// (The tables of this section, as a kernel reaches them.  A lane's index is its own: through the global pointer that is a gather for the
// vector-memory pipe — 16 cycles of its address unit per wave and load at best, shared by the CU's four SIMDs; P4, with three such
// loads per voice-sample, was bound by exactly that (rocprofv3: 94 VALU instructions in 677 SIMD cycles per wave-sample, the same at
// two and at four waves per SIMD).  A kernel generated for a patch copies the tables into LDS once per launch and passes LDS pointers:
// ds_read_b64 / b128 with per-lane addresses.)
typedef __attribute__((address_space(3))) const uint64_t LdsTab;
struct GlobalTables {
    SRK_DEV uint64_t exp2f(uint32_t i) const { return kExp2fTab[i]; }
    SRK_DEV uint64_t log2_inv_c(uint32_t i) const { return kPowfLog2Tab[i][0]; }
    SRK_DEV uint64_t log2_c(uint32_t i) const { return kPowfLog2Tab[i][1]; }
};
struct LdsTables {
    LdsTab* exp2f_tab;  // [32]
    LdsTab* log2_tab;   // [16][2]
    SRK_DEV uint64_t exp2f(uint32_t i) const { return exp2f_tab[i]; }
    SRK_DEV uint64_t log2_inv_c(uint32_t i) const { return log2_tab[2u * i]; }
    SRK_DEV uint64_t log2_c(uint32_t i) const { return log2_tab[2u * i + 1u]; }
};

// (branch-free; `cold` is set where an argument is zero, infinite or a NaN — or x negative — and the value returned is not powf's)
template <class Tab = GlobalTables>
SRK_DEV float powf_libm_plain(float x, float y, bool& cold, const Tab tab = Tab{})
{
    uint32_t ix = __float_as_uint(x);
    const uint32_t iy = __float_as_uint(y);
    cold = cold || !(ix - 1u < 0x7f7fffffu) || !(2u * iy - 1u < 2u * 0x7f800000u - 1u);   // plain: 0 < x < inf, y finite and not zero
    // a subnormal x: normalise (`ix = asuint(x * 0x1p23f) & 0x7fffffff; ix -= 23 << 23`)
    const uint32_t in = (__float_as_uint(x * 0x1p23f) & 0x7fffffffu) - (23u << 23);
    ix = ix < 0x00800000u ? in : ix;
    // log2_inline
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const uint32_t top = tmp & 0xff800000u;
    const int k = (int)top >> 23;   // arithmetic shift
    const double invc = __longlong_as_double((long long)tab.log2_inv_c(i)), logc = __longlong_as_double((long long)tab.log2_c(i));
    const double z = (double)__uint_as_float(ix - top);
    const double r = __builtin_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double r2 = r * r;
    const double yy = __builtin_fma(r, 0x1.27616c9496e0bp-2, -0x1.71969a075c67ap-2);
    const double p = __builtin_fma(r, 0x1.ec70a6ca7baddp-2, -0x1.7154748bef6c8p-1);
    const double r4 = r2 * r2;
    double q = __builtin_fma(r, 0x1.71547652ab82bp+0, y0);
    q = __builtin_fma(r2, p, q);
    const double logx = __builtin_fma(yy, r4, q);
    const double ylogx = (double)y * logx;   // (cannot overflow: y is single precision)
    // exp2_inline (sign_bias 0)
    const double shift = 0x1.8p+52 / 32.0;
    double kd = ylogx + shift;
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= shift;
    const double re = ylogx - kd;
    const uint64_t t = tab.exp2f((uint32_t)ki & 31u) + (ki << 47);
    const double sc = __longlong_as_double((long long)t);
    const double zz = __builtin_fma(re, 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3);
    const double re2 = re * re;
    double v = __builtin_fma(re, 0x1.62e42ff0c52d6p-1, 1.0);
    v = __builtin_fma(zz, re2, v);
    float out = (float)(v * sc);              // subnormal results round here, as in the libm
    out = ylogx <= -150.0 ? 0.0f : out;       // __math_uflowf
    out = ylogx > 0x1.fffffffd1d571p+6 ? __builtin_inff() : out;   // __math_oflowf
    return out;
}
// (ocml's powf behind the cold branch, out of line: inlined, its few hundred instructions sat in every sample's straight-line code)
__device__ __attribute__((noinline)) float powf_cold(float x, float b) { return ::powf(x, b); }
template <class Tab = GlobalTables>
SRK_DEV float powf_pos(float x, float b, bool /* exact: every flavour but NONLIN_LOOSE is the libm's own since round 6 */, const Tab tab = Tab{})
{
    bool cold = false;
    float r = powf_libm_plain(x, b, cold, tab);
    if (__builtin_amdgcn_ballot_w64(cold)) {
        if (cold) r = powf_cold(x, b);
    }
    return r;
}

// NONLIN_LOOSE (the flattener: nothing integrates or thresholds this module's output): x^b = 2^(b log2 x) through v_log_f32 / v_exp_f32 —
// one ulp each, i.e. a relative error of about 1.3e-7 |b log2 x|; taken while |b log2 x| < 32 (4e-6), everything else as powf_pos.
template <class Tab = GlobalTables>
SRK_DEV float powf_pos_loose(float x, float b, const Tab tab = Tab{})
{
    const float y = b * __builtin_amdgcn_logf(x >= 0x1p-126f ? x : 1.0f);
    const bool fast = x >= 0x1p-126f && x < __builtin_inff() && __builtin_fabsf(y) < 32.0f;  // NaN x / b / y: false
    float r = __builtin_amdgcn_exp2f(fast ? y : 0.0f);
    if (__builtin_amdgcn_ballot_w64(!fast)) {
        if (!fast) r = powf_pos(x, b, false, tab);
    }
    return r;
}

template <class Tab = GlobalTables>
SRK_DEV float nonlin_step(uint32_t flags, float in1, float in2, float constant, const Tab tab = Tab{})
{
    const float a = (flags & MATH_HAS_IN1) ? in1 : zero_f32();
    const float b = (flags & MATH_HAS_IN2) ? in2 : constant;
    const bool pos = a > 0.0f;
    const float r = (flags & NONLIN_LOOSE) ? powf_pos_loose(pos ? a : -a, b, tab) : powf_pos(pos ? a : -a, b, (flags & NONLIN_EXACT) != 0, tab);
    return pos ? r : -r;
}

// ---------------------------------------------------------------------------------------------
// SampleModule — sample.rs:192-240
// ---------------------------------------------------------------------------------------------
// `2.0_f32.powf(cv)` bit for bit as glibc computes it (Rust's f32::powf is libm's powf).  For base 2 powf's
// log2 step is exact (its table entry for 1.0 is {1, 0}: log2(2) = 1 with no rounding), so powf(2, y) is its
// exp2 kernel applied to f64(y): y = k/32 + r, 2^(k/32) from a 32-entry table of correctly rounded values
// (tab[i] = bits(2^(i/32)) - (i << 47)), 2^r by the cubic below, one rounding to f32.  Checked against glibc
// 2.35's powf on 2e8 random arguments in [-20, 20) with zero mismatches (tests/test_oracle.py holds the
// CPU-side check of this same formula); the read position it scales is an INDEX, hence bit-exactness.

template <class Tab = GlobalTables>
SRK_DEV float pow2f_libm(float y, const Tab tab = Tab{})
{
    // (branch-free since round 6: the three special answers are selected at the end — as early returns they were three exec-mask branches per
    // sample of the sample player's position recurrence; the arithmetic in between has no traps, whatever it makes of a NaN or 1e30)
    const double xd = (double)y;
    const double shift = 0x1.8p+52 / 32.0;
    double kd = xd + shift;                       // round to a multiple of 1/32
    const uint64_t ki = (uint64_t)__double_as_longlong(kd);
    kd -= shift;
    const double r = xd - kd;
    const uint64_t t = tab.exp2f((uint32_t)ki & 31u) + (ki << 47);
    const double s = __longlong_as_double((long long)t);
    const double z = 0x1.c6af84b912394p-5 * r + 0x1.ebfce50fac4f3p-3;
    const double r2 = r * r;
    double p = 0x1.62e42ff0c52d6p-1 * r + 1.0;
    p = z * r2 + p;
    p = p * s;
    float out = (float)p;                         // subnormal results round here, as in libm
    out = y <= -150.0f ? 0.0f : out;              // underflow (covers -inf)
    out = y >= 128.0f ? __builtin_inff() : out;   // f64(y) > 0x1.fffffffd1d571p+6: overflow (covers +inf)
    return y != y ? y + 2.0f : out;               // NaN
}

// ---------------------------------------------------------------------------------------------
// Sequencers — GridSequencerModule::calc sequencer.rs:190-246, PatternSequencerModule::calc :482-533
// ---------------------------------------------------------------------------------------------
struct SeqRegs {
    uint32_t current_step;
    bool step_last, sync_last;
};

SRK_DEV uint32_t seq_advance(SeqRegs& s, float step_in, float sync_in, uint32_t length)
{
    if (rising_edge(s.step_last, step_in)) s.current_step = (s.current_step + 1u) & 0xffffu;  // u16 in the reference
    if (rising_edge(s.sync_last, sync_in)) s.current_step = 0u;
    uint32_t cs = s.current_step;
    if (cs >= length) {
        s.current_step = 0u;
        cs = 0u;
    }
    return cs;
}

// a grid cell: bit 31 = Some, bit 30 = hold, low 16 bits = the note value; `last` = the CV held through rests
SRK_DEV void gridseq_outputs(uint32_t cell, uint32_t cs, float step_in, float inv_spo, float& last, float& cv, float& gate, float& sync)
{
    const bool present = cell & 0x80000000u, hold = cell & 0x40000000u;
    cv = present ? (float)(cell & 0xffffu) * inv_spo : last;
    gate = present ? (hold ? 1.0f : step_in) : 0.0f;
    sync = cs == 0u ? 1.0f : 0.0f;
    last = cv;
}

// a pattern cell: two bits per channel (bit 0 = Some, bit 1 = held)
SRK_DEV float patseq_gate(uint32_t cell, int channel, float step_in)
{
    const uint32_t b = (cell >> (2 * (channel & 7))) & 3u;
    return (b & 1u) ? ((b & 2u) ? 1.0f : step_in) : 0.0f;
}

struct SmpRegs {
    float pos;
    bool playing, gate_last;
};

// One sample of the position state machine; returns the index read this sample.  `ratio` =
// wavebox.sample_rate / self.sample_rate (f32 divide, loop-invariant).  `as usize` saturates and maps NaN to 0;
// the clamp below does the same within u32 (wave lengths are < 2^31, so any index >= 2^32 is out of range anyway).
template <class Tab = GlobalTables>
SRK_DEV uint32_t sample_advance(uint32_t flags, SmpRegs& s, float ratio, uint32_t n_wave, float gate, float cv, const Tab tab = Tab{})
{
    if (rising_edge(s.gate_last, (flags & SMP_HAS_GATE) ? gate : 0.0f)) {
        s.pos = 0.0f;
        s.playing = true;
    }
    uint32_t idx = (uint32_t)__builtin_fminf(__builtin_fmaxf(s.pos, 0.0f), 4294967040.0f);
    if (idx >= n_wave) {
        s.pos = 0.0f;
        s.playing = false;
        idx = 0u;
    }
    // (the step is computed in every lane and selected: behind `if (playing)` it was an exec-mask branch around 2^cv in every sample of the
    // position's recurrence — for a player that is playing nearly always.  Measured, one box, three rounds: P4 18.65 -> 17.72 ms per step; a
    // wave-uniform `if (any lane playing)` instead keeps the patches whose players are never triggered at their old speed (the survey's seeds 1
    // and 4: 124 / 123 ms per second of audio against 134 / 150) and costs P4 the same 5 % as the branch it replaces — with 64 voices on
    // their own clocks some lane is nearly always playing, so the select is what a polyphonic render wants)
    const float step = (flags & SMP_HAS_CV) ? ratio * pow2f_libm(cv, tab) : ratio * 1.0f;
    const float next = s.pos + step;
    s.pos = s.playing ? next : s.pos;
    return idx;
}

}  // namespace dev
}  // namespace srack
