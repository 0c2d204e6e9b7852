// interp.hip.h — the generic tile interpreter: executes a flattened op list tile by tile.
//
// Per tile and per op, one module-type device function (`tile_*`) runs `tile` samples with the module's state in
// VGPRs; wires between ops are [tile][64] f32 tiles in LDS; the voice table (state + per-voice parameters) and the
// tile's slices of the control tracks sit in LDS.  HBM is touched for: the voice table (once in, once out per launch),
// rendered frames (coalesced 256 B per wave-store), mix partials, control tracks, the rings of broken feedback edges
// when buffer_size > 16, and SampleModule's wave (gathered).
// Kernels: render_interp<exact> (one launch = all voices of a voice program, or the single unit of a control program),
// render_interp_stages<exact> (the control pipeline: one block per control unit).
#pragma once
#include <hip/hip_runtime.h>

#include "kernel_args.hip.h"
#include "modules.hip.h"
#include "wave.hip.h"

namespace srack {

namespace dev {

// LDS pointers carry their address space: a plain float* inside a struct handed to a noinline function degrades every
// access to flat_load / flat_store (measured: 25 VMEM instructions and 54 % wait cycles per voice-sample).
// The op list is read-only, wave-uniform data: seen through the constant address space its fields arrive by scalar
// loads (s_load_dword*) instead of flat loads on the vector memory path.
typedef const __attribute__((address_space(4))) DevOp COp;
typedef const __attribute__((address_space(4))) KernelArgs CArgs;  // the kernel's own argument block, read in place (kernarg segment)
// Arguments of a non-kernel function travel in VGPRs, so the compiler no longer knows the op pointer is the same in
// every lane and would fetch each field with a vector load.  readfirstlane makes the uniformity explicit again.
SRK_DEV COp& uniform_op(COp& op)
{
    const uint64_t p = (uint64_t)&op;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return *(COp*)(((uint64_t)hi << 32) | lo);
}
SRK_DEV CArgs& uniform_args(CArgs& a)
{
    const uint64_t p = (uint64_t)&a;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)p), hi = __builtin_amdgcn_readfirstlane((uint32_t)(p >> 32));
    return *(CArgs*)(((uint64_t)hi << 32) | lo);
}
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

struct Ctx {            // what every tile function sees
    lds_u32* rows;      // LDS [n_rows][64]
    lds_f32* wires;     // LDS [n_slots][tile][64]
    lds_f32* zero;      // LDS row of zeros: what an unconnected input reads (stride 0)
    lds_f32* trash;     // LDS row nobody reads: where an unread output goes (stride 0)
    lds_f32* trk;       // LDS [n_tracks][64]: this tile's samples of every control track (same for all lanes)
    int tile, n, lane;  // tile capacity, samples in this tile, lane
};

// Arguments of a non-inlined device function travel in VGPRs, so the compiler must assume they differ per lane: loops
// over c.n become exec-masked loops and every address sum a vector add.  Everything in Ctx but `lane` IS wave-uniform;
// saying so (v_readfirstlane) moves loop control and address arithmetic to the scalar unit.
template <class P>
SRK_DEV P uniform_lds(P p)
{
    return (P)(uintptr_t)__builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)p);  // an LDS address is 32 bits
}
SRK_DEV Ctx uniform_ctx(const Ctx& v)
{
    Ctx c;
    c.rows = uniform_lds(v.rows);
    c.wires = uniform_lds(v.wires);
    c.zero = uniform_lds(v.zero);
    c.trash = uniform_lds(v.trash);
    c.trk = uniform_lds(v.trk);
    c.tile = __builtin_amdgcn_readfirstlane(v.tile);
    c.n = __builtin_amdgcn_readfirstlane(v.n);
    c.lane = v.lane;
    return c;
}

#define ROW(r) c.rows[(r) * 64 + c.lane]
#define WIRE(slot, i) c.wires[((slot) * c.tile + (i)) * 64 + c.lane]

SRK_DEV float par(const Ctx& c, COp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(ROW(op.par_row[k])) : op.par_val[k]; }

// A port as (lane pointer, stride in floats per sample).  Unconnected inputs read the zero row, unread outputs
// write the trash row, both with stride 0 — so tile loops carry no per-sample "is it wired" branches.
struct Port {
    lds_f32* p;
    int stride;
};
SRK_DEV Port in_port(const Ctx& c, int slot)
{
    if (slot >= kTrackSlot) return Port{c.trk + (slot - kTrackSlot) * 64, 1};  // a control track: same address in every lane (LDS broadcast)
    return slot >= 0 ? Port{c.wires + slot * c.tile * 64 + c.lane, 64} : Port{c.zero + c.lane, 0};
}
SRK_DEV Port out_port(const Ctx& c, int slot) { return slot >= 0 ? Port{c.wires + slot * c.tile * 64 + c.lane, 64} : Port{c.trash + c.lane, 0}; }

// Runs step(x[NI], y[NO]) for every sample of the tile, kU samples at a time: the kU x NI input reads are issued
// together, then the kU steps, then the kU x NO writes — one LDS round trip per kU samples instead of per sample.
// An output may share its slot with an input of the same op (flatten.cpp reuses the slot of an input that dies here):
// that is safe because a group's inputs are all read before any of its outputs is written, and sample i only lives at row i.
template <int NI, int NO, class Step>
SRK_DEV void tile_run(const Ctx& c, const Port (&in)[NI], const Port (&out)[NO], Step step)
{
    constexpr int kU = 4;
    int i = 0;
    for (; i + kU <= c.n; i += kU) {
        float x[kU][NI], y[kU][NO];
#pragma unroll
        for (int u = 0; u < kU; u++)
#pragma unroll
            for (int k = 0; k < NI; k++) x[u][k] = in[k].p[(i + u) * in[k].stride];
#pragma unroll
        for (int u = 0; u < kU; u++) step(x[u], y[u]);
#pragma unroll
        for (int u = 0; u < kU; u++)
#pragma unroll
            for (int k = 0; k < NO; k++) out[k].p[(i + u) * out[k].stride] = y[u][k];
    }
    for (; i < c.n; i++) {
        float x[NI], y[NO];
#pragma unroll
        for (int k = 0; k < NI; k++) x[k] = in[k].p[i * in[k].stride];
        step(x, y);
#pragma unroll
        for (int k = 0; k < NO; k++) out[k].p[i * out[k].stride] = y[k];
    }
}

// ---- one tile of one module type -----------------------------------------------------------------
// Every tile function is a template on the kernel flavour, also where the code does not depend on it: the register
// budget a kernel asks for (amdgpu_waves_per_eu, see render_interp) only reaches callees that no other kernel shares.

// The oscillator has three tile functions, one per regime, so that the f64 sine / pow of the general one does not cost
// the two carried-phase ones registers (under the 96-VGPR budget the single function spilled 112 bytes).
struct OscSetup {
    OscRegs s;
    OscConst k;
};
SRK_DEV OscSetup osc_setup(const Ctx& c, COp& op)
{
    OscSetup u;
    const int sr = op.state_row;
    u.s.pos = make_f64(ROW(sr + OSC_S_POS_LO), ROW(sr + OSC_S_POS_HI));
    u.s.sync_last = ROW(sr + OSC_S_SYNC_LAST) != 0;
    u.k.sr = op.sample_rate;
    u.k.val = (double)par(c, op, OSC_P_VAL);
    u.k.delta = op.delta_row >= 0 ? make_f64(ROW(op.delta_row), ROW(op.delta_row + 1)) : op.delta;
    u.k.inv_dt = inv_dt_f32(u.k.delta);
    return u;
}
SRK_DEV void osc_store(const Ctx& c, COp& op, double pos, bool sync_last)
{
    const int sr = op.state_row;
    ROW(sr + OSC_S_POS_LO) = f64_lo(pos);
    ROW(sr + OSC_S_POS_HI) = f64_hi(pos);
    ROW(sr + OSC_S_SYNC_LAST) = sync_last ? 1u : 0u;
}

// no CV, no sync, one live port, delta < 0.25 for every voice (host-checked): the carried-phase oscillator
template <bool kExact>
__device__ __noinline__ void tile_osc_const(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const uint32_t fl = op.flags;
    const Port w[1] = {out_port(c, op.out_slot[(fl & OSC_OUT_SAW) ? 2 : (fl & OSC_OUT_SQUARE) ? 1 : 0])};
    const Port none[1] = {w[0]};  // no input: the dummy read goes to the op's own output row
    if (fl & OSC_FIXED_PHASE) {  // (host-set: default mode, saw only) phase and increment rows hold value * 2^64 (modules.hip.h, FOsc)
        const int sr = op.state_row;
        const uint64_t dbits = (uint64_t)__double_as_longlong(op.delta);
        FOsc fo;
        fosc_init(fo, ROW(sr + OSC_S_POS_LO), ROW(sr + OSC_S_POS_HI), op.delta_row >= 0 ? ROW(op.delta_row) : (uint32_t)dbits,
                  op.delta_row >= 0 ? ROW(op.delta_row + 1) : (uint32_t)(dbits >> 32));
        tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = fosc_saw(fo); });
        ROW(sr + OSC_S_POS_LO) = fo.lo;
        ROW(sr + OSC_S_POS_HI) = fo.hi;
        ROW(sr + OSC_S_SYNC_LAST) = 0u;
        return;
    }
    const OscSetup u = osc_setup(c, op);
    COsc o;
    cosc_init(o, u.s.pos, u.k.delta);
    if (fl & OSC_OUT_SAW)
        tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = cosc_saw(o); });
    else if (fl & OSC_OUT_SQUARE)
        tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = cosc_square(o); });
    else
        tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = cosc_sine(o); });
    osc_store(c, op, o.pos, false);  // sync unconnected: `last` follows the constant 0.0 input
}

// A sequencer-driven pitch: the carried-phase oscillator between note changes (modules.hip.h, StepOsc).
template <bool kExact>
__device__ __noinline__ void tile_osc_stepwise(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const uint32_t fl = op.flags;
    const OscSetup u = osc_setup(c, op);
    const OscConst k = u.k;
    StepOsc so;
    steposc_init(so, u.s.pos);
    const Port cvp[1] = {in_port(c, op.in_slot[0])};
    const Port w[1] = {out_port(c, op.out_slot[(fl & OSC_OUT_SAW) ? 2 : (fl & OSC_OUT_SQUARE) ? 1 : 0])};
    if (fl & OSC_OUT_SAW)
        tile_run<1, 1>(c, cvp, w, [&](const float* x, float* y) { y[0] = steposc_step<OSC_OUT_SAW>(so, k, x[0]); });
    else if (fl & OSC_OUT_SQUARE)
        tile_run<1, 1>(c, cvp, w, [&](const float* x, float* y) { y[0] = steposc_step<OSC_OUT_SQUARE>(so, k, x[0]); });
    else
        tile_run<1, 1>(c, cvp, w, [&](const float* x, float* y) { y[0] = steposc_step<OSC_OUT_SINE>(so, k, x[0]); });
    osc_store(c, op, so.o.pos, false);
}

// everything else: CV at audio rate, sync, several live ports, no anti-aliasing, the exact flavour
template <bool kExact>
__device__ __noinline__ void tile_osc_general(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const uint32_t fl = op.flags;
    OscSetup u = osc_setup(c, op);
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[3] = {out_port(c, op.out_slot[0]), out_port(c, op.out_slot[1]), out_port(c, op.out_slot[2])};
    const uint32_t f = kExact ? (fl | OSC_EXACT) : fl;  // (default flavour: single oscillators may still be exact — the flattener's OSC_EXACT on that op)
    tile_run<2, 3>(c, in, out, [&](const float* x, float* y) {
        y[0] = y[1] = y[2] = 0.0f;
        osc_step(f, u.s, u.k, x[0], x[1], y[0], y[1], y[2]);
    });
    osc_store(c, op, u.s.pos, u.s.sync_last);
}

template <bool kExact>
SRK_DEV void tile_osc(const Ctx& c, COp& op)
{
    const uint32_t fl = op.flags;  // wave-uniform (scalar load)
    const uint32_t ports = fl & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
    if (fl & OSC_CONST_FAST)
        tile_osc_const<kExact>(c, op);
    else if (!kExact && (fl & (OSC_HAS_CV | OSC_CV_STEPWISE | OSC_HAS_SYNC | OSC_AA | OSC_EXACT_BLEP | OSC_EXACT)) == (OSC_HAS_CV | OSC_CV_STEPWISE | OSC_AA) && ports && !(ports & (ports - 1)))
        tile_osc_stepwise<kExact>(c, op);
    else
        tile_osc_general<kExact>(c, op);
}

SRK_DEV void vcf_load(const Ctx& c, int sr, VcfRegs& s)
{
    s.f = __uint_as_float(ROW(sr + VCF_S_F));
    s.p = __uint_as_float(ROW(sr + VCF_S_P));
    s.q = __uint_as_float(ROW(sr + VCF_S_Q));
    s.b0 = __uint_as_float(ROW(sr + VCF_S_B0 + 0));
    s.b1 = __uint_as_float(ROW(sr + VCF_S_B0 + 1));
    s.b2 = __uint_as_float(ROW(sr + VCF_S_B0 + 2));
    s.b3 = __uint_as_float(ROW(sr + VCF_S_B0 + 3));
    s.b4 = __uint_as_float(ROW(sr + VCF_S_B0 + 4));
    s.freq = __uint_as_float(ROW(sr + VCF_S_FREQ));
    s.res = __uint_as_float(ROW(sr + VCF_S_RES));
}

SRK_DEV void vcf_store(const Ctx& c, int sr, const VcfRegs& s)
{
    ROW(sr + VCF_S_F) = __float_as_uint(s.f);
    ROW(sr + VCF_S_P) = __float_as_uint(s.p);
    ROW(sr + VCF_S_Q) = __float_as_uint(s.q);
    ROW(sr + VCF_S_B0 + 0) = __float_as_uint(s.b0);
    ROW(sr + VCF_S_B0 + 1) = __float_as_uint(s.b1);
    ROW(sr + VCF_S_B0 + 2) = __float_as_uint(s.b2);
    ROW(sr + VCF_S_B0 + 3) = __float_as_uint(s.b3);
    ROW(sr + VCF_S_B0 + 4) = __float_as_uint(s.b4);
    ROW(sr + VCF_S_FREQ) = __float_as_uint(s.freq);
    ROW(sr + VCF_S_RES) = __float_as_uint(s.res);
}

// kFast: the default mode's contracted ladder; otherwise the literal one (exact mode, or VCF_LITERAL: an output reaches a pitch input)
template <bool kFast>
SRK_DEV void tile_vcf_body(const Ctx& c, COp& op)
{
    const uint32_t fl = op.flags;
    VcfRegs s;
    vcf_load(c, op.state_row, s);
    bool fin = kFast || vcf_nan_free(s);
    const float freq = par(c, op, VCF_P_FREQ), exp_amt = par(c, op, VCF_P_EXP);
    const float res = vcf_resonance(par(c, op, VCF_P_RES));
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[3] = {out_port(c, op.out_slot[0]), out_port(c, op.out_slot[1]), out_port(c, op.out_slot[2])};
    const uint32_t ports = fl & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP);
    const bool one = ports && !(ports & (ports - 1));  // the usual case: one port read => one wire written, not three
    const Port w[1] = {out[ports == VCF_OUT_HP ? 2 : ports == VCF_OUT_BP ? 1 : 0]};
    auto pick = [&](float lp, float bp, float hp) { return ports == VCF_OUT_LP ? lp : (ports == VCF_OUT_BP ? bp : hp); };
    if (fl & VCF_HAS_CV) {
        if (one)
            tile_run<2, 1>(c, in, w, [&](const float* x, float* y) {
                float lp, bp, hp;
                vcf_coeffs<kFast>(s, vcf_frequency(freq, x[1], exp_amt), res);
                vcf_run<kFast>(s, fin, x[0], lp, bp, hp);
                y[0] = pick(lp, bp, hp);
            });
        else
            tile_run<2, 3>(c, in, out, [&](const float* x, float* y) {
                vcf_coeffs<kFast>(s, vcf_frequency(freq, x[1], exp_amt), res);
                vcf_run<kFast>(s, fin, x[0], y[0], y[1], y[2]);
            });
    } else {
        // constant cutoff: the "did (frequency, res) change" check can only fire on the first sample
        vcf_coeffs<kFast>(s, vcf_frequency(freq, 0.0f, exp_amt), res);
        const Port audio[1] = {in[0]};
        if (one)
            tile_run<1, 1>(c, audio, w, [&](const float* x, float* y) {
                float lp, bp, hp;
                vcf_run<kFast>(s, fin, x[0], lp, bp, hp);
                y[0] = pick(lp, bp, hp);
            });
        else
            tile_run<1, 3>(c, audio, out, [&](const float* x, float* y) { vcf_run<kFast>(s, fin, x[0], y[0], y[1], y[2]); });
    }
    vcf_store(c, op.state_row, s);
}

template <bool kExact>
__device__ __noinline__ void tile_vcf(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    if (kExact || (op.flags & VCF_LITERAL))
        tile_vcf_body<false>(c, op);
    else
        tile_vcf_body<true>(c, op);
}

template <bool kExact>
__device__ __noinline__ void tile_adsr(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const int sr = op.state_row;
    AdsrRegs s;
    s.phase = __uint_as_float(ROW(sr + ADSR_S_PHASE));
    s.mode = (int)ROW(sr + ADSR_S_MODE);
    s.r_val = __uint_as_float(ROW(sr + ADSR_S_R_VAL));
    s.from_a_val = __uint_as_float(ROW(sr + ADSR_S_FROM_A));
    s.gate_last = ROW(sr + ADSR_S_GATE_LAST) != 0;
    const AdsrConst k = adsr_consts(par(c, op, ADSR_P_A), par(c, op, ADSR_P_D), par(c, op, ADSR_P_S), par(c, op, ADSR_P_R), par(c, op, ADSR_P_SR));
    const Port in[1] = {in_port(c, op.in_slot[0])};
    const Port out[1] = {out_port(c, op.out_slot[0])};
    if (op.flags & ADSR_HAS_GATE) {
        AdsrSeg g;
        adsr_seg_enter(s, k, g);
        tile_run<1, 1>(c, in, out, [&](const float* x, float* y) { y[0] = adsr_seg_step(s, k, g, x[0]); });
        adsr_seg_flush(s, g);
    } else {
        tile_run<1, 1>(c, in, out, [&](const float*, float* y) { y[0] = adsr_step(op.flags, s, k, 0.0f); });
    }
    ROW(sr + ADSR_S_PHASE) = __float_as_uint(s.phase);
    ROW(sr + ADSR_S_MODE) = (uint32_t)s.mode;
    ROW(sr + ADSR_S_R_VAL) = __float_as_uint(s.r_val);
    ROW(sr + ADSR_S_FROM_A) = __float_as_uint(s.from_a_val);
    ROW(sr + ADSR_S_GATE_LAST) = s.gate_last ? 1u : 0u;
}

template <bool kExact>
__device__ __noinline__ void tile_vca(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const bool negative = par(c, op, VCA_P_NEG) != 0.0f;
    const uint32_t fl = op.flags;
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[1] = {out_port(c, op.out_slot[0])};
    tile_run<2, 1>(c, in, out, [&](const float* x, float* y) { y[0] = vca_step(fl, negative, x[0], x[1]); });
}

template <bool kExact>
__device__ __noinline__ void tile_mix(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    float gain[4];
    for (int k = 0; k < 4; k++) gain[k] = par(c, op, MIX_P_GAIN0 + k);
    const uint32_t fl = op.flags;
    const Port in[4] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1]), in_port(c, op.in_slot[2]), in_port(c, op.in_slot[3])};
    const Port out[1] = {out_port(c, op.out_slot[0])};
    tile_run<4, 1>(c, in, out, [&](const float* x, float* y) { y[0] = mixer_step(fl, x, gain); });
}

template <bool kExact>
__device__ __noinline__ void tile_math(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const float constant = par(c, op, MATH_P_CONST);
    const uint32_t fl = op.flags;
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[1] = {out_port(c, op.out_slot[0])};
    tile_run<2, 1>(c, in, out, [&](const float* x, float* y) { y[0] = math_step(fl, x[0], x[1], constant); });
}

template <bool kExact>
__device__ __noinline__ void tile_nonlin(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const float constant = par(c, op, NONLIN_P_CONST);
    const uint32_t fl = op.flags;
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[1] = {out_port(c, op.out_slot[0])};
    tile_run<2, 1>(c, in, out, [&](const float* x, float* y) { y[0] = nonlin_step(fl, x[0], x[1], constant); });
}

// FreeverbModule (freeverb.rs:208-270) around the freeverb crate's Freeverb::tick, restated (PARITY UNPINNED: the crate is
// not vendored in the reference tree; oracle/srack_oracle.c states what the restatement rests on).  All arithmetic in f64,
// one operation per rounding, as the crate's:
//   input_mixed = (l + r) * 0.015 * input_gain
//   comb:    out = line.read(); state = out * (1 - damp) + state * damp; line.write(input_mixed + state * feedback)
//   allpass: d = line.read(); out = -in + d; line.write(in + d * 0.5)
//   l' = out.0 * wet_gains.0 + out.1 * wet_gains.1 + l * dry   (and mirrored for r')
// The 24 delay lines live in HBM, one double per voice per slot (voice-minor: a wave reads 512 contiguous bytes per slot).
// Every line advances one slot per sample from slot 0 at sample 0, so its position is n mod length — no per-voice index.
// The combs are parallel and a line is read `length` >= 4 samples after it was written, so a group of four samples of one
// line is four independent loads, then the four recurrence steps, then four stores: line by line, group by group.
template <bool kExact>
__device__ __noinline__ void tile_freeverb(const Ctx c_v, COp& op_v, CArgs& a_v, uint64_t n_abs, uint32_t voice, uint32_t voice_c, bool active)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    CArgs& a = uniform_args(a_v);
    typedef const __attribute__((address_space(4))) uint32_t CU32;
    CU32* tab = (CU32*)(uintptr_t)(a.seqtab + op.aux);
    // lane j < 24 holds line j's length, first row and the slot of the tile's first sample
    const int jl = c.lane < kFvLines ? c.lane : 0;
    const uint32_t len_l = a.seqtab[op.aux + jl], first_l = a.seqtab[op.aux + kFvLines + jl];
    const uint32_t idx_l = (uint32_t)(n_abs % (uint64_t)len_l);
    auto dbl = [&](int k) { return __hiloint2double((int)tab[2 * kFvLines + 2 * k + 1], (int)tab[2 * kFvLines + 2 * k]); };
    const double feedback = dbl(0), damp = dbl(1), damp_inv = dbl(2), wet0 = dbl(3), wet1 = dbl(4), dry = dbl(5), gain = dbl(6);
    const size_t V = a.V;
    double* blk = a.fv + (size_t)op.delta_row * V;
    const double* ld = blk + voice_c;
    double* st = blk + voice;
    double fs[kFvStates];
#pragma unroll
    for (int k = 0; k < kFvStates; k++) fs[k] = ld[(size_t)k * V];
    const Port in_l = in_port(c, op.in_slot[0]), in_r = in_port(c, op.in_slot[1]);
    const Port out_l = out_port(c, op.out_slot[0]), out_r = out_port(c, op.out_slot[1]);
    constexpr int G = 4;
    for (int s0 = 0; s0 < c.n; s0 += G) {
        const int m = min(G, c.n - s0);
        float l[G], r[G];
        double x[G], o0[G], o1[G];
#pragma unroll
        for (int i = 0; i < G; i++) {
            const int s = s0 + min(i, m - 1);  // (past the end of a short last group: re-read its last sample; the result is dropped)
            l[i] = in_l.p[s * in_l.stride];
            r[i] = in_r.p[s * in_r.stride];
            x[i] = ((double)l[i] + (double)r[i]) * 0.015 * gain;
            o0[i] = 0.0;
            o1[i] = 0.0;
        }
        // A unit's left and right lines over the group: both lines' four slots are loaded before either recurrence runs (one
        // memory round trip per pair of lines instead of one per line — the tile function is latency-bound).
        struct Slots {
            size_t row0;
            uint32_t p[G];
        };
        auto locate = [&](int j) {
            const uint32_t len = (uint32_t)__builtin_amdgcn_readlane((int)len_l, j), first = (uint32_t)__builtin_amdgcn_readlane((int)first_l, j);
            uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)idx_l, j) + (uint32_t)s0;
            while (pos >= len) pos -= len;
            Slots q;
            q.row0 = (size_t)kFvStates + first;
#pragma unroll
            for (int i = 0; i < G; i++) {
                q.p[i] = pos + (uint32_t)i;
                if (q.p[i] >= len) q.p[i] -= len;
            }
            return q;
        };
        auto unit = [&](int j, auto step_l, auto step_r) {
            const Slots ql = locate(j), qr = locate(j + 1);
            double rl[G], rr[G];
#pragma unroll
            for (int i = 0; i < G; i++) {
                rl[i] = ld[(ql.row0 + ql.p[i]) * V];
                rr[i] = ld[(qr.row0 + qr.p[i]) * V];
            }
#pragma unroll
            for (int i = 0; i < G; i++)
                if (i < m) {
                    const double wl = step_l(i, rl[i]), wr = step_r(i, rr[i]);
                    if (active) {
                        st[(ql.row0 + ql.p[i]) * V] = wl;
                        st[(qr.row0 + qr.p[i]) * V] = wr;
                    }
                }
        };
#pragma unroll
        for (int k = 0; k < 8; k++)
            unit(2 * k,
                 [&](int i, double out) {
                     fs[2 * k] = out * damp_inv + fs[2 * k] * damp;
                     o0[i] += out;
                     return x[i] + fs[2 * k] * feedback;
                 },
                 [&](int i, double out) {
                     fs[2 * k + 1] = out * damp_inv + fs[2 * k + 1] * damp;
                     o1[i] += out;
                     return x[i] + fs[2 * k + 1] * feedback;
                 });
#pragma unroll
        for (int k = 0; k < 4; k++)
            unit(16 + 2 * k,
                 [&](int i, double delayed) {
                     const double in = o0[i];
                     o0[i] = -in + delayed;
                     return in + delayed * 0.5;
                 },
                 [&](int i, double delayed) {
                     const double in = o1[i];
                     o1[i] = -in + delayed;
                     return in + delayed * 0.5;
                 });
#pragma unroll
        for (int i = 0; i < G; i++)
            if (i < m) {
                out_l.p[(s0 + i) * out_l.stride] = (float)(o0[i] * wet0 + o1[i] * wet1 + (double)l[i] * dry);
                out_r.p[(s0 + i) * out_r.stride] = (float)(o1[i] * wet0 + o0[i] * wet1 + (double)r[i] * dry);
            }
    }
    if (active)
#pragma unroll
        for (int k = 0; k < kFvStates; k++) st[(size_t)k * V] = fs[k];
}

// NoiseModule: stateless — the tile's samples are n_abs .. n_abs + c.n - 1 of this voice's stream.
template <bool kExact>
__device__ __noinline__ void tile_noise(const Ctx c_v, COp& op_v, uint64_t n_abs, uint32_t voice_c)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const uint64_t base = (uint64_t)__double_as_longlong(op.delta), first = (uint64_t)__double_as_longlong(op.sample_rate);
    const uint64_t key = noise_voice_key(base, first + voice_c);
    const Port out = out_port(c, op.out_slot[0]);
    for (int i = 0; i < c.n; i++) out.p[i * out.stride] = noise_sample(key, n_abs + (uint64_t)i);
}

// SampleModule (sample.rs:192-240) in two passes over the tile: the position state machine does not depend on the
// samples it reads, so pass 1 leaves each sample's read INDEX in the output wire and pass 2 turns indices into
// samples with independent gathers from the shared wave (8 loads in flight per lane instead of one per step).
template <bool kExact>
__device__ __noinline__ void tile_sample(const Ctx c_v, COp& op_v, CArgs& a_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    CArgs& a = uniform_args(a_v);
    const int sr = op.state_row;
    const uint32_t fl = op.flags;
    SmpRegs s;
    s.pos = __uint_as_float(ROW(sr + SMP_S_POS));
    s.playing = ROW(sr + SMP_S_PLAYING) != 0;
    s.gate_last = ROW(sr + SMP_S_GATE_LAST) != 0;
    const float ratio = par(c, op, SMP_P_WAVE_SR) / par(c, op, SMP_P_SR);
    const uint32_t n_wave = (uint32_t)op.seq_len;
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[1] = {out_port(c, op.out_slot[0])};
    tile_run<2, 1>(c, in, out, [&](const float* x, float* y) { y[0] = __uint_as_float(sample_advance(fl, s, ratio, n_wave, x[0], x[1])); });
    ROW(sr + SMP_S_POS) = __float_as_uint(s.pos);
    ROW(sr + SMP_S_PLAYING) = s.playing ? 1u : 0u;
    ROW(sr + SMP_S_GATE_LAST) = s.gate_last ? 1u : 0u;
    if (op.out_slot[0] < 0) return;
    const uint32_t* wave = a.seqtab + op.aux;
    const Port w = out[0];
    int i = 0;
    for (; i + 8 <= c.n; i += 8) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = __float_as_uint(w.p[(i + u) * w.stride]);
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = n_wave ? wave[v[u]] : 0u;  // empty wave: `*out = 0.0`
#pragma unroll
        for (int u = 0; u < 8; u++) w.p[(i + u) * w.stride] = __uint_as_float(v[u]);
    }
    for (; i < c.n; i++) {
        const uint32_t idx = __float_as_uint(w.p[i * w.stride]);
        w.p[i * w.stride] = __uint_as_float(n_wave ? wave[idx] : 0u);
    }
}

// Sequencers (sequencer.rs:190-246, 482-533): the step machine and the cell decoding live in modules.hip.h.  The 64 grid cells
// are wave-shared data: staged once per tile in an LDS row indexed by STEP (not by lane); every lane then gathers the cell
// of its own current_step.
template <bool kExact>
__device__ __noinline__ void tile_seq(const Ctx c_v, COp& op_v, CArgs& a_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    CArgs& a = uniform_args(a_v);
    const int sr = op.state_row;
    SeqRegs s;
    s.current_step = ROW(sr + SEQ_S_CURRENT);
    s.step_last = ROW(sr + SEQ_S_STEP_LAST) != 0;
    s.sync_last = ROW(sr + SEQ_S_SYNC_LAST) != 0;
    __syncthreads();
    c.rows[op.seq_row * 64 + c.lane] = a.seqtab[op.aux + c.lane];
    __syncthreads();
    const lds_u32* cells = c.rows + op.seq_row * 64;
    const uint32_t length = (uint32_t)op.seq_len;
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    if (op.kind == OP_GRIDSEQ) {
        float last = __uint_as_float(ROW(sr + GRIDSEQ_S_LAST));
        const float inv_spo = 1.0f / par(c, op, GRIDSEQ_P_SPO);  // 1.0 / steps_per_octave as f32 (sequencer.rs:236)
        const Port out[3] = {out_port(c, op.out_slot[0]), out_port(c, op.out_slot[1]), out_port(c, op.out_slot[2])};
        tile_run<2, 3>(c, in, out, [&](const float* x, float* y) {
            const uint32_t cs = seq_advance(s, x[0], x[1], length);
            gridseq_outputs(cells[cs], cs, x[0], inv_spo, last, y[0], y[1], y[2]);
        });
        ROW(sr + GRIDSEQ_S_LAST) = __float_as_uint(last);
    } else {
        // 8 gates + sync, of which a patch reads a few: per group of samples the step machine runs once, then only the
        // ports somebody reads are computed and written (nine LDS rows per sample made this the slowest control module).
        const uint32_t live = op.flags & 0x1ffu;
        Port out[9];
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = out_port(c, op.out_slot[k]);
        constexpr int kU = 4;
        for (int i = 0; i < c.n; i += kU) {
            const int m = min(kU, c.n - i);
            float step_in[kU], sync_in[kU];
            uint32_t cell[kU];
            bool first[kU];
#pragma unroll
            for (int u = 0; u < kU; u++) {  // all reads of the group in flight together (the tail re-reads its last sample)
                const int iu = i + min(u, m - 1);
                step_in[u] = in[0].p[iu * in[0].stride];
                sync_in[u] = in[1].p[iu * in[1].stride];
            }
#pragma unroll
            for (int u = 0; u < kU; u++) {
                if (u >= m) break;
                const uint32_t cs = seq_advance(s, step_in[u], sync_in[u], length);
                cell[u] = cells[cs];
                first[u] = cs == 0u;
            }
#pragma unroll
            for (int k = 0; k < 9; k++) {
                if (!((live >> k) & 1u)) continue;  // wave-uniform
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    if (u >= m) break;
                    out[k].p[(i + u) * out[k].stride] = k == 8 ? (first[u] ? 1.0f : 0.0f) : patseq_gate(cell[u], k, step_in[u]);
                }
            }
        }
    }
    ROW(sr + SEQ_S_CURRENT) = s.current_step;
    ROW(sr + SEQ_S_STEP_LAST) = s.step_last ? 1u : 0u;
    ROW(sr + SEQ_S_SYNC_LAST) = s.sync_last ? 1u : 0u;
}

// Sum the first `rows` rows of an LDS tile [..][64] over the 64 lanes.  R = the power of two >= rows (<= 64): lane l
// owns row l % R and the column segment l / R (64 / R segments of R columns each); columns are visited skewed by the
// row so the 32 lanes of a half-wave hit 32 different banks.  Lanes whose row is past `rows` idle.  Valid in lanes < rows.
template <class Ptr>
SRK_DEV float tile_row_sum(Ptr t, int rows, int lane)
{
    const int R = rows <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)rows - 1u));
    const int row = lane & (R - 1);
    const int seg = lane / R;
    const Ptr p = t + row * 64 + seg * R;
    float sum = 0.0f;
    if (row < rows)
        for (int j = 0; j < R; j++) sum += p[(j + row) & (R - 1)];
    for (int m = R; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
    return sum;
}

template <bool kExact>
__device__ __noinline__ void tile_out(const Ctx c_v, COp& op_v, CArgs& a_v, uint32_t t0, uint32_t voice, bool active)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    CArgs& a = uniform_args(a_v);
    const int slot = op.in_slot[0], plane = op.aux;
    const Port in = in_port(c, slot);  // an LDS wire, or a control track when every voice plays the same thing
    if (a.frames) {
        float* f = a.frames + (size_t)plane * a.plane_stride + (size_t)t0 * a.V + voice;
        if (active) {
            int i = 0;
            for (; i + 8 <= c.n; i += 8) {  // 8 reads in flight, then 8 coalesced 256-B row stores
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) v[u] = in.p[(i + u) * in.stride];
#pragma unroll
                for (int u = 0; u < 8; u++) f[(size_t)(i + u) * a.V] = v[u];
            }
            for (; i < c.n; i++) f[(size_t)i * a.V] = in.p[i * in.stride];
        }
    }
    if (a.mixpart) {
        float* mp = a.mixpart + ((size_t)plane * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride + t0;
        if (slot >= kTrackSlot) {  // identical voices: the wave's partial is (number of real voices) x sample
            if (c.lane < c.n) mp[c.lane] = (float)min(a.lanes, a.V - wave_index(a) * a.lanes) * in.p[c.lane];
            return;
        }
        if (!active)
            for (int i = 0; i < c.n; i++) WIRE(slot, i) = 0.0f;  // lanes past V contribute nothing
        __syncthreads();
        float sum = tile_row_sum(c.wires + slot * c.tile * 64, c.tile, c.lane);
        if (c.lane < c.n) mp[c.lane] = sum;
        __syncthreads();
    }
}

template <bool kExact>
__device__ __noinline__ void tile_delay_rd(const Ctx c_v, COp& op_v, CArgs& a_v, uint64_t n_abs, uint32_t voice_c)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    CArgs& a = uniform_args(a_v);
    const int o = op.out_slot[0];
    const uint32_t B = (uint32_t)a.prog.buffer_size;
    if (op.flags & DELAY_RING_GLOBAL) {
        const float* ring = a.rings + (size_t)op.aux * B * a.V + voice_c;
        uint32_t p = (uint32_t)(n_abs % B);
        int i = 0;
        for (; i + 8 <= c.n; i += 8) {  // 8 ring rows in flight per round trip to HBM / L2
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                v[u] = ring[(size_t)p * a.V];
                p = p + 1 == B ? 0 : p + 1;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) WIRE(o, i + u) = v[u];
        }
        for (; i < c.n; i++) {
            WIRE(o, i) = ring[(size_t)p * a.V];
            p = p + 1 == B ? 0 : p + 1;
        }
    } else {
        uint32_t p = (uint32_t)(n_abs % B);
        for (int i = 0; i < c.n; i++) {
            WIRE(o, i) = __uint_as_float(ROW(op.aux + p));
            p = p + 1 == B ? 0 : p + 1;
        }
    }
}

template <bool kExact>
__device__ __noinline__ void tile_delay_wr(const Ctx c_v, COp& op_v, CArgs& a_v, uint64_t n_abs, uint32_t voice, bool active)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    CArgs& a = uniform_args(a_v);
    const int s = op.in_slot[0];
    const uint32_t B = (uint32_t)a.prog.buffer_size;
    uint32_t p = (uint32_t)(n_abs % B);
    if (op.flags & DELAY_RING_GLOBAL) {
        float* ring = a.rings + (size_t)op.aux * B * a.V + voice;
        for (int i = 0; i < c.n; i++) {
            if (active) ring[(size_t)p * a.V] = WIRE(s, i);
            p = p + 1 == B ? 0 : p + 1;
        }
    } else {
        for (int i = 0; i < c.n; i++) {
            ROW(op.aux + p) = __float_as_uint(WIRE(s, i));
            p = p + 1 == B ? 0 : p + 1;
        }
    }
}

}  // namespace dev

// ---- generic tile interpreter ----------------------------------------------------------------------
// `a` is the argument block seen through the constant address space (the kernarg segment itself, or one entry of the
// stage table in global memory): every field arrives by a scalar load, and tile functions can take its address.
template <bool kExact>
SRK_DEV void interp_body(dev::CArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    if (a.T == 0) return;  // an idle slot of the control pipeline
    dev::CArgs& ca = a;
    const dev::WaveMap wm = dev::wave_map(a, lane);
    const uint32_t voice = wm.voice, voice_c = wm.vc;  // idle lanes shadow the wave's last voice; they never store
    const bool active = wm.active;
    const int n_rows = a.prog.n_rows, tile = a.prog.tile;
    dev::Ctx c;
    c.rows = (dev::lds_u32*)lds;
    c.zero = (dev::lds_f32*)(lds + (size_t)n_rows * 64);
    c.trash = c.zero + 64;
    c.trk = c.trash + 64;
    c.wires = c.trk + a.prog.n_tracks * 64;
    c.zero[lane] = 0.0f;
    c.tile = tile;
    c.lane = lane;
    c.n = 0;
    for (int r = 0; r < n_rows; r++) c.rows[r * 64 + lane] = a.table[(size_t)r * a.V + voice_c];

    // Control tracks are staged 64 samples at a time — a whole LDS row per track, one coalesced load each — and c.trk slides
    // over the staged window tile by tile.  The load's latency is exposed (nothing can be in flight across the calls of the
    // tile functions), so it is paid once per 64 samples instead of once per tile (tile = 12: 750 instead of 4000 round trips
    // per second of audio; worth 2 % on P1 through the interpreter — the per-call overhead of the tile functions is the
    // larger cost there).
    dev::lds_f32* const trk_base = c.trk;
    uint32_t staged_t0 = 0, staged_n = 0;
    for (uint32_t t0 = 0; t0 < a.T; t0 += (uint32_t)tile) {
        c.n = (int)min((uint32_t)tile, a.T - t0);
        if (a.prog.n_tracks > 0) {
            if (t0 + (uint32_t)c.n > staged_t0 + staged_n) {
                staged_t0 = t0;
                staged_n = min(64u, a.T - t0);
                __syncthreads();
                for (int k = 0; k < a.prog.n_tracks; k++)
                    if ((uint32_t)lane < staged_n) trk_base[k * 64 + lane] = a.tracks[(size_t)a.prog.track_id[k] * a.t_stride + t0 + lane];
                __syncthreads();
            }
            c.trk = trk_base + (t0 - staged_t0);
        }
        for (int i = 0; i < a.prog.n_ops; i++) {
            dev::COp& op = ((dev::COp*)a.ops)[i];
            switch (op.kind) {
            case OP_OSC: dev::tile_osc<kExact>(c, op); break;
            case OP_VCF: dev::tile_vcf<kExact>(c, op); break;
            case OP_ADSR: dev::tile_adsr<kExact>(c, op); break;
            case OP_VCA: dev::tile_vca<kExact>(c, op); break;
            case OP_MIX: dev::tile_mix<kExact>(c, op); break;
            case OP_MATH: dev::tile_math<kExact>(c, op); break;
            case OP_OUT: dev::tile_out<kExact>(c, op, ca, t0, voice, active); break;
            case OP_GRIDSEQ:
            case OP_PATSEQ: dev::tile_seq<kExact>(c, op, ca); break;
            case OP_NONLIN: dev::tile_nonlin<kExact>(c, op); break;
            case OP_SAMPLE: dev::tile_sample<kExact>(c, op, ca); break;
            case OP_NOISE: dev::tile_noise<kExact>(c, op, a.n0 + t0, voice_c); break;
            case OP_FREEVERB: dev::tile_freeverb<kExact>(c, op, ca, a.n0 + t0, voice, voice_c, active); break;
            case OP_DELAY_RD: dev::tile_delay_rd<kExact>(c, op, ca, a.n0 + t0, voice_c); break;
            case OP_DELAY_WR: dev::tile_delay_wr<kExact>(c, op, ca, a.n0 + t0, voice, active); break;
            default: break;
            }
        }
    }
    if (active)
        for (int r = 0; r < a.prog.n_state_rows; r++) a.table[(size_t)r * a.V + voice] = c.rows[r * 64 + lane];
}

// Two entry points over one body.  The default flavour is told to fit five waves per SIMD (<= 96 VGPRs; it needs 103
// unconstrained): resident waves are what hides the latency of the per-module dependency chains, and at the headline
// size (16 waves per CU) four per SIMD leaves no slack for the dispatcher.  The exact flavour (f64 PolyBLEP / sin / pow,
// 184 VGPRs) would spill heavily under that cap and is left alone.
template <bool kExact>
__global__ __launch_bounds__(64) void render_interp(KernelArgs a);
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8))) void render_interp<false>(KernelArgs a)
{
    interp_body<false>(*(dev::CArgs*)__builtin_amdgcn_kernarg_segment_ptr());
}
template <>
__global__ __launch_bounds__(64) void render_interp<true>(KernelArgs a)
{
    interp_body<true>(*(dev::CArgs*)__builtin_amdgcn_kernarg_segment_ptr());
}

// The control pipeline: block b runs control unit b (one module) on the chunk its entry of `slots` describes (T == 0:
// nothing to do in this launch).  A unit trails the units it reads by at least one chunk, i.e. it reads tracks written by
// an EARLIER launch: the kernel boundary provides the ordering; within a launch the units touch disjoint state and
// disjoint track ranges.
// (Same register budget as render_interp<false>: the tile functions are shared, and the budget only propagates to
// callees whose callers all agree.)
template <bool kExact>
__global__ __launch_bounds__(64) void render_interp_stages(const KernelArgs* slots);
template <>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(5, 8))) void render_interp_stages<false>(const KernelArgs* slots)
{
    interp_body<false>(*(dev::CArgs*)(uintptr_t)(slots + blockIdx.x));
}
template <>
__global__ __launch_bounds__(64) void render_interp_stages<true>(const KernelArgs* slots)
{
    interp_body<true>(*(dev::CArgs*)(uintptr_t)(slots + blockIdx.x));
}

}  // namespace srack
