// render.hip — the HIP kernels of the batch-render path and their launch code (gfx950 only).
//
// Execution model: one voice per lane, 64 voices per wave, one wave per workgroup (no barriers
// between waves: voices never interact).  A wave owns its voices for the whole render because
// every module is a recurrence in time (phase accumulator, IIR state, envelope state).
//
//   render_interp       generic: executes the flattened op list tile by tile.  Per tile and per
//                       op, one module-type device function runs `tile` samples with the module's
//                       state in VGPRs; wires between ops are [tile][64] f32 tiles in LDS; the
//                       voice table (state + per-voice parameters) sits in LDS for the whole
//                       render.  HBM is touched for: the voice table (once in, once out), rendered
//                       frames (coalesced 256 B per wave-store), mix partials, and the rings of
//                       broken feedback edges when buffer_size > 16.
//   render_voice_chain  fused special case for patch P1's shape: every wire and all state in
//                       VGPRs, no LDS except the mix-down transpose tile.
//   mix_reduce          second pass of the mix-down: sums the per-wave partials (deterministic
//                       order, no atomics).
//
// HBM layout (all voice-minor so that lane == voice gives 256 B contiguous per wave access):
//   table   u32 [n_rows][V]           state rows, then per-voice parameter rows
//   frames  f32 [planes][T][V]
//   rings   f32 [n_rings][B][V]
//   mixpart f32 [planes][n_waves][T]  per-wave partial sums, T contiguous
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "modules.hip.h"
#include "runtime.hpp"

namespace srack {

struct KernelArgs {
    const DevOp* ops;
    DevProgram prog;
    uint32_t* table;
    float* rings;
    float* frames;
    float* mixpart;
    const float* tracks;  // control tracks [n_tracks][t_stride] written by the control program (may be null)
    const uint32_t* seqtab;  // sequencer grids, 64 cells per sequencer op
    uint32_t V, T, n_waves;
    uint32_t lanes;  // voices per wave: 64, or 32 / 16 when there are too few voices to fill the SIMDs (idle lanes shadow the wave's last voice)
    // A launch covers T samples of a render of t_stride samples; frames / mixpart / tracks arrive pre-offset to
    // the launch's first sample and keep the whole render's strides.
    uint64_t plane_stride;  // frames: elements between planes (= t_stride * V)
    uint32_t t_stride;
    uint32_t block0;  // blocks [0, block0) of the grid are not voice waves (a co-scheduled control block); wave = blockIdx.x - block0
    uint64_t n0;  // absolute index of this launch's first sample (phase of the feedback rings)
};

struct ChainRoles {  // op indices of the fused voice chain (osc_l / adsr unused in the track variant)
    int osc_a, osc_l, vcf, adsr, vca, out, track;
};

struct SeqRoles {  // the fused sequencer-driven voice chain: op indices and rows of the track buffer
    int math, osc, vcf, vca, out;        // math = -1: the oscillator's CV is the note track itself
    int trk_pitch, trk_cutoff, trk_env;  // trk_cutoff = -1: the filter has no CV
    int n_extra;                         // further output planes that carry a track as it is
    int extra_plane[4], extra_trk[4];
};

namespace dev {

SRK_DEV double make_f64(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
SRK_DEV uint32_t f64_lo(double d) { return (uint32_t)__double2loint(d); }
SRK_DEV uint32_t f64_hi(double d) { return (uint32_t)__double2hiint(d); }

struct WaveMap {   // which voices a wave owns
    uint32_t wave0;     // first voice of the wave
    uint32_t n_active;  // real voices in it (lanes >= n_active shadow voice wave0 + n_active - 1: same work, same stores)
    uint32_t voice;     // this lane's voice (meaningful when active)
    uint32_t vc;        // this lane's voice clamped to a real one (safe to load from)
    bool active;
};

template <class Args>
SRK_DEV WaveMap wave_map(const Args& a, int lane)
{
    WaveMap m;
    m.wave0 = (blockIdx.x - a.block0) * a.lanes;
    m.n_active = min(a.lanes, a.V - m.wave0);
    m.active = (uint32_t)lane < m.n_active;
    m.voice = m.wave0 + (uint32_t)lane;
    m.vc = m.active ? m.voice : m.wave0 + m.n_active - 1;
    return m;
}

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

template <bool kExact>
__device__ __noinline__ void tile_osc(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const uint32_t fl = op.flags;
    const int sr = op.state_row;
    OscRegs s;
    s.pos = make_f64(ROW(sr + OSC_S_POS_LO), ROW(sr + OSC_S_POS_HI));
    s.sync_last = ROW(sr + OSC_S_SYNC_LAST) != 0;
    OscConst k;
    k.sr = op.sample_rate;
    k.val = (double)par(c, op, OSC_P_VAL);
    k.delta = op.delta_row >= 0 ? make_f64(ROW(op.delta_row), ROW(op.delta_row + 1)) : op.delta;
    k.inv_dt = 1.0f / (float)k.delta;
    const Port out[3] = {out_port(c, op.out_slot[0]), out_port(c, op.out_slot[1]), out_port(c, op.out_slot[2])};
    if (fl & OSC_CONST_FAST) {  // no CV, no sync, one live port, delta < 0.25 for every voice (host-checked)
        COsc o;
        cosc_init(o, s.pos, k.delta);
        const Port none[1] = {in_port(c, -1)};
        if (fl & OSC_OUT_SAW) {
            const Port w[1] = {out[2]};
            tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = cosc_saw(o); });
        } else if (fl & OSC_OUT_SQUARE) {
            const Port w[1] = {out[1]};
            tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = cosc_square(o); });
        } else {
            const Port w[1] = {out[0]};
            tile_run<1, 1>(c, none, w, [&](const float*, float* y) { y[0] = cosc_sine(o); });
        }
        ROW(sr + OSC_S_POS_LO) = f64_lo(o.pos);
        ROW(sr + OSC_S_POS_HI) = f64_hi(o.pos);
        ROW(sr + OSC_S_SYNC_LAST) = 0u;  // sync unconnected: `last` follows the constant 0.0 input
        return;
    }
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const uint32_t ports = fl & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
    if (!kExact && (fl & (OSC_HAS_CV | OSC_CV_STEPWISE | OSC_HAS_SYNC | OSC_AA)) == (OSC_HAS_CV | OSC_CV_STEPWISE | OSC_AA) && ports && !(ports & (ports - 1))) {
        // A sequencer-driven pitch: the carried-phase oscillator between note changes.  When some lane's CV differs from
        // the one its increment was computed for (a wave-uniform test), that increment is recomputed — 440 / sr x 2^(cv +
        // val), as osc_step does — and the carried terms are rebuilt from the exact f64 phase.  An increment of 0.25 or
        // more (or NaN) breaks the carried form's "one PolyBLEP window at a time": those samples take osc_step.
        COsc o;
        float seen_cv = __builtin_nanf("");
        bool carried = false;
        const uint32_t f = fl & ~OSC_EXACT;
        const Port cvp[1] = {in[0]};
        const Port w[1] = {out[(fl & OSC_OUT_SAW) ? 2 : (fl & OSC_OUT_SQUARE) ? 1 : 0]};
        o.pos = s.pos;
        tile_run<1, 1>(c, cvp, w, [&](const float* x, float* y) {
            const float cv = x[0];
            if (__builtin_amdgcn_ballot_w64(cv != seen_cv) != 0) {
                const double delta = (440.0 / k.sr) * exp2_fast((double)cv + k.val);
                seen_cv = cv;
                carried = __builtin_amdgcn_ballot_w64(!(delta < 0.25)) == 0;
                cosc_init(o, o.pos, delta);
            }
            if (carried) {
                y[0] = (fl & OSC_OUT_SAW) ? cosc_saw(o) : (fl & OSC_OUT_SQUARE) ? cosc_square(o) : cosc_sine(o);
            } else {
                OscRegs g;
                g.pos = o.pos;
                g.sync_last = false;
                g.seen_cv = seen_cv;
                g.seen_delta = o.delta;
                float o3[3] = {0.0f, 0.0f, 0.0f};
                osc_step(f, g, k, cv, 0.0f, o3[0], o3[1], o3[2]);
                y[0] = (fl & OSC_OUT_SAW) ? o3[2] : (fl & OSC_OUT_SQUARE) ? o3[1] : o3[0];
                cosc_init(o, g.pos, o.delta);
            }
        });
        ROW(sr + OSC_S_POS_LO) = f64_lo(o.pos);
        ROW(sr + OSC_S_POS_HI) = f64_hi(o.pos);
        ROW(sr + OSC_S_SYNC_LAST) = 0u;
        return;
    }
    const uint32_t f = kExact ? (fl | OSC_EXACT) : (fl & ~OSC_EXACT);
    tile_run<2, 3>(c, in, out, [&](const float* x, float* y) {
        y[0] = y[1] = y[2] = 0.0f;
        osc_step(f, s, k, x[0], x[1], y[0], y[1], y[2]);
    });
    ROW(sr + OSC_S_POS_LO) = f64_lo(s.pos);
    ROW(sr + OSC_S_POS_HI) = f64_hi(s.pos);
    ROW(sr + OSC_S_SYNC_LAST) = s.sync_last ? 1u : 0u;
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

template <bool kExact>
__device__ __noinline__ void tile_vcf(const Ctx c_v, COp& op_v)
{
    const Ctx c = uniform_ctx(c_v);
    COp& op = uniform_op(op_v);
    const uint32_t fl = op.flags;
    VcfRegs s;
    vcf_load(c, op.state_row, s);
    const float freq = par(c, op, VCF_P_FREQ), exp_amt = par(c, op, VCF_P_EXP);
    const float res = vcf_resonance(par(c, op, VCF_P_RES));
    const Port in[2] = {in_port(c, op.in_slot[0]), in_port(c, op.in_slot[1])};
    const Port out[3] = {out_port(c, op.out_slot[0]), out_port(c, op.out_slot[1]), out_port(c, op.out_slot[2])};
    if (fl & VCF_HAS_CV) {
        tile_run<2, 3>(c, in, out, [&](const float* x, float* y) {
            vcf_coeffs(s, vcf_frequency(freq, x[1], exp_amt), res);
            vcf_step<!kExact>(s, x[0], y[0], y[1], y[2]);
        });
    } else {
        // constant cutoff: the "did (frequency, res) change" check can only fire on the first sample
        vcf_coeffs(s, vcf_frequency(freq, 0.0f, exp_amt), res);
        tile_run<2, 3>(c, in, out, [&](const float* x, float* y) { vcf_step<!kExact>(s, x[0], y[0], y[1], y[2]); });
    }
    vcf_store(c, op.state_row, s);
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

// Sequencers (sequencer.rs:190-246, 482-533).  The 64 grid cells are wave-shared data: staged once per tile in an LDS
// row indexed by STEP (not by lane); every lane then gathers the cell of its own current_step.
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
            const uint32_t cell = cells[cs];
            const bool present = cell & 0x80000000u, hold = cell & 0x40000000u;
            y[0] = present ? (float)(cell & 0xffffu) * inv_spo : last;
            y[1] = present ? (hold ? 1.0f : x[0]) : 0.0f;
            y[2] = cs == 0u ? 1.0f : 0.0f;
            last = y[0];
        });
        ROW(sr + GRIDSEQ_S_LAST) = __float_as_uint(last);
    } else {
        Port out[9];
#pragma unroll
        for (int k = 0; k < 9; k++) out[k] = out_port(c, op.out_slot[k]);
        tile_run<2, 9>(c, in, out, [&](const float* x, float* y) {
            const uint32_t cs = seq_advance(s, x[0], x[1], length);
            const uint32_t cell = cells[cs];
#pragma unroll
            for (int ch = 0; ch < 8; ch++) {
                const uint32_t b = (cell >> (2 * ch)) & 3u;
                y[ch] = (b & 1u) ? ((b & 2u) ? 1.0f : x[0]) : 0.0f;
            }
            y[8] = cs == 0u ? 1.0f : 0.0f;
        });
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
            if (c.lane < c.n) mp[c.lane] = (float)min(a.lanes, a.V - (blockIdx.x - a.block0) * a.lanes) * in.p[c.lane];
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

    for (uint32_t t0 = 0; t0 < a.T; t0 += (uint32_t)tile) {
        c.n = (int)min((uint32_t)tile, a.T - t0);
        if (a.prog.n_tracks > 0) {  // this tile's slice of every control track the program reads: one coalesced load per track
            __syncthreads();
            for (int k = 0; k < a.prog.n_tracks; k++)
                if (lane < c.n) c.trk[k * 64 + lane] = a.tracks[(size_t)a.prog.track_id[k] * a.t_stride + t0 + lane];
            __syncthreads();
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

constexpr int kMixRows = 32;

// ---- per-sample output of the fused kernels ---------------------------------------------------------------
// kOut: 0 = decide at run time (exact-mode kernels), 1 = frames only, 2 = mix only, 3 = frames + mix.
// Frames: SGPR row base advanced by V per sample + a constant per-lane offset; lanes past V (only in the
// last wave) shadow voice V-1, compute the identical sample and store it to the identical address, so the
// store needs no exec mask.  Mix: the sample goes into a 32-row LDS tile; every 32 samples (and at the end)
// the rows are summed over the 64 lanes (tile_row_sum) and one lane per row writes the wave's partial.
struct Emit {
    float* frame_row;   // wave-uniform
    float* mp;          // wave-uniform: mixpart row of this wave
    bool has_frames, has_mix, full_wave;
    int lane, lane_c;
    uint32_t n_active;  // lanes of this wave that are real voices
};

template <int kOut>
SRK_DEV void emit_put(Emit& e, float* mix_tile, float o, int i, uint32_t V)  // i = row of the current 32-sample tile
{
    const bool frames = kOut == 0 ? e.has_frames : (kOut & 1) != 0;
    const bool mix = kOut == 0 ? e.has_mix : (kOut & 2) != 0;
    if (frames) {
        __builtin_nontemporal_store(o, &e.frame_row[e.lane_c]);  // write-once stream: keep it out of the L2's way
        e.frame_row += V;
    }
    if (mix) mix_tile[i * 64 + e.lane] = o;
}

template <int kOut>
SRK_DEV void emit_flush(Emit& e, float* mix_tile, uint32_t t0, int n)  // the tile holds samples t0 .. t0+n-1
{
    using dev::tile_row_sum;
    const bool mix = kOut == 0 ? e.has_mix : (kOut & 2) != 0;
    if (!mix) return;
    if (!e.full_wave && (uint32_t)e.lane >= e.n_active)  // shadow lanes contribute nothing to the mix
        for (int r = 0; r < kMixRows; r++) mix_tile[r * 64 + e.lane] = 0.0f;
    __syncthreads();
    const float sum = tile_row_sum(mix_tile, kMixRows, e.lane);
    if (e.lane < n) e.mp[t0 + e.lane] = sum;
    __syncthreads();
}

SRK_DEV Emit make_emit(const KernelArgs& a, int plane, int lane)
{
    using dev::WaveMap;
    using dev::wave_map;
    Emit e;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t wave0 = wm.wave0;
    e.n_active = wm.n_active;
    e.full_wave = e.n_active == 64u;
    e.lane = lane;
    e.lane_c = min(lane, (int)e.n_active - 1);
    e.frame_row = a.frames ? a.frames + (size_t)plane * a.plane_stride + wave0 : nullptr;
    e.mp = a.mixpart ? a.mixpart + ((size_t)plane * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride : nullptr;
    e.has_frames = e.frame_row != nullptr;
    e.has_mix = e.mp != nullptr;
    return e;
}

// ---- fused control chain: OSC (constant pitch) -> ADSR -> track ---------------------------------------------
// The voice-invariant half of patch P1's shape: one voice, one wave, every lane computes the same numbers.
// It is a pure latency chain (phase accumulate -> gate -> envelope state machine), so it is kept short: state
// in VGPRs, the carried-phase oscillator and the segmented ADSR, 64 samples gathered across lanes per store.
// What the control block needs: a slice of KernelArgs small enough to ride along with a voice kernel's arguments.
struct CtlWork {
    const DevOp* ops;
    uint32_t* table;   // the control program's one-voice table
    float* track;      // this chunk's first sample of the envelope track
    uint32_t T;        // samples to produce (0: nothing to do)
    uint32_t port;     // OSC_OUT_* of the gate oscillator
};

template <uint32_t kOscPort>
SRK_DEV void ctl_gate_env_body(const CtlWork& a)
{
    using namespace dev;
    const ChainRoles r{0, 0, 0, 1, 0, 2, 0};  // op order of the matched control program: OSC, ADSR, OUT
    const int lane = threadIdx.x;
    auto row = [&](int rr) { return a.table[rr]; };  // V == 1
    const DevOp& ol = a.ops[r.osc_l];
    const DevOp& od = a.ops[r.adsr];
    float* __restrict__ track = a.track;

    COsc cl;
    cosc_init(cl, make_f64(row(ol.state_row + OSC_S_POS_LO), row(ol.state_row + OSC_S_POS_HI)), ol.delta);
    AdsrRegs sd;
    sd.phase = __uint_as_float(row(od.state_row + ADSR_S_PHASE));
    sd.mode = (int)row(od.state_row + ADSR_S_MODE);
    sd.r_val = __uint_as_float(row(od.state_row + ADSR_S_R_VAL));
    sd.from_a_val = __uint_as_float(row(od.state_row + ADSR_S_FROM_A));
    sd.gate_last = row(od.state_row + ADSR_S_GATE_LAST) != 0;
    const AdsrConst kd = adsr_consts(od.par_val[ADSR_P_A], od.par_val[ADSR_P_D], od.par_val[ADSR_P_S], od.par_val[ADSR_P_R], od.par_val[ADSR_P_SR]);
    AdsrSeg seg;
    adsr_seg_enter(sd, kd, seg);

    // Four samples at a time on the assumption that nothing happens in them: the square stays outside its PolyBLEP
    // windows (so it is exactly -1/+1) and the envelope stays in its segment.  One scalar test per group instead
    // of two per sample; when the assumption fails the group is redone one sample at a time (cosc / adsr_seg).
    float out[4];
    auto try_group = [&]() -> bool {
        double pos = cl.pos;
        float ph = sd.phase;
        uint64_t last = seg.last, bad = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int hw = __double2hiint(pos);
            bad |= __builtin_amdgcn_ballot_w64(hw <= cl.hA) | __builtin_amdgcn_ballot_w64(hw >= cl.hB) |
                   __builtin_amdgcn_ballot_w64((uint32_t)(hw - cl.hQ0) <= cl.hQspan);
            const uint64_t high = __builtin_amdgcn_ballot_w64(hw >= 0x3fe00000);  // square = +1 > 0  <=>  pos >= 0.5
            pos = __builtin_amdgcn_fract(pos + cl.delta);
            ph = ph + seg.inc;
            bad |= __builtin_amdgcn_ballot_w64(ph >= 1.0f) | (high & seg.on_high) | (~high & seg.on_low) | (high & ~last & seg.on_edge);
            last = high;
            out[q] = seg.c0 + seg.c1 * (seg.k0 + seg.k1 * ph);
        }
        if (bad != 0) return false;
        cl.pos = pos;
        sd.phase = ph;
        seg.last = last;
        seg.held = out[3];
        return true;
    };

    for (uint32_t t0 = 0; t0 < a.T; t0 += 64) {
        const int n = (int)min(64u, a.T - t0);
        float keep_v = 0.0f;  // lane j keeps sample t0 + j
        int j = 0;
        while (j < n) {
            if (kOscPort == OSC_OUT_SQUARE && j + 4 <= n && try_group()) {
#pragma unroll
                for (int q = 0; q < 4; q++) keep_v = lane == j + q ? out[q] : keep_v;
                j += 4;
                continue;
            }
            const int stop = min(n, j + 4);
            for (; j < stop; j++) {
                const float gate = cosc_step<kOscPort>(cl);
                const float env = adsr_seg_step(sd, kd, seg, gate);
                keep_v = lane == j ? env : keep_v;
            }
        }
        if (lane < n) track[t0 + lane] = keep_v;
    }
    adsr_seg_flush(sd, seg);
    if (lane == 0) {
        a.table[ol.state_row + OSC_S_POS_LO] = f64_lo(cl.pos);
        a.table[ol.state_row + OSC_S_POS_HI] = f64_hi(cl.pos);
        a.table[ol.state_row + OSC_S_SYNC_LAST] = 0u;
        a.table[od.state_row + ADSR_S_PHASE] = __float_as_uint(sd.phase);
        a.table[od.state_row + ADSR_S_MODE] = (uint32_t)sd.mode;
        a.table[od.state_row + ADSR_S_R_VAL] = __float_as_uint(sd.r_val);
        a.table[od.state_row + ADSR_S_FROM_A] = __float_as_uint(sd.from_a_val);
        a.table[od.state_row + ADSR_S_GATE_LAST] = sd.gate_last ? 1u : 0u;
    }
}


SRK_DEV void ctl_gate_env(const CtlWork& w)
{
    if (w.port == OSC_OUT_SQUARE)
        ctl_gate_env_body<OSC_OUT_SQUARE>(w);
    else if (w.port == OSC_OUT_SAW)
        ctl_gate_env_body<OSC_OUT_SAW>(w);
    else
        ctl_gate_env_body<OSC_OUT_SINE>(w);
}

__global__ __launch_bounds__(64) void render_ctl_gate_env(CtlWork w) { ctl_gate_env(w); }

// ---- fused voice chain (patch P1's shape) -------------------------------------------------------------
// OSC_A.<port> -> VCF.<port> -> VCA <- ADSR <- OSC_L.<port>; all wires and all state in VGPRs.

template <uint32_t kOscAPort, uint32_t kOscLPort, uint32_t kVcfPort, bool kExact, int kOut>
__global__ __launch_bounds__(64) void render_voice_chain(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    __shared__ float mix_tile[kMixRows * 64];
    const int lane = threadIdx.x;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t voice = wm.voice, vc = wm.vc, V = a.V;
    const bool active = wm.active;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };

    const DevOp& oa = a.ops[r.osc_a];
    const DevOp& ol = a.ops[r.osc_l];
    const DevOp& ov = a.ops[r.vcf];
    const DevOp& od = a.ops[r.adsr];
    const DevOp& oc = a.ops[r.vca];
    const int plane = a.ops[r.out].aux;

    constexpr uint32_t kEx = kExact ? OSC_EXACT : 0u;
    constexpr uint32_t fa = OSC_AA | kOscAPort | kEx;
    constexpr uint32_t fl = OSC_AA | kOscLPort | kEx;

    OscRegs sa, sl;
    OscConst ka, kl;
    sa.pos = make_f64(row(oa.state_row + OSC_S_POS_LO), row(oa.state_row + OSC_S_POS_HI));
    sa.sync_last = row(oa.state_row + OSC_S_SYNC_LAST) != 0;
    sl.pos = make_f64(row(ol.state_row + OSC_S_POS_LO), row(ol.state_row + OSC_S_POS_HI));
    sl.sync_last = row(ol.state_row + OSC_S_SYNC_LAST) != 0;
    ka.sr = oa.sample_rate;
    ka.val = 0.0;
    ka.delta = oa.delta_row >= 0 ? make_f64(row(oa.delta_row), row(oa.delta_row + 1)) : oa.delta;
    ka.inv_dt = 1.0f / (float)ka.delta;
    kl.sr = ol.sample_rate;
    kl.val = 0.0;
    kl.delta = ol.delta_row >= 0 ? make_f64(row(ol.delta_row), row(ol.delta_row + 1)) : ol.delta;
    kl.inv_dt = 1.0f / (float)kl.delta;

    VcfRegs sv;
    {
        const int s0 = ov.state_row;
        sv.f = __uint_as_float(row(s0 + VCF_S_F));
        sv.p = __uint_as_float(row(s0 + VCF_S_P));
        sv.q = __uint_as_float(row(s0 + VCF_S_Q));
        sv.b0 = __uint_as_float(row(s0 + VCF_S_B0 + 0));
        sv.b1 = __uint_as_float(row(s0 + VCF_S_B0 + 1));
        sv.b2 = __uint_as_float(row(s0 + VCF_S_B0 + 2));
        sv.b3 = __uint_as_float(row(s0 + VCF_S_B0 + 3));
        sv.b4 = __uint_as_float(row(s0 + VCF_S_B0 + 4));
        sv.freq = __uint_as_float(row(s0 + VCF_S_FREQ));
        sv.res = __uint_as_float(row(s0 + VCF_S_RES));
    }
    if (a.T > 0) vcf_coeffs(sv, vcf_frequency(parv(ov, VCF_P_FREQ), 0.0f, parv(ov, VCF_P_EXP)), vcf_resonance(parv(ov, VCF_P_RES)));

    AdsrRegs sd;
    sd.phase = __uint_as_float(row(od.state_row + ADSR_S_PHASE));
    sd.mode = (int)row(od.state_row + ADSR_S_MODE);
    sd.r_val = __uint_as_float(row(od.state_row + ADSR_S_R_VAL));
    sd.from_a_val = __uint_as_float(row(od.state_row + ADSR_S_FROM_A));
    sd.gate_last = row(od.state_row + ADSR_S_GATE_LAST) != 0;
    const AdsrConst kd = adsr_consts(parv(od, ADSR_P_A), parv(od, ADSR_P_D), parv(od, ADSR_P_S), parv(od, ADSR_P_R), parv(od, ADSR_P_SR));
    const bool negative = parv(oc, VCA_P_NEG) != 0.0f;

    Emit em = make_emit(a, plane, lane);

    COsc ca, cl;
    AdsrSeg seg;
    float x = 0.0f, gate = 0.0f;
    double pos_a = sa.pos, pos_l = sl.pos;  // oscillator phases after exactly t samples (the loop runs one sample ahead)
    if (!kExact) {
        cosc_init(ca, sa.pos, ka.delta);
        cosc_init(cl, sl.pos, kl.delta);
        adsr_seg_enter(sd, kd, seg);
        if (a.T > 0) {  // software pipeline: the oscillators of sample t+1 are evaluated beside the filter of sample t
            x = cosc_step<kOscAPort>(ca);
            gate = cosc_step<kOscLPort>(cl);
        }
    }

    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        for (int i = 0; i < n; i++) {
            float env, x_next = 0.0f, gate_next = 0.0f;
            if (kExact) {
                float sine = 0.0f, square = 0.0f, saw = 0.0f;
                osc_step(fa, sa, ka, 0.0f, 0.0f, sine, square, saw);
                x = kOscAPort == OSC_OUT_SINE ? sine : (kOscAPort == OSC_OUT_SQUARE ? square : saw);
                float gs = 0.0f, gq = 0.0f, gw = 0.0f;
                osc_step(fl, sl, kl, 0.0f, 0.0f, gs, gq, gw);
                gate = kOscLPort == OSC_OUT_SINE ? gs : (kOscLPort == OSC_OUT_SQUARE ? gq : gw);
            }
            float lp, bp, hp;
            vcf_step<!kExact>(sv, x, lp, bp, hp);
            const float y = kVcfPort == VCF_OUT_LP ? lp : (kVcfPort == VCF_OUT_BP ? bp : hp);
            if (!kExact) {  // next sample's oscillators: same basic block as the filter chain above => they interleave
                pos_a = ca.pos;
                pos_l = cl.pos;
                x_next = cosc_step<kOscAPort>(ca);
                gate_next = cosc_step<kOscLPort>(cl);
            }
            if (kExact)
                env = adsr_step(ADSR_HAS_GATE, sd, kd, gate);
            else
                env = adsr_seg_step(sd, kd, seg, gate);
            const float o = vca_step(VCA_HAS_AUDIO | VCA_HAS_CV, negative, y, env);
            emit_put<kOut>(em, mix_tile, o, i, V);
            if (!kExact) {
                x = x_next;
                gate = gate_next;
            }
        }
        emit_flush<kOut>(em, mix_tile, t0, n);
    }
    if (!kExact) {
        sa.pos = pos_a;
        sl.pos = pos_l;
        sa.sync_last = sl.sync_last = false;  // sync unconnected: `last` follows the constant 0.0 input
        adsr_seg_flush(sd, seg);
    }

    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(oa.state_row + OSC_S_POS_LO, f64_lo(sa.pos));
        put(oa.state_row + OSC_S_POS_HI, f64_hi(sa.pos));
        put(oa.state_row + OSC_S_SYNC_LAST, sa.sync_last ? 1u : 0u);
        put(ol.state_row + OSC_S_POS_LO, f64_lo(sl.pos));
        put(ol.state_row + OSC_S_POS_HI, f64_hi(sl.pos));
        put(ol.state_row + OSC_S_SYNC_LAST, sl.sync_last ? 1u : 0u);
        const int s0 = ov.state_row;
        put(s0 + VCF_S_F, __float_as_uint(sv.f));
        put(s0 + VCF_S_P, __float_as_uint(sv.p));
        put(s0 + VCF_S_Q, __float_as_uint(sv.q));
        put(s0 + VCF_S_B0 + 0, __float_as_uint(sv.b0));
        put(s0 + VCF_S_B0 + 1, __float_as_uint(sv.b1));
        put(s0 + VCF_S_B0 + 2, __float_as_uint(sv.b2));
        put(s0 + VCF_S_B0 + 3, __float_as_uint(sv.b3));
        put(s0 + VCF_S_B0 + 4, __float_as_uint(sv.b4));
        put(s0 + VCF_S_FREQ, __float_as_uint(sv.freq));
        put(s0 + VCF_S_RES, __float_as_uint(sv.res));
        put(od.state_row + ADSR_S_PHASE, __float_as_uint(sd.phase));
        put(od.state_row + ADSR_S_MODE, (uint32_t)sd.mode);
        put(od.state_row + ADSR_S_R_VAL, __float_as_uint(sd.r_val));
        put(od.state_row + ADSR_S_FROM_A, __float_as_uint(sd.from_a_val));
        put(od.state_row + ADSR_S_GATE_LAST, sd.gate_last ? 1u : 0u);
    }
}

// ---- fused voice chain, envelope from a control track (P1 after uniform hoisting) ---------------------
// OSC_A.<port> -> VCF.<port> -> VCA <- track[t]; the track sample is wave-uniform (scalar load, SGPR operand).
// The loop body is one basic block: the filter chain of sample t interleaves with the oscillator of t+1.
template <uint32_t kOscAPort, uint32_t kVcfPort, bool kExact, int kOut>
__global__ __launch_bounds__(64) void render_voice_chain_track(KernelArgs a, ChainRoles r, CtlWork co)
{
    using namespace dev;
    __shared__ float mix_tile[kMixRows * 64];
    // Block 0 of a co-scheduled launch is not a voice wave: it computes the NEXT chunk's envelope track while the
    // voice blocks consume this chunk's (written by the previous launch).  Same stream, no events, no second queue.
    if (blockIdx.x < a.block0) {
        // a latency chain sharing its SIMD with four throughput-bound voice waves: without priority it gets a
        // fifth of the issue slots and can outlast the voice blocks (measured: 1.9 -> 2.6 ms per launch)
        __builtin_amdgcn_s_setprio(3);
        ctl_gate_env(co);
        return;
    }
    const int lane = threadIdx.x;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t voice = wm.voice, vc = wm.vc, V = a.V;
    const bool active = wm.active;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };

    const DevOp& oa = a.ops[r.osc_a];
    const DevOp& ov = a.ops[r.vcf];
    const DevOp& oc = a.ops[r.vca];
    const int plane = a.ops[r.out].aux;
    const float* __restrict__ env_track = a.tracks + (size_t)r.track * a.t_stride;

    constexpr uint32_t fa = OSC_AA | kOscAPort | (kExact ? OSC_EXACT : 0u);
    OscRegs sa;
    OscConst ka;
    sa.pos = make_f64(row(oa.state_row + OSC_S_POS_LO), row(oa.state_row + OSC_S_POS_HI));
    sa.sync_last = row(oa.state_row + OSC_S_SYNC_LAST) != 0;
    ka.sr = oa.sample_rate;
    ka.val = 0.0;
    ka.delta = oa.delta_row >= 0 ? make_f64(row(oa.delta_row), row(oa.delta_row + 1)) : oa.delta;
    ka.inv_dt = 1.0f / (float)ka.delta;

    VcfRegs sv;
    const int s0 = ov.state_row;
    sv.f = __uint_as_float(row(s0 + VCF_S_F));
    sv.p = __uint_as_float(row(s0 + VCF_S_P));
    sv.q = __uint_as_float(row(s0 + VCF_S_Q));
    sv.b0 = __uint_as_float(row(s0 + VCF_S_B0 + 0));
    sv.b1 = __uint_as_float(row(s0 + VCF_S_B0 + 1));
    sv.b2 = __uint_as_float(row(s0 + VCF_S_B0 + 2));
    sv.b3 = __uint_as_float(row(s0 + VCF_S_B0 + 3));
    sv.b4 = __uint_as_float(row(s0 + VCF_S_B0 + 4));
    sv.freq = __uint_as_float(row(s0 + VCF_S_FREQ));
    sv.res = __uint_as_float(row(s0 + VCF_S_RES));
    if (a.T > 0) vcf_coeffs(sv, vcf_frequency(parv(ov, VCF_P_FREQ), 0.0f, parv(ov, VCF_P_EXP)), vcf_resonance(parv(ov, VCF_P_RES)));
    const bool negative = parv(oc, VCA_P_NEG) != 0.0f;

    Emit em = make_emit(a, plane, lane);

    COsc ca;
    float x = 0.0f;
    double pos_a = sa.pos;
    if (!kExact) {
        cosc_init(ca, sa.pos, ka.delta);
        if (a.T > 0) x = cosc_step<kOscAPort>(ca);
    }
    // The envelope track is wave-uniform.  Lane l prefetches sample t0 + l of the NEXT 64-sample tile with one
    // coalesced load while the current tile is consumed through v_readlane (SGPR operand): no per-sample memory wait.
    float env_tile = env_track[min((uint32_t)(lane & (kMixRows - 1)), a.T - 1)];
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const float env_next = env_track[min(t0 + kMixRows + (uint32_t)(lane & (kMixRows - 1)), a.T - 1)];
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        auto sample = [&](int i) {
            const float env = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(env_tile), i));
            if (kExact) {
                float sine = 0.0f, square = 0.0f, saw = 0.0f;
                osc_step(fa, sa, ka, 0.0f, 0.0f, sine, square, saw);
                x = kOscAPort == OSC_OUT_SINE ? sine : (kOscAPort == OSC_OUT_SQUARE ? square : saw);
            }
            float lp, bp, hp;
            vcf_step<!kExact>(sv, x, lp, bp, hp);
            const float y = kVcfPort == VCF_OUT_LP ? lp : (kVcfPort == VCF_OUT_BP ? bp : hp);
            if (!kExact) {
                pos_a = ca.pos;
                x = cosc_step<kOscAPort>(ca);  // sample t+1
            }
            // vca.rs:132: (negative || cv > 0.0) ? audio * cv : 0.0 — cv is wave-uniform here, so `cv > 0.0` is decided
            // on the scalar unit from the bit pattern: positive, non-zero, not NaN  <=>  0 < bits <= 0x7f800000
            const bool cv_pos = (uint32_t)(__float_as_int(env) - 1) < 0x7f800000u;
            const float o = (negative || cv_pos) ? y * env : 0.0f;
            emit_put<kOut>(em, mix_tile, o, i, V);
        };
        if (n == kMixRows) {  // constant trip count: unrollable (readlane is convergent, so a runtime count is not)
#pragma unroll 8
            for (int i = 0; i < kMixRows; i++) sample(i);
        } else {
            for (int i = 0; i < n; i++) sample(i);
        }
        emit_flush<kOut>(em, mix_tile, t0, n);
        env_tile = env_next;
    }
    if (!kExact) {
        sa.pos = pos_a;
        sa.sync_last = false;
    }
    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(oa.state_row + OSC_S_POS_LO, f64_lo(sa.pos));
        put(oa.state_row + OSC_S_POS_HI, f64_hi(sa.pos));
        put(oa.state_row + OSC_S_SYNC_LAST, sa.sync_last ? 1u : 0u);
        put(s0 + VCF_S_F, __float_as_uint(sv.f));
        put(s0 + VCF_S_P, __float_as_uint(sv.p));
        put(s0 + VCF_S_Q, __float_as_uint(sv.q));
        put(s0 + VCF_S_B0 + 0, __float_as_uint(sv.b0));
        put(s0 + VCF_S_B0 + 1, __float_as_uint(sv.b1));
        put(s0 + VCF_S_B0 + 2, __float_as_uint(sv.b2));
        put(s0 + VCF_S_B0 + 3, __float_as_uint(sv.b3));
        put(s0 + VCF_S_B0 + 4, __float_as_uint(sv.b4));
        put(s0 + VCF_S_FREQ, __float_as_uint(sv.freq));
        put(s0 + VCF_S_RES, __float_as_uint(sv.res));
    }
}

// ---- fused sequencer-driven voice chain (patch P3's shape after hoisting) ------------------------------------------
//   [MATH(note track, k)] -> OSC.cv ; OSC -> VCF (cutoff CV = envelope track) -> VCA (CV = envelope track) -> OUT,
//   plus output channels that carry a track unchanged (a raw gate).  The three tracks are wave-uniform: each is
//   prefetched one 32-sample tile ahead (lane l holds sample l) and read per sample with v_readlane, so "did the note /
//   the cutoff CV change" is a scalar compare.  Between note changes the oscillator is the carried-phase one; at a
//   change every lane recomputes its increment 440 / sr * 2^(cv + val) and rebuilds the carried terms from the exact f64
//   phase (as tile_osc's stepwise path).  The filter coefficients are recomputed only when the cutoff CV's bits changed
//   (vcf_coeffs re-checks per lane, as filter.rs:61 does).
template <uint32_t kOscPort, int kOut>
__global__ __launch_bounds__(64) void render_voice_chain_seq(KernelArgs a, SeqRoles r)
{
    using namespace dev;
    __shared__ float mix_tile[kMixRows * 64];
    const int lane = threadIdx.x;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t voice = wm.voice, vc = wm.vc, V = a.V;
    const bool active = wm.active;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };

    const DevOp& oo = a.ops[r.osc];
    const DevOp& ov = a.ops[r.vcf];
    const DevOp& oc = a.ops[r.vca];
    const int plane = a.ops[r.out].aux;
    const bool has_math = r.math >= 0, has_cut = r.trk_cutoff >= 0;
    const uint32_t mflags = has_math ? a.ops[r.math].flags : 0u;
    const float mconst = has_math ? parv(a.ops[r.math], MATH_P_CONST) : 0.0f;
    const float* __restrict__ pitch_track = a.tracks + (size_t)r.trk_pitch * a.t_stride;
    const float* __restrict__ cut_track = a.tracks + (size_t)(has_cut ? r.trk_cutoff : r.trk_env) * a.t_stride;
    const float* __restrict__ env_track = a.tracks + (size_t)r.trk_env * a.t_stride;

    constexpr uint32_t fo = OSC_HAS_CV | OSC_AA | kOscPort;
    OscConst ko;
    ko.sr = oo.sample_rate;
    ko.val = (double)parv(oo, OSC_P_VAL);
    ko.delta = 0.0;
    ko.inv_dt = 0.0f;
    const double hz_scale = 440.0 / ko.sr;
    COsc co;
    co.pos = make_f64(row(oo.state_row + OSC_S_POS_LO), row(oo.state_row + OSC_S_POS_HI));
    co.delta = 0.0;
    bool carried = false, have_pitch = false, have_cut = false;
    uint32_t seen_pitch = 0u, seen_cut = 0u;
    float cv_lane = 0.0f;

    VcfRegs sv;
    const int s0 = ov.state_row;
    sv.f = __uint_as_float(row(s0 + VCF_S_F));
    sv.p = __uint_as_float(row(s0 + VCF_S_P));
    sv.q = __uint_as_float(row(s0 + VCF_S_Q));
    sv.b0 = __uint_as_float(row(s0 + VCF_S_B0 + 0));
    sv.b1 = __uint_as_float(row(s0 + VCF_S_B0 + 1));
    sv.b2 = __uint_as_float(row(s0 + VCF_S_B0 + 2));
    sv.b3 = __uint_as_float(row(s0 + VCF_S_B0 + 3));
    sv.b4 = __uint_as_float(row(s0 + VCF_S_B0 + 4));
    sv.freq = __uint_as_float(row(s0 + VCF_S_FREQ));
    sv.res = __uint_as_float(row(s0 + VCF_S_RES));
    const float vfreq = parv(ov, VCF_P_FREQ), vexp = parv(ov, VCF_P_EXP), vres = vcf_resonance(parv(ov, VCF_P_RES));
    const uint32_t vport = ov.flags & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP);
    if (!has_cut && a.T > 0) vcf_coeffs(sv, vcf_frequency(vfreq, 0.0f, vexp), vres);
    const bool negative = parv(oc, VCA_P_NEG) != 0.0f;

    Emit em = make_emit(a, plane, lane);
    float* extra_row[4] = {nullptr, nullptr, nullptr, nullptr};   // frame rows of the track-fed planes (wave-uniform)
    float* extra_mp[4] = {nullptr, nullptr, nullptr, nullptr};    // ... and their mix partials
    const float* extra_track[4] = {env_track, env_track, env_track, env_track};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (e >= r.n_extra) continue;
        extra_track[e] = a.tracks + (size_t)r.extra_trk[e] * a.t_stride;
        if (a.frames) extra_row[e] = a.frames + (size_t)r.extra_plane[e] * a.plane_stride + wm.wave0;
        if (a.mixpart) extra_mp[e] = a.mixpart + ((size_t)r.extra_plane[e] * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride;
    }

    const uint32_t l32 = (uint32_t)(lane & (kMixRows - 1));
    auto fetch = [&](const float* trk, uint32_t t0) { return trk[min(t0 + l32, a.T - 1)]; };
    float pitch_tile = fetch(pitch_track, 0), cut_tile = fetch(cut_track, 0), env_tile = fetch(env_track, 0);
    float extra_tile[4];
#pragma unroll
    for (int e = 0; e < 4; e++) extra_tile[e] = fetch(extra_track[e], 0);
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const float pitch_next = fetch(pitch_track, t0 + kMixRows), cut_next = fetch(cut_track, t0 + kMixRows), env_next = fetch(env_track, t0 + kMixRows);
        float extra_next[4];
#pragma unroll
        for (int e = 0; e < 4; e++) extra_next[e] = fetch(extra_track[e], t0 + kMixRows);
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        auto sample = [&](int i) {
            const uint32_t pb = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(pitch_tile), i);
            if (!have_pitch || pb != seen_pitch) {  // a new note (scalar test): new increment, carried terms rebuilt
                have_pitch = true;
                seen_pitch = pb;
                const float note = __uint_as_float(pb);
                cv_lane = has_math ? math_step(mflags, note, 0.0f, mconst) : note;
                const double delta = hz_scale * exp2_fast((double)cv_lane + ko.val);
                carried = __builtin_amdgcn_ballot_w64(!(delta < 0.25)) == 0;
                cosc_init(co, co.pos, delta);
            }
            float x;
            if (carried) {
                x = cosc_step<kOscPort>(co);
            } else {  // an increment of a quarter cycle or more somewhere in the wave: the literal per-sample form
                OscRegs g;
                g.pos = co.pos;
                g.sync_last = false;
                g.seen_cv = cv_lane;
                g.seen_delta = co.delta;
                float sine = 0.0f, square = 0.0f, saw = 0.0f;
                osc_step(fo, g, ko, cv_lane, 0.0f, sine, square, saw);
                x = kOscPort == OSC_OUT_SINE ? sine : (kOscPort == OSC_OUT_SQUARE ? square : saw);
                co.pos = g.pos;
            }
            if (has_cut) {
                const uint32_t cb = (uint32_t)__builtin_amdgcn_readlane(__float_as_int(cut_tile), i);
                if (!have_cut || cb != seen_cut) {
                    have_cut = true;
                    seen_cut = cb;
                    vcf_coeffs(sv, vcf_frequency(vfreq, __uint_as_float(cb), vexp), vres);
                }
            }
            float lp, bp, hp;
            vcf_step<true>(sv, x, lp, bp, hp);
            const float y = vport == VCF_OUT_LP ? lp : (vport == VCF_OUT_BP ? bp : hp);
            const float env = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(env_tile), i));
            const bool cv_pos = (uint32_t)(__float_as_int(env) - 1) < 0x7f800000u;  // env > 0.0 on the scalar unit
            const float o = (negative || cv_pos) ? y * env : 0.0f;
            emit_put<kOut>(em, mix_tile, o, i, V);
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (e < r.n_extra && extra_row[e]) {
                    const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(extra_tile[e]), i));
                    __builtin_nontemporal_store(v, &extra_row[e][em.lane_c]);
                    extra_row[e] += V;
                }
        };
        if (n == kMixRows) {
#pragma unroll 4
            for (int i = 0; i < kMixRows; i++) sample(i);
        } else {
            for (int i = 0; i < n; i++) sample(i);
        }
        emit_flush<kOut>(em, mix_tile, t0, n);
#pragma unroll
        for (int e = 0; e < 4; e++)  // every voice carries the same sample: the wave's partial is count x sample
            if (e < r.n_extra && extra_mp[e] && lane < n) extra_mp[e][t0 + lane] = (float)em.n_active * extra_tile[e];
        pitch_tile = pitch_next;
        cut_tile = cut_next;
        env_tile = env_next;
#pragma unroll
        for (int e = 0; e < 4; e++) extra_tile[e] = extra_next[e];
    }
    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(oo.state_row + OSC_S_POS_LO, f64_lo(co.pos));
        put(oo.state_row + OSC_S_POS_HI, f64_hi(co.pos));
        put(oo.state_row + OSC_S_SYNC_LAST, 0u);  // sync unconnected: `last` follows the constant 0.0 input
        put(s0 + VCF_S_F, __float_as_uint(sv.f));
        put(s0 + VCF_S_P, __float_as_uint(sv.p));
        put(s0 + VCF_S_Q, __float_as_uint(sv.q));
        put(s0 + VCF_S_B0 + 0, __float_as_uint(sv.b0));
        put(s0 + VCF_S_B0 + 1, __float_as_uint(sv.b1));
        put(s0 + VCF_S_B0 + 2, __float_as_uint(sv.b2));
        put(s0 + VCF_S_B0 + 3, __float_as_uint(sv.b3));
        put(s0 + VCF_S_B0 + 4, __float_as_uint(sv.b4));
        put(s0 + VCF_S_FREQ, __float_as_uint(sv.freq));
        put(s0 + VCF_S_RES, __float_as_uint(sv.res));
    }
}

// ---- fused 2-operator FM with a z^-1 feedback edge (patch P2's shape, buffer_size == 1) --------------------
//   MATH_FB(in1 = OSC_M.sine delayed by one sample) -> OSC_M.cv ; OSC_M.sine -> MATH_IDX -> OSC_C.cv ; OSC_C.sine -> out
// The broken edge is a one-sample delay, so the fed-back sine lives in a VGPR ("in-register recurrence").
// Both oscillators have CV: 2^x and sin per sample per operator (oscillator.rs:45,132-133).  The modulator of
// sample t+1 depends only on its own sine of sample t, so it runs one sample ahead of the carrier.
template <bool kExact, int kOut>
__global__ __launch_bounds__(64) void render_fm_pair(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    __shared__ float mix_tile[kMixRows * 64];
    const int lane = threadIdx.x;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t voice = wm.voice, vc = wm.vc, V = a.V;
    const bool active = wm.active;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };

    const DevOp& ofb = a.ops[r.adsr];    // MATH on the feedback path   (roles reuse the ChainRoles slots)
    const DevOp& om = a.ops[r.osc_l];    // modulator
    const DevOp& oix = a.ops[r.vca];     // MATH scaling the modulation index
    const DevOp& ocr = a.ops[r.osc_a];   // carrier
    const int plane = a.ops[r.out].aux;
    const int ring_row = r.track;        // the z^-1 ring: one state row

    constexpr uint32_t fo = OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE | (kExact ? OSC_EXACT : 0u);
    OscRegs sm, sc;
    OscConst km, kc;
    sm.pos = make_f64(row(om.state_row + OSC_S_POS_LO), row(om.state_row + OSC_S_POS_HI));
    sm.sync_last = row(om.state_row + OSC_S_SYNC_LAST) != 0;
    sc.pos = make_f64(row(ocr.state_row + OSC_S_POS_LO), row(ocr.state_row + OSC_S_POS_HI));
    sc.sync_last = row(ocr.state_row + OSC_S_SYNC_LAST) != 0;
    km.sr = om.sample_rate;
    km.val = (double)parv(om, OSC_P_VAL);
    km.delta = 0.0;
    km.inv_dt = 0.0f;
    kc = km;
    kc.sr = ocr.sample_rate;
    kc.val = (double)parv(ocr, OSC_P_VAL);
    // both MATH modules are Multiply by a constant (host-checked): in1 * constant (math.rs:152)
    const float c_fb = parv(ofb, MATH_P_CONST), c_ix = parv(oix, MATH_P_CONST);
    float fed = __uint_as_float(row(ring_row));  // OSC_M.sine of the previous tick (0.0 before the first)

    Emit em = make_emit(a, plane, lane);
    float sq = 0.0f, sw = 0.0f;
    float sine_m = 0.0f;
    double pos_m = sm.pos;  // modulator phase after exactly t samples (the loop runs it one sample ahead)
    if (a.T > 0) osc_step(fo, sm, km, fed * c_fb, 0.0f, sine_m, sq, sw);  // modulator of sample 0
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        for (int i = 0; i < n; i++) {
            const float cur = sine_m;  // OSC_M.sine[t]: feeds the carrier now and, through the z^-1 ring, the modulator of t+1
            float out = 0.0f;
            osc_step(fo, sc, kc, cur * c_ix, 0.0f, out, sq, sw);      // carrier of sample t
            pos_m = sm.pos;
            osc_step(fo, sm, km, cur * c_fb, 0.0f, sine_m, sq, sw);   // modulator of sample t+1 (independent of the carrier)
            fed = cur;
            emit_put<kOut>(em, mix_tile, out, i, V);
        }
        emit_flush<kOut>(em, mix_tile, t0, n);
    }
    sm.pos = pos_m;  // drop the look-ahead step
    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(om.state_row + OSC_S_POS_LO, f64_lo(sm.pos));
        put(om.state_row + OSC_S_POS_HI, f64_hi(sm.pos));
        put(om.state_row + OSC_S_SYNC_LAST, 0u);
        put(ocr.state_row + OSC_S_POS_LO, f64_lo(sc.pos));
        put(ocr.state_row + OSC_S_POS_HI, f64_hi(sc.pos));
        put(ocr.state_row + OSC_S_SYNC_LAST, 0u);
        put(ring_row, __float_as_uint(fed));
    }
}

// ---- mix-down, passes 2 and 3: mix[c][i] = sum over waves of mixpart[plane(c)][w][i] ---------------------------
// Deterministic (fixed order, no atomics).  Pass 2 splits the waves into kMixSplit groups so that enough loads
// are in flight to stream the partials at HBM rate: block (x, y) sums group y for 256 consecutive samples into
// mixgroup[plane][y][i].  Pass 3 adds the kMixSplit group sums and fans planes out to channels.
constexpr uint32_t kMixSplit = 16;

struct MixArgs {
    const float* mixpart;   // [planes][n_waves][T]
    float* mixgroup;        // [planes][kMixSplit][T]
    float* mix;             // [channels][T]
    uint32_t T, n_waves, n_channels, n_planes;
    int32_t channel_plane[8];
};

__global__ __launch_bounds__(256) void mix_reduce_groups(MixArgs m)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m.T) return;
    const uint32_t per = (m.n_waves + kMixSplit - 1) / kMixSplit;
    const uint32_t w0 = blockIdx.y * per, w1 = min(m.n_waves, w0 + per);
    for (uint32_t plane = 0; plane < m.n_planes; plane++) {
        const float* p = m.mixpart + (size_t)plane * m.n_waves * m.T + i;
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // fixed 8-way split: deterministic, 8 loads in flight per thread
        uint32_t w = w0;
        for (; w + 8 <= w1; w += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] += p[(size_t)(w + k) * m.T];
        }
        for (; w < w1; w++) s[0] += p[(size_t)w * m.T];
        m.mixgroup[((size_t)plane * kMixSplit + blockIdx.y) * m.T + i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
}

__global__ __launch_bounds__(256) void mix_reduce_final(MixArgs m)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m.T) return;
    for (uint32_t c = 0; c < m.n_channels; c++) {
        const int plane = m.channel_plane[c];
        float s = 0.0f;
        if (plane >= 0)
            for (uint32_t y = 0; y < kMixSplit; y++) s += m.mixgroup[((size_t)plane * kMixSplit + y) * m.T + i];
        m.mix[(size_t)c * m.T + i] = s;
    }
}

__global__ void fill_zero(float* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

// ====================================================================================================
// host side
// ====================================================================================================

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                               \
            return SRACK_ERR_DEVICE;                                                                    \
        }                                                                                               \
    } while (0)

struct DevProg {  // device copy of one FlatProgram
    DevOp* d_ops = nullptr;
    uint32_t* d_table = nullptr;
    float* d_rings = nullptr;
    uint32_t* d_seqtab = nullptr;
    void release()
    {
        (void)hipFree(d_ops);
        (void)hipFree(d_table);
        (void)hipFree(d_rings);
        (void)hipFree(d_seqtab);
        d_seqtab = nullptr;
        d_ops = nullptr;
        d_table = nullptr;
        d_rings = nullptr;
    }
};

struct DeviceState {
    DevProg voice;
    std::vector<DevProg> ctl;  // one per control stage
    KernelArgs* d_stage_slots = nullptr;  // [launches][stages] argument blocks of the staged control pipeline
    size_t stage_slots_cap = 0;
    std::vector<KernelArgs> h_stage_slots;
    float* d_mixpart = nullptr;
    size_t mixpart_bytes = 0;
    float* d_mixgroup = nullptr;
    size_t mixgroup_bytes = 0;
    float* d_tracks = nullptr;
    size_t tracks_bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timings;  // (start, stop) pairs of the dominant kernel, not yet read
    std::vector<hipEvent_t> pool;
    const char* kernel_name = "";
    hipStream_t ctl_stream = nullptr;      // the control program runs ahead of the voice kernels on its own stream
    hipEvent_t ev_begin = nullptr;         // render start on the caller's stream
    std::vector<hipEvent_t> ev_chunk;      // control chunk k finished
};

void device_release(DeviceState* d)
{
    if (!d) return;
    d->voice.release();
    for (DevProg& c : d->ctl) c.release();
    (void)hipFree(d->d_stage_slots);
    (void)hipFree(d->d_mixpart);
    (void)hipFree(d->d_mixgroup);
    (void)hipFree(d->d_tracks);
    for (auto& p : d->timings) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : d->pool) (void)hipEventDestroy(e);
    for (auto e : d->ev_chunk) (void)hipEventDestroy(e);
    if (d->ev_begin) (void)hipEventDestroy(d->ev_begin);
    if (d->ctl_stream) (void)hipStreamDestroy(d->ctl_stream);
    delete d;
}

PatchHandle::~PatchHandle() { device_release(dev); }

int ensure_program(PatchHandle& h, uint32_t flags)
{
    if (h.prog_valid && h.prog_graph_revision == h.graph.revision && h.prog_voices_revision == h.voices_revision && h.prog_flags == flags)
        return SRACK_OK;
    int rc = flatten(h.graph, h.n_voices, h.overrides, flags, h.prog);
    if (rc != SRACK_OK) return rc;
    h.prog_valid = true;
    h.prog_graph_revision = h.graph.revision;
    h.prog_voices_revision = h.voices_revision;
    h.prog_flags = flags;
    h.samples_rendered = 0;
    if (h.dev) {  // device copies are rebuilt lazily by device_render
        device_release(h.dev);
        h.dev = nullptr;
    }
    return SRACK_OK;
}

// rings[row][v] = init[row] for every voice (row = ring * B + sample)
__global__ void ring_fill(float* rings, const float* init, uint32_t V)
{
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < V) rings[(size_t)blockIdx.y * V + v] = init[blockIdx.y];
}

static int upload_one(const FlatProgram& P, DevProg& d)
{
    if (!P.ops.empty()) {
        HIP_TRY(hipMalloc(&d.d_ops, sizeof(DevOp) * P.ops.size()));
        HIP_TRY(hipMemcpy(d.d_ops, P.ops.data(), sizeof(DevOp) * P.ops.size(), hipMemcpyHostToDevice));
    }
    if (!P.table.empty()) {
        HIP_TRY(hipMalloc(&d.d_table, sizeof(uint32_t) * P.table.size()));
        HIP_TRY(hipMemcpy(d.d_table, P.table.data(), sizeof(uint32_t) * P.table.size(), hipMemcpyHostToDevice));
    }
    if (!P.seqtab.empty()) {
        HIP_TRY(hipMalloc(&d.d_seqtab, sizeof(uint32_t) * P.seqtab.size()));
        HIP_TRY(hipMemcpy(d.d_seqtab, P.seqtab.data(), sizeof(uint32_t) * P.seqtab.size(), hipMemcpyHostToDevice));
    }
    if (P.hdr.n_rings > 0) {
        size_t bytes = sizeof(float) * (size_t)P.hdr.n_rings * (size_t)P.hdr.buffer_size * P.n_voices;
        HIP_TRY(hipMalloc(&d.d_rings, bytes));
        if (P.ring_init.empty()) {
            HIP_TRY(hipMemset(d.d_rings, 0, bytes));  // AudioBuffer::new fills 0.0 (synth.rs:31-33)
        } else {  // a loaded patch: every voice starts from the saved block
            float* d_init = nullptr;
            HIP_TRY(hipMalloc(&d_init, sizeof(float) * P.ring_init.size()));
            HIP_TRY(hipMemcpy(d_init, P.ring_init.data(), sizeof(float) * P.ring_init.size(), hipMemcpyHostToDevice));
            const uint32_t n_rows = (uint32_t)P.ring_init.size();
            hipLaunchKernelGGL(ring_fill, dim3((P.n_voices + 255) / 256, n_rows), dim3(256), 0, 0, d.d_rings, d_init, P.n_voices);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(d_init));
        }
    }
    return SRACK_OK;
}

static int upload_program(PatchHandle& h)
{
    h.dev = new DeviceState();
    int rc = upload_one(h.prog.voice, h.dev->voice);
    h.dev->ctl.resize(h.prog.ctl.size());
    for (size_t s = 0; s < h.prog.ctl.size() && rc == SRACK_OK; s++) rc = upload_one(h.prog.ctl[s], h.dev->ctl[s]);
    return rc;
}

static int grow(float*& p, size_t& have, size_t need)
{
    if (need <= have) return SRACK_OK;
    (void)hipFree(p);
    p = nullptr;
    have = 0;
    HIP_TRY(hipMalloc(&p, need));
    have = need;
    return SRACK_OK;
}

// ---- fused-kernel dispatch (template parameters from runtime port flags) ---------------------------
template <uint32_t A, uint32_t F, bool E, int O>
static void launch_fused4(bool track, const KernelArgs& ka, const ChainRoles& roles, const CtlWork& co, dim3 grid, hipStream_t st)
{
    if (track)
        hipLaunchKernelGGL((render_voice_chain_track<A, F, E, O>), grid, dim3(64), 0, st, ka, roles, co);
    else
        hipLaunchKernelGGL((render_voice_chain<A, OSC_OUT_SQUARE, F, E, O>), grid, dim3(64), 0, st, ka, roles);
}

template <uint32_t A, uint32_t F>
static void launch_fused3(bool exact, int out_mode, bool track, const KernelArgs& ka, const ChainRoles& roles, const CtlWork& co, dim3 grid, hipStream_t st)
{
    if (exact)  // exact mode is the validation flavour: one instantiation, output mode decided at run time
        launch_fused4<A, F, true, 0>(track, ka, roles, co, grid, st);
    else if (out_mode == 3)
        launch_fused4<A, F, false, 3>(track, ka, roles, co, grid, st);
    else if (out_mode == 1)
        launch_fused4<A, F, false, 1>(track, ka, roles, co, grid, st);
    else if (out_mode == 2)
        launch_fused4<A, F, false, 2>(track, ka, roles, co, grid, st);
    else  // neither frames nor mix requested: the render only advances the voice state
        launch_fused4<A, F, false, 0>(track, ka, roles, co, grid, st);
}

template <uint32_t A>
static void launch_fused2(uint32_t vcf_port, bool exact, int out_mode, bool track, const KernelArgs& ka, const ChainRoles& roles, const CtlWork& co, dim3 grid,
                          hipStream_t st)
{
    if (vcf_port == VCF_OUT_LP)
        launch_fused3<A, VCF_OUT_LP>(exact, out_mode, track, ka, roles, co, grid, st);
    else if (vcf_port == VCF_OUT_BP)
        launch_fused3<A, VCF_OUT_BP>(exact, out_mode, track, ka, roles, co, grid, st);
    else
        launch_fused3<A, VCF_OUT_HP>(exact, out_mode, track, ka, roles, co, grid, st);
}

static void launch_fused(uint32_t osc_port, uint32_t vcf_port, bool exact, int out_mode, bool track, const KernelArgs& ka, const ChainRoles& roles,
                         const CtlWork& co, dim3 grid, hipStream_t st)
{
    if (osc_port == OSC_OUT_SAW)
        launch_fused2<OSC_OUT_SAW>(vcf_port, exact, out_mode, track, ka, roles, co, grid, st);
    else if (osc_port == OSC_OUT_SQUARE)
        launch_fused2<OSC_OUT_SQUARE>(vcf_port, exact, out_mode, track, ka, roles, co, grid, st);
    else
        launch_fused2<OSC_OUT_SINE>(vcf_port, exact, out_mode, track, ka, roles, co, grid, st);
}

template <uint32_t kPort>
static void launch_seq2(int out_mode, const KernelArgs& ka, const SeqRoles& r, dim3 grid, hipStream_t st)
{
    if (out_mode == 3)
        hipLaunchKernelGGL((render_voice_chain_seq<kPort, 3>), grid, dim3(64), 0, st, ka, r);
    else if (out_mode == 1)
        hipLaunchKernelGGL((render_voice_chain_seq<kPort, 1>), grid, dim3(64), 0, st, ka, r);
    else
        hipLaunchKernelGGL((render_voice_chain_seq<kPort, 0>), grid, dim3(64), 0, st, ka, r);
}

static void launch_seq(uint32_t osc_port, int out_mode, const KernelArgs& ka, const SeqRoles& r, dim3 grid, hipStream_t st)
{
    if (osc_port == OSC_OUT_SAW)
        launch_seq2<OSC_OUT_SAW>(out_mode, ka, r, grid, st);
    else if (osc_port == OSC_OUT_SQUARE)
        launch_seq2<OSC_OUT_SQUARE>(out_mode, ka, r, grid, st);
    else
        launch_seq2<OSC_OUT_SINE>(out_mode, ka, r, grid, st);
}

static void launch_interp(const FlatProgram& P, const KernelArgs& ka, hipStream_t st)
{
    size_t lds = ((size_t)P.hdr.n_rows + 2 + (size_t)P.hdr.n_tracks + (size_t)P.hdr.n_slots * P.hdr.tile) * 256;  // + zero, trash and track rows
    if (getenv("SRACK_DEBUG_OCC")) {  // tools/: what the runtime says about resident workgroups per CU for this LDS size
        int n = -1;
        hipError_t e = (P.render_flags & SRACK_RENDER_EXACT_OSC) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_interp<true>, 64, lds)
                                                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_interp<false>, 64, lds);
        fprintf(stderr, "[srack] render_interp lds=%zu B/wave occupancy=%d workgroups/CU (%s) waves=%u\n", lds, n, hipGetErrorString(e), ka.n_waves);
    }
    if (P.render_flags & SRACK_RENDER_EXACT_OSC)
        hipLaunchKernelGGL(render_interp<true>, dim3(ka.n_waves), dim3(64), lds, st, ka);
    else
        hipLaunchKernelGGL(render_interp<false>, dim3(ka.n_waves), dim3(64), lds, st, ka);
}

static void launch_ctl(const FlatProgram& Cp, const KernelArgs& kc, hipStream_t st)
{
    if (Cp.fused == FUSED_CTL_GATE_ENV) {
        CtlWork w{kc.ops, kc.table, kc.frames + (size_t)Cp.ops[2].aux * kc.plane_stride, kc.T,
                  Cp.ops[0].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW)};
        hipLaunchKernelGGL(render_ctl_gate_env, dim3(1), dim3(64), 0, st, w);
    } else if (Cp.fused == FUSED_FM_PAIR) {
        ChainRoles roles{};
        roles.adsr = 1;
        roles.osc_l = 2;
        roles.vca = 4;
        roles.osc_a = 5;
        roles.out = 6;
        roles.track = Cp.ops[0].aux;
        if (Cp.render_flags & SRACK_RENDER_EXACT_OSC)
            hipLaunchKernelGGL((render_fm_pair<true, 0>), dim3(1), dim3(64), 0, st, kc, roles);
        else
            hipLaunchKernelGGL((render_fm_pair<false, 1>), dim3(1), dim3(64), 0, st, kc, roles);
    } else {
        launch_interp(Cp, kc, st);
    }
}

// How a render is scheduled.  Without a control program: one launch of the voice kernel.  With one: the
// render is cut into chunks; control chunk k (one wave, a latency chain) runs on its own stream and voice
// chunk k waits only for it, so all but the first control chunk hide behind voice kernels of earlier chunks.
int device_render(PatchHandle& h, uint32_t n_samples, float* d_frames, float* d_mix, uint32_t flags, void* stream)
{
    int rc = ensure_program(h, flags);
    if (rc != SRACK_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const FlatProgram& P = h.prog.voice;
    const uint32_t V = P.n_voices, T = n_samples, C = (uint32_t)P.hdr.n_channels;
    if (T == 0) return SRACK_OK;
    if (!h.dev) {
        rc = upload_program(h);
        if (rc != SRACK_OK) return rc;
    }
    DeviceState* d = h.dev;
    // Voices per wave.  A full wave (64) is right whenever there are enough voices to give every SIMD work.
    // With few voices, half- or quarter-filled waves double / quadruple the number of waves: a VALU instruction
    // costs the same for 16 lanes as for 64, so this only pays while SIMDs would otherwise sit idle (VALU-bound
    // kernels: up to one wave per SIMD) or while waves are latency-bound (FM pair, interpreter: up to four).
    const uint32_t kSimds = 1024;
    uint32_t want_waves = kSimds;
    if (const char* e = getenv("SRACK_WANT_WAVES")) want_waves = (uint32_t)atoi(e);  // tuning knob (tools/): waves to aim for
    uint32_t lanes = 64;
    while (lanes > 16 && (V + lanes - 1) / lanes * 2 <= want_waves) lanes >>= 1;
    const uint32_t n_waves = (V + lanes - 1) / lanes;

    if (P.hdr.n_planes == 0) {  // nothing reaches the output: silence (output.rs:55)
        if (d_mix) {
            size_t n = (size_t)C * T;
            hipLaunchKernelGGL(fill_zero, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_mix, n);
        }
        h.samples_rendered += T;
        return SRACK_OK;
    }
    if (d_mix && (rc = grow(d->d_mixpart, d->mixpart_bytes, sizeof(float) * (size_t)P.hdr.n_planes * n_waves * T)) != SRACK_OK) return rc;
    if (d_mix && (rc = grow(d->d_mixgroup, d->mixgroup_bytes, sizeof(float) * (size_t)P.hdr.n_planes * kMixSplit * T)) != SRACK_OK) return rc;

    const bool has_ctl = h.prog.n_tracks > 0;
    // Two ways to overlap the control program with the voice kernels:
    //  co-scheduled (fused track kernel + fused gate-envelope control program): voice launch k carries one extra
    //    block that computes the track of chunk k+1; everything stays on the caller's stream.
    //  two streams (any other combination): control chunks run on a private stream, voice chunk k waits on event k.
    //    This overlaps only while the two streams map to different hardware queues (GPU_MAX_HW_QUEUES, default 4,
    //    shared with the host's other streams): measured 15 ms -> 20 ms per step once RCCL's streams are alive.
    const uint32_t n_stages = (uint32_t)h.prog.ctl.size();
    const bool co_ctl = has_ctl && P.fused == FUSED_VOICE_CHAIN_TRACK && n_stages == 1 && h.prog.ctl[0].fused == FUSED_CTL_GATE_ENV && h.prog.n_tracks == 1;
    auto ctl_work = [&](uint32_t t_off, uint32_t len) {
        const FlatProgram& Cp = h.prog.ctl[0];
        return CtlWork{d->ctl[0].d_ops, d->ctl[0].d_table, d->d_tracks + (size_t)Cp.ops[2].aux * T + t_off, len,
                       Cp.ops[0].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW)};
    };
    // chunk schedule: short first chunks (only control chunk 0 is exposed), doubling up to kChunkMax
    // With a control pipeline of depth L the first voice chunk waits for L + 1 control launches: those stay short.
    constexpr uint32_t kChunkFirst = 1024, kChunkMax = 8192;
    uint32_t max_lag = 0;
    for (int lag : h.prog.ctl_lag) max_lag = std::max(max_lag, (uint32_t)lag);
    std::vector<std::pair<uint32_t, uint32_t>> chunks;       // (t_off, len)
    if (has_ctl) {
        uint32_t k = 0;
        for (uint32_t t_off = 0, len = kChunkFirst; t_off < T; t_off += len, k++) {
            if (k > max_lag) len = std::min(len * 2, kChunkMax);
            len = std::min(len, T - t_off);
            chunks.emplace_back(t_off, len);
        }
    } else {
        chunks.emplace_back(0u, T);
    }
    const uint32_t n_chunks = (uint32_t)chunks.size();
    auto get_event = [&](hipEvent_t& e) -> int {
        if (!d->pool.empty()) {
            e = d->pool.back();
            d->pool.pop_back();
            return SRACK_OK;
        }
        HIP_TRY(hipEventCreate(&e));
        return SRACK_OK;
    };

    if (has_ctl && (rc = grow(d->d_tracks, d->tracks_bytes, sizeof(float) * (size_t)h.prog.n_tracks * T)) != SRACK_OK) return rc;
    if (co_ctl) {  // chunk 0's track: the only control work that is not hidden (1024 samples, ~0.15 ms)
        hipLaunchKernelGGL(render_ctl_gate_env, dim3(1), dim3(64), 0, st, ctl_work(chunks[0].first, chunks[0].second));
        HIP_TRY(hipGetLastError());
    } else if (has_ctl) {
        if (!d->ctl_stream) HIP_TRY(hipStreamCreateWithFlags(&d->ctl_stream, hipStreamNonBlocking));
        if (!d->ev_begin) HIP_TRY(hipEventCreateWithFlags(&d->ev_begin, hipEventDisableTiming));
        while (d->ev_chunk.size() < n_chunks) {
            hipEvent_t e;
            HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            d->ev_chunk.push_back(e);
        }
        // the track buffer may still be read by the previous render on `st`: start after it
        HIP_TRY(hipEventRecord(d->ev_begin, st));
        HIP_TRY(hipStreamWaitEvent(d->ctl_stream, d->ev_begin, 0));
        auto stage_args = [&](uint32_t s, uint32_t k) {
            const uint32_t t_off = chunks[k].first, len = chunks[k].second;
            KernelArgs kc{};
            kc.ops = d->ctl[s].d_ops;
            kc.prog = h.prog.ctl[s].hdr;
            kc.table = d->ctl[s].d_table;
            kc.rings = d->ctl[s].d_rings;
            kc.seqtab = d->ctl[s].d_seqtab;
            kc.frames = d->d_tracks + t_off;  // the control program's planes are the tracks: [n_tracks][T][1]
            kc.tracks = d->d_tracks + t_off;  // ... and later stages read earlier stages' tracks from the same buffer
            kc.plane_stride = T;
            kc.t_stride = T;
            kc.V = 1;
            kc.T = len;
            kc.n_waves = 1;
            kc.lanes = 64;
            kc.n0 = h.samples_rendered + t_off;
            return kc;
        };
        const bool staged = h.prog.ctl[0].fused == FUSED_NONE;  // the interpreter: all stages side by side in one launch
        if (staged) {
            // launch j runs unit s on chunk j - lag[s]; chunk c is complete after launch c + max_lag
            const uint32_t n_launch = n_chunks + max_lag;
            d->h_stage_slots.assign((size_t)n_launch * n_stages, KernelArgs{});
            size_t lds = 0;
            for (uint32_t s = 0; s < n_stages; s++) {
                const DevProgram& H = h.prog.ctl[s].hdr;
                lds = std::max(lds, ((size_t)H.n_rows + 2 + (size_t)H.n_tracks + (size_t)H.n_slots * H.tile) * 256);
                for (uint32_t k = 0; k < n_chunks; k++) {
                    KernelArgs& slot = d->h_stage_slots[(size_t)(k + (uint32_t)h.prog.ctl_lag[s]) * n_stages + s];
                    slot = stage_args(s, k);
                    slot.block0 = s;  // the stage's one wave is wave 0 of its program
                }
            }
            const size_t bytes = sizeof(KernelArgs) * d->h_stage_slots.size();
            if (bytes > d->stage_slots_cap) {
                (void)hipFree(d->d_stage_slots);
                d->d_stage_slots = nullptr;
                d->stage_slots_cap = 0;
                HIP_TRY(hipMalloc(&d->d_stage_slots, bytes));
                d->stage_slots_cap = bytes;
            }
            HIP_TRY(hipMemcpyAsync(d->d_stage_slots, d->h_stage_slots.data(), bytes, hipMemcpyHostToDevice, d->ctl_stream));
            for (uint32_t j = 0; j < n_launch; j++) {
                const KernelArgs* slots = d->d_stage_slots + (size_t)j * n_stages;
                if (flags & SRACK_RENDER_EXACT_OSC)
                    hipLaunchKernelGGL(render_interp_stages<true>, dim3(n_stages), dim3(64), lds, d->ctl_stream, slots);
                else
                    hipLaunchKernelGGL(render_interp_stages<false>, dim3(n_stages), dim3(64), lds, d->ctl_stream, slots);
                HIP_TRY(hipGetLastError());
                if (j >= max_lag) HIP_TRY(hipEventRecord(d->ev_chunk[j - max_lag], d->ctl_stream));
            }
        } else {
            for (uint32_t k = 0; k < n_chunks; k++) {
                launch_ctl(h.prog.ctl[0], stage_args(0, k), d->ctl_stream);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(d->ev_chunk[k], d->ctl_stream));
            }
        }
    }

    ChainRoles roles{};
    uint32_t osc_port = 0, vcf_port = 0;
    const bool fused = P.fused == FUSED_VOICE_CHAIN || P.fused == FUSED_VOICE_CHAIN_TRACK;
    const bool track = P.fused == FUSED_VOICE_CHAIN_TRACK;
    if (fused) {
        for (int i = 0; i < (int)P.ops.size(); i++) {
            const DevOp& op = P.ops[(size_t)i];
            if (op.kind == OP_VCF) { roles.vcf = i; vcf_port = op.flags & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP); }
            if (op.kind == OP_ADSR) roles.adsr = i;
            if (op.kind == OP_VCA) roles.vca = i;
            if (op.kind == OP_OUT) roles.out = i;
            if (op.kind == OP_VCA && track) roles.track = P.hdr.track_id[op.in_slot[1] - kTrackSlot];
        }
        const Graph& g = h.graph;
        roles.osc_a = P.op_of_module[(size_t)g.modules[(size_t)P.ops[(size_t)roles.vcf].module].in[0].src];
        if (!track) roles.osc_l = P.op_of_module[(size_t)g.modules[(size_t)P.ops[(size_t)roles.adsr].module].in[0].src];
        osc_port = P.ops[(size_t)roles.osc_a].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
        d->kernel_name = track ? "render_voice_chain_track" : "render_voice_chain";
    } else {
        d->kernel_name = "render_interp";
    }

    const bool seq_chain = P.fused == FUSED_VOICE_CHAIN_SEQ;
    SeqRoles seq{};
    uint32_t seq_port = 0;
    if (seq_chain) {
        seq.math = seq.trk_cutoff = -1;
        auto track_row = [&](int slot) { return P.hdr.track_id[slot - kTrackSlot]; };
        for (int i = 0; i < (int)P.ops.size(); i++) {
            const DevOp& op = P.ops[(size_t)i];
            if (op.kind == OP_MATH) { seq.math = i; seq.trk_pitch = track_row(op.in_slot[0]); }
            if (op.kind == OP_OSC) { seq.osc = i; seq_port = op.flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW); }
            if (op.kind == OP_VCF) { seq.vcf = i; if (op.flags & VCF_HAS_CV) seq.trk_cutoff = track_row(op.in_slot[1]); }
            if (op.kind == OP_VCA) { seq.vca = i; seq.trk_env = track_row(op.in_slot[1]); }
            if (op.kind == OP_OUT) {
                if (op.in_slot[0] >= kTrackSlot) {
                    seq.extra_plane[seq.n_extra] = op.aux;
                    seq.extra_trk[seq.n_extra++] = track_row(op.in_slot[0]);
                } else {
                    seq.out = i;
                }
            }
        }
        if (seq.math < 0) seq.trk_pitch = track_row(P.ops[(size_t)seq.osc].in_slot[0]);
        d->kernel_name = "render_voice_chain_seq";
    }
    const bool fm_pair = P.fused == FUSED_FM_PAIR;
    if (fm_pair) {  // op order fixed by the matcher: DELAY_RD, MATH_FB, OSC_M, DELAY_WR, MATH_IDX, OSC_C, OUT
        roles.adsr = 1;
        roles.osc_l = 2;
        roles.vca = 4;
        roles.osc_a = 5;
        roles.out = 6;
        roles.track = P.ops[0].aux;  // the ring's state row
        d->kernel_name = "render_fm_pair";
    }
    for (uint32_t k = 0; k < n_chunks; k++) {
        const uint32_t t_off = chunks[k].first, len = chunks[k].second;
        KernelArgs ka{};
        ka.ops = d->voice.d_ops;
        ka.prog = P.hdr;
        ka.table = d->voice.d_table;
        ka.rings = d->voice.d_rings;
        ka.seqtab = d->voice.d_seqtab;
        ka.frames = d_frames ? d_frames + (size_t)t_off * V : nullptr;
        ka.mixpart = d_mix ? d->d_mixpart + t_off : nullptr;
        ka.tracks = has_ctl ? d->d_tracks + t_off : nullptr;
        ka.plane_stride = (uint64_t)T * V;
        ka.t_stride = T;
        ka.V = V;
        ka.T = len;
        ka.n_waves = n_waves;
        ka.lanes = lanes;
        ka.n0 = h.samples_rendered + t_off;
        CtlWork co{};
        if (co_ctl && k + 1 < n_chunks) {  // this launch's block 0 prepares the next chunk's track
            co = ctl_work(chunks[k + 1].first, chunks[k + 1].second);
            ka.block0 = 1;
        }
        if (has_ctl && !co_ctl) HIP_TRY(hipStreamWaitEvent(st, d->ev_chunk[k], 0));
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if ((rc = get_event(e0)) != SRACK_OK || (rc = get_event(e1)) != SRACK_OK) return rc;
        HIP_TRY(hipEventRecord(e0, st));
        if (fused) {
            const int out_mode = (ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0);
            launch_fused(osc_port, vcf_port, (flags & SRACK_RENDER_EXACT_OSC) != 0, out_mode, track, ka, roles, co, dim3(n_waves + ka.block0), st);
        } else if (seq_chain) {
            const int out_mode = (ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0);
            launch_seq(seq_port, out_mode, ka, seq, dim3(n_waves), st);
        } else if (fm_pair) {
            const int out_mode = (ka.frames ? 1 : 0) | (ka.mixpart ? 2 : 0);
            if (flags & SRACK_RENDER_EXACT_OSC)
                hipLaunchKernelGGL((render_fm_pair<true, 0>), dim3(n_waves), dim3(64), 0, st, ka, roles);
            else if (out_mode == 3)
                hipLaunchKernelGGL((render_fm_pair<false, 3>), dim3(n_waves), dim3(64), 0, st, ka, roles);
            else if (out_mode == 1)
                hipLaunchKernelGGL((render_fm_pair<false, 1>), dim3(n_waves), dim3(64), 0, st, ka, roles);
            else if (out_mode == 2)
                hipLaunchKernelGGL((render_fm_pair<false, 2>), dim3(n_waves), dim3(64), 0, st, ka, roles);
            else
                hipLaunchKernelGGL((render_fm_pair<false, 0>), dim3(n_waves), dim3(64), 0, st, ka, roles);
        } else {
            launch_interp(P, ka, st);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(e1, st));
        d->timings.emplace_back(e0, e1);
        if (d->timings.size() > 4096) {  // nobody is reading them: recycle the oldest
            d->pool.push_back(d->timings.front().first);
            d->pool.push_back(d->timings.front().second);
            d->timings.erase(d->timings.begin());
        }
    }

    if (d_mix) {
        MixArgs m{};
        m.mixpart = d->d_mixpart;
        m.mixgroup = d->d_mixgroup;
        m.mix = d_mix;
        m.T = T;
        m.n_waves = n_waves;
        m.n_channels = C;
        m.n_planes = (uint32_t)P.hdr.n_planes;
        for (int c = 0; c < 8; c++) m.channel_plane[c] = P.hdr.channel_plane[c];
        hipLaunchKernelGGL(mix_reduce_groups, dim3((T + 255) / 256, kMixSplit), dim3(256), 0, st, m);
        hipLaunchKernelGGL(mix_reduce_final, dim3((T + 255) / 256), dim3(256), 0, st, m);
        HIP_TRY(hipGetLastError());
    }
    h.samples_rendered += T;
    return SRACK_OK;
}

int device_kernel_ms(PatchHandle& h, double* avg_ms, int* n_launches, int reset)
{
    double total = 0.0;
    int n = 0;
    if (h.dev) {
        for (auto& p : h.dev->timings) {
            HIP_TRY(hipEventSynchronize(p.second));
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, p.first, p.second));
            total += ms;
            n++;
        }
        if (reset) {
            for (auto& p : h.dev->timings) {
                h.dev->pool.push_back(p.first);
                h.dev->pool.push_back(p.second);
            }
            h.dev->timings.clear();
        }
    }
    if (avg_ms) *avg_ms = n ? total / n : 0.0;
    if (n_launches) *n_launches = n;
    return SRACK_OK;
}

// rows of the voice program's table (ctl = false) or of the control program's one-voice table
int device_read_rows(PatchHandle& h, int ctl_stage, int first_row, int n_rows, uint32_t* host_dst)
{
    const FlatProgram& P = ctl_stage >= 0 ? h.prog.ctl[(size_t)ctl_stage] : h.prog.voice;
    const size_t V = P.n_voices;
    const uint32_t* d_table = h.dev ? (ctl_stage >= 0 ? h.dev->ctl[(size_t)ctl_stage].d_table : h.dev->voice.d_table) : nullptr;
    if (d_table) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(host_dst, d_table + (size_t)first_row * V, sizeof(uint32_t) * V * (size_t)n_rows, hipMemcpyDeviceToHost));
    } else {  // nothing rendered yet: the initial table
        std::memcpy(host_dst, P.table.data() + (size_t)first_row * V, sizeof(uint32_t) * V * (size_t)n_rows);
    }
    return SRACK_OK;
}

const char* device_kernel_name(const PatchHandle& h) { return h.dev ? h.dev->kernel_name : ""; }

}  // namespace srack
