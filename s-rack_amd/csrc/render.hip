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

#include <cstdio>
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
    uint32_t V, T, n_waves, pad_;
    uint64_t n0;  // absolute index of this render's first sample
};

struct ChainRoles {  // op indices of the fused voice chain
    int osc_a, osc_l, vcf, adsr, vca, out;
};

namespace dev {

SRK_DEV double make_f64(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
SRK_DEV uint32_t f64_lo(double d) { return (uint32_t)__double2loint(d); }
SRK_DEV uint32_t f64_hi(double d) { return (uint32_t)__double2hiint(d); }

struct Ctx {            // what every tile function sees
    uint32_t* rows;     // LDS [n_rows][64]
    float* wires;       // LDS [n_slots][tile][64]
    int tile, n, lane;  // tile capacity, samples in this tile, lane
};

#define ROW(r) c.rows[(r) * 64 + c.lane]
#define WIRE(slot, i) c.wires[((slot) * c.tile + (i)) * 64 + c.lane]

SRK_DEV float par(const Ctx& c, const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(ROW(op.par_row[k])) : op.par_val[k]; }

// ---- one tile of one module type -----------------------------------------------------------------

__device__ __noinline__ void tile_osc(const Ctx& c, const DevOp& op)
{
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
    const int i_cv = op.in_slot[0], i_sync = op.in_slot[1];
    const int o_sine = op.out_slot[0], o_square = op.out_slot[1], o_saw = op.out_slot[2];
    for (int i = 0; i < c.n; i++) {
        float cv = (fl & OSC_HAS_CV) ? WIRE(i_cv, i) : 0.0f;
        float sync = (fl & OSC_HAS_SYNC) ? WIRE(i_sync, i) : 0.0f;
        float sine = 0.0f, square = 0.0f, saw = 0.0f;
        osc_step(fl, s, k, cv, sync, sine, square, saw);
        if (fl & OSC_OUT_SINE) WIRE(o_sine, i) = sine;
        if (fl & OSC_OUT_SQUARE) WIRE(o_square, i) = square;
        if (fl & OSC_OUT_SAW) WIRE(o_saw, i) = saw;
    }
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

__device__ __noinline__ void tile_vcf(const Ctx& c, const DevOp& op)
{
    const uint32_t fl = op.flags;
    VcfRegs s;
    vcf_load(c, op.state_row, s);
    const float freq = par(c, op, VCF_P_FREQ), exp_amt = par(c, op, VCF_P_EXP);
    const float res = vcf_resonance(par(c, op, VCF_P_RES));
    const int i_audio = op.in_slot[0], i_cv = op.in_slot[1];
    const int o_lp = op.out_slot[0], o_bp = op.out_slot[1], o_hp = op.out_slot[2];
    if (!(fl & VCF_HAS_CV)) vcf_coeffs(s, vcf_frequency(freq, 0.0f, exp_amt), res);  // constant cutoff: the check can only fire on the first sample
    for (int i = 0; i < c.n; i++) {
        float audio = (fl & VCF_HAS_AUDIO) ? WIRE(i_audio, i) : 0.0f;
        if (fl & VCF_HAS_CV) vcf_coeffs(s, vcf_frequency(freq, WIRE(i_cv, i), exp_amt), res);
        float lp, bp, hp;
        vcf_step(s, audio, lp, bp, hp);
        if (fl & VCF_OUT_LP) WIRE(o_lp, i) = lp;
        if (fl & VCF_OUT_BP) WIRE(o_bp, i) = bp;
        if (fl & VCF_OUT_HP) WIRE(o_hp, i) = hp;
    }
    vcf_store(c, op.state_row, s);
}

__device__ __noinline__ void tile_adsr(const Ctx& c, const DevOp& op)
{
    const int sr = op.state_row;
    AdsrRegs s;
    s.phase = __uint_as_float(ROW(sr + ADSR_S_PHASE));
    s.mode = (int)ROW(sr + ADSR_S_MODE);
    s.r_val = __uint_as_float(ROW(sr + ADSR_S_R_VAL));
    s.from_a_val = __uint_as_float(ROW(sr + ADSR_S_FROM_A));
    s.gate_last = ROW(sr + ADSR_S_GATE_LAST) != 0;
    const AdsrConst k = adsr_consts(par(c, op, ADSR_P_A), par(c, op, ADSR_P_D), par(c, op, ADSR_P_S), par(c, op, ADSR_P_R), par(c, op, ADSR_P_SR));
    const int i_gate = op.in_slot[0], o = op.out_slot[0];
    for (int i = 0; i < c.n; i++) {
        float gate = (op.flags & ADSR_HAS_GATE) ? WIRE(i_gate, i) : 0.0f;
        WIRE(o, i) = adsr_step(op.flags, s, k, gate);
    }
    ROW(sr + ADSR_S_PHASE) = __float_as_uint(s.phase);
    ROW(sr + ADSR_S_MODE) = (uint32_t)s.mode;
    ROW(sr + ADSR_S_R_VAL) = __float_as_uint(s.r_val);
    ROW(sr + ADSR_S_FROM_A) = __float_as_uint(s.from_a_val);
    ROW(sr + ADSR_S_GATE_LAST) = s.gate_last ? 1u : 0u;
}

__device__ __noinline__ void tile_vca(const Ctx& c, const DevOp& op)
{
    const bool negative = par(c, op, VCA_P_NEG) != 0.0f;
    const int i_audio = op.in_slot[0], i_cv = op.in_slot[1], o = op.out_slot[0];
    const bool both = (op.flags & (VCA_HAS_AUDIO | VCA_HAS_CV)) == (VCA_HAS_AUDIO | VCA_HAS_CV);
    for (int i = 0; i < c.n; i++) WIRE(o, i) = both ? vca_step(op.flags, negative, WIRE(i_audio, i), WIRE(i_cv, i)) : 0.0f;
}

__device__ __noinline__ void tile_mix(const Ctx& c, const DevOp& op)
{
    float gain[4];
    for (int k = 0; k < 4; k++) gain[k] = par(c, op, MIX_P_GAIN0 + k);
    const int o = op.out_slot[0];
    for (int i = 0; i < c.n; i++) {
        float in[4];
#pragma unroll
        for (int k = 0; k < 4; k++) in[k] = (op.flags & (1u << k)) ? WIRE(op.in_slot[k], i) : 0.0f;
        WIRE(o, i) = mixer_step(op.flags, in, gain);
    }
}

__device__ __noinline__ void tile_math(const Ctx& c, const DevOp& op)
{
    const float constant = par(c, op, MATH_P_CONST);
    const int o = op.out_slot[0];
    for (int i = 0; i < c.n; i++) {
        float a = (op.flags & MATH_HAS_IN1) ? WIRE(op.in_slot[0], i) : 0.0f;
        float b = (op.flags & MATH_HAS_IN2) ? WIRE(op.in_slot[1], i) : 0.0f;
        WIRE(o, i) = math_step(op.flags, a, b, constant);
    }
}

// Sum an LDS tile [rows_in_tile][64] over the 64 lanes: lane l owns row l % R and the column
// segment l / R (R = tile capacity, a power of two <= 64); columns are visited skewed by the row
// so the 32 lanes of a half-wave hit 32 different banks.  Result valid in lanes < R.
SRK_DEV float tile_row_sum(const float* t, int R, int lane)
{
    const int row = lane & (R - 1);
    const int seg = lane / R;          // 64 / R segments of R columns each
    const float* p = t + row * 64 + seg * R;
    float sum = 0.0f;
    for (int j = 0; j < R; j++) sum += p[(j + row) & (R - 1)];
    for (int m = R; m < 64; m <<= 1) sum += __shfl_xor(sum, m);
    return sum;
}

__device__ __noinline__ void tile_out(const Ctx& c, const DevOp& op, const KernelArgs& a, uint32_t t0, uint32_t voice, bool active)
{
    const int slot = op.in_slot[0], plane = op.aux;
    if (a.frames) {
        float* f = a.frames + ((size_t)plane * a.T + t0) * a.V + voice;
        if (active)
            for (int i = 0; i < c.n; i++) f[(size_t)i * a.V] = WIRE(slot, i);
    }
    if (a.mixpart) {
        if (!active)
            for (int i = 0; i < c.n; i++) WIRE(slot, i) = 0.0f;  // lanes past V contribute nothing
        __syncthreads();
        float sum = tile_row_sum(c.wires + (size_t)slot * c.tile * 64, c.tile, c.lane);
        if (c.lane < c.n) a.mixpart[((size_t)plane * a.n_waves + blockIdx.x) * a.T + t0 + c.lane] = sum;
        __syncthreads();
    }
}

__device__ __noinline__ void tile_delay_rd(const Ctx& c, const DevOp& op, const KernelArgs& a, uint64_t n_abs, uint32_t voice_c)
{
    const int o = op.out_slot[0];
    const uint32_t B = (uint32_t)a.prog.buffer_size;
    if (op.flags & DELAY_RING_GLOBAL) {
        const float* ring = a.rings + (size_t)op.aux * B * a.V + voice_c;
        uint32_t p = (uint32_t)(n_abs % B);
        for (int i = 0; i < c.n; i++) {
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

__device__ __noinline__ void tile_delay_wr(const Ctx& c, const DevOp& op, const KernelArgs& a, uint64_t n_abs, uint32_t voice, bool active)
{
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
__global__ __launch_bounds__(64) void render_interp(KernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const uint32_t voice = blockIdx.x * 64u + lane;
    const bool active = voice < a.V;
    const uint32_t voice_c = active ? voice : a.V - 1;  // idle lanes shadow the last voice; they never store
    const int n_rows = a.prog.n_rows, tile = a.prog.tile;
    dev::Ctx c;
    c.rows = lds;
    c.wires = reinterpret_cast<float*>(lds + (size_t)n_rows * 64);
    c.tile = tile;
    c.lane = lane;
    c.n = 0;
    for (int r = 0; r < n_rows; r++) c.rows[r * 64 + lane] = a.table[(size_t)r * a.V + voice_c];

    for (uint32_t t0 = 0; t0 < a.T; t0 += (uint32_t)tile) {
        c.n = (int)min((uint32_t)tile, a.T - t0);
        for (int i = 0; i < a.prog.n_ops; i++) {
            const DevOp& op = a.ops[i];
            switch (op.kind) {
            case OP_OSC: dev::tile_osc(c, op); break;
            case OP_VCF: dev::tile_vcf(c, op); break;
            case OP_ADSR: dev::tile_adsr(c, op); break;
            case OP_VCA: dev::tile_vca(c, op); break;
            case OP_MIX: dev::tile_mix(c, op); break;
            case OP_MATH: dev::tile_math(c, op); break;
            case OP_OUT: dev::tile_out(c, op, a, t0, voice, active); break;
            case OP_DELAY_RD: dev::tile_delay_rd(c, op, a, a.n0 + t0, voice_c); break;
            case OP_DELAY_WR: dev::tile_delay_wr(c, op, a, a.n0 + t0, voice, active); break;
            default: break;
            }
        }
    }
    if (active)
        for (int r = 0; r < a.prog.n_state_rows; r++) a.table[(size_t)r * a.V + voice] = c.rows[r * 64 + lane];
}

// ---- fused voice chain (patch P1's shape) -------------------------------------------------------------
// OSC_A.<port> -> VCF.<port> -> VCA <- ADSR <- OSC_L.<port>; all wires and all state in VGPRs.
constexpr int kMixRows = 32;

template <uint32_t kOscAPort, uint32_t kOscLPort, uint32_t kVcfPort, bool kExact>
__global__ __launch_bounds__(64) void render_voice_chain(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    __shared__ float mix_tile[kMixRows * 64];
    const int lane = threadIdx.x;
    const uint32_t voice = blockIdx.x * 64u + lane;
    const bool active = voice < a.V;
    const uint32_t vc = active ? voice : a.V - 1;
    const uint32_t V = a.V;
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

    // frames: uniform base pointer advanced by V per sample + a constant per-lane offset => SGPR base, 0 VALU
    float* frame_row = a.frames ? a.frames + (size_t)plane * a.T * V + (size_t)blockIdx.x * 64 : nullptr;
    float* mp = a.mixpart ? a.mixpart + ((size_t)plane * a.n_waves + blockIdx.x) * a.T : nullptr;
    const bool has_frames = frame_row != nullptr, has_mix = mp != nullptr;

    COsc ca, cl;
    AdsrSeg seg;
    if (!kExact) {
        cosc_init(ca, sa.pos, ka.delta);
        cosc_init(cl, sl.pos, kl.delta);
        adsr_seg_enter(sd, kd, seg);
    }

    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        for (int i = 0; i < n; i++) {
            float x, gate, env;
            if (kExact) {
                float sine = 0.0f, square = 0.0f, saw = 0.0f;
                osc_step(fa, sa, ka, 0.0f, 0.0f, sine, square, saw);
                x = kOscAPort == OSC_OUT_SINE ? sine : (kOscAPort == OSC_OUT_SQUARE ? square : saw);
                float gs = 0.0f, gq = 0.0f, gw = 0.0f;
                osc_step(fl, sl, kl, 0.0f, 0.0f, gs, gq, gw);
                gate = kOscLPort == OSC_OUT_SINE ? gs : (kOscLPort == OSC_OUT_SQUARE ? gq : gw);
            } else {
                x = cosc_step<kOscAPort>(ca);
                gate = cosc_step<kOscLPort>(cl);
            }
            float lp, bp, hp;
            vcf_step<!kExact>(sv, x, lp, bp, hp);
            const float y = kVcfPort == VCF_OUT_LP ? lp : (kVcfPort == VCF_OUT_BP ? bp : hp);
            if (kExact)
                env = adsr_step(ADSR_HAS_GATE, sd, kd, gate);
            else
                env = adsr_seg_step(sd, kd, seg, gate);
            const float o = vca_step(VCA_HAS_AUDIO | VCA_HAS_CV, negative, y, env);
            if (has_frames) {
                if (active) frame_row[lane] = o;
                frame_row += V;
            }
            if (has_mix) mix_tile[i * 64 + lane] = active ? o : 0.0f;
        }
        if (has_mix) {
            __syncthreads();
            float sum = tile_row_sum(mix_tile, kMixRows, lane);
            if (lane < n) mp[t0 + lane] = sum;
            __syncthreads();
        }
    }
    if (!kExact) {
        sa.pos = ca.pos;
        sl.pos = cl.pos;
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

// ---- mix-down, pass 2: mix[c][i] = sum over waves of mixpart[plane(c)][w][i] --------------------------
struct MixArgs {
    const float* mixpart;
    float* mix;
    uint32_t T, n_waves, n_channels;
    int32_t channel_plane[8];
};

__global__ __launch_bounds__(256) void mix_reduce(MixArgs m)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m.T) return;
    for (uint32_t c = 0; c < m.n_channels; c++) {
        const int plane = m.channel_plane[c];
        float s = 0.0f;
        if (plane >= 0) {
            const float* p = m.mixpart + (size_t)plane * m.n_waves * m.T + i;
            float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;  // fixed 4-way split: deterministic, 4 loads in flight
            uint32_t w = 0;
            for (; w + 4 <= m.n_waves; w += 4) {
                s0 += p[(size_t)(w + 0) * m.T];
                s1 += p[(size_t)(w + 1) * m.T];
                s2 += p[(size_t)(w + 2) * m.T];
                s3 += p[(size_t)(w + 3) * m.T];
            }
            for (; w < m.n_waves; w++) s0 += p[(size_t)w * m.T];
            s = (s0 + s1) + (s2 + s3);
        }
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

struct DeviceState {
    DevOp* d_ops = nullptr;
    uint32_t* d_table = nullptr;
    float* d_rings = nullptr;
    float* d_mixpart = nullptr;
    size_t mixpart_bytes = 0;
    size_t rings_bytes = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timings;  // (start, stop) pairs not yet read
    std::vector<hipEvent_t> pool;
    const char* kernel_name = "";
};

void device_release(DeviceState* d)
{
    if (!d) return;
    (void)hipFree(d->d_ops);
    (void)hipFree(d->d_table);
    (void)hipFree(d->d_rings);
    (void)hipFree(d->d_mixpart);
    for (auto& p : d->timings) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (auto e : d->pool) (void)hipEventDestroy(e);
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
    // device copies are refreshed lazily by device_render
    if (h.dev) {
        device_release(h.dev);
        h.dev = nullptr;
    }
    return SRACK_OK;
}

static int upload_program(PatchHandle& h)
{
    auto* d = new DeviceState();
    h.dev = d;
    const FlatProgram& P = h.prog;
    if (!P.ops.empty()) {
        HIP_TRY(hipMalloc(&d->d_ops, sizeof(DevOp) * P.ops.size()));
        HIP_TRY(hipMemcpy(d->d_ops, P.ops.data(), sizeof(DevOp) * P.ops.size(), hipMemcpyHostToDevice));
    }
    if (!P.table.empty()) {
        HIP_TRY(hipMalloc(&d->d_table, sizeof(uint32_t) * P.table.size()));
        HIP_TRY(hipMemcpy(d->d_table, P.table.data(), sizeof(uint32_t) * P.table.size(), hipMemcpyHostToDevice));
    }
    if (P.hdr.n_rings > 0) {
        d->rings_bytes = sizeof(float) * (size_t)P.hdr.n_rings * (size_t)P.hdr.buffer_size * P.n_voices;
        HIP_TRY(hipMalloc(&d->d_rings, d->rings_bytes));
        HIP_TRY(hipMemset(d->d_rings, 0, d->rings_bytes));  // AudioBuffer::new fills 0.0 (synth.rs:31-33)
    }
    return SRACK_OK;
}

template <uint32_t A, uint32_t F, bool E>
static void launch_chain(uint32_t lfo_port, const KernelArgs& ka, const ChainRoles& roles, dim3 grid, hipStream_t st)
{
    if (lfo_port == OSC_OUT_SQUARE)
        hipLaunchKernelGGL((render_voice_chain<A, OSC_OUT_SQUARE, F, E>), grid, dim3(64), 0, st, ka, roles);
    else if (lfo_port == OSC_OUT_SAW)
        hipLaunchKernelGGL((render_voice_chain<A, OSC_OUT_SAW, F, E>), grid, dim3(64), 0, st, ka, roles);
    else
        hipLaunchKernelGGL((render_voice_chain<A, OSC_OUT_SINE, F, E>), grid, dim3(64), 0, st, ka, roles);
}

template <uint32_t A, bool E>
static void launch_chain_f(uint32_t vcf_port, uint32_t lfo_port, const KernelArgs& ka, const ChainRoles& roles, dim3 grid, hipStream_t st)
{
    if (vcf_port == VCF_OUT_LP)
        launch_chain<A, VCF_OUT_LP, E>(lfo_port, ka, roles, grid, st);
    else if (vcf_port == VCF_OUT_BP)
        launch_chain<A, VCF_OUT_BP, E>(lfo_port, ka, roles, grid, st);
    else
        launch_chain<A, VCF_OUT_HP, E>(lfo_port, ka, roles, grid, st);
}

template <bool E>
static void launch_chain_a(uint32_t osc_port, uint32_t vcf_port, uint32_t lfo_port, const KernelArgs& ka, const ChainRoles& roles, dim3 grid, hipStream_t st)
{
    if (osc_port == OSC_OUT_SAW)
        launch_chain_f<OSC_OUT_SAW, E>(vcf_port, lfo_port, ka, roles, grid, st);
    else if (osc_port == OSC_OUT_SQUARE)
        launch_chain_f<OSC_OUT_SQUARE, E>(vcf_port, lfo_port, ka, roles, grid, st);
    else
        launch_chain_f<OSC_OUT_SINE, E>(vcf_port, lfo_port, ka, roles, grid, st);
}

int device_render(PatchHandle& h, uint32_t n_samples, float* d_frames, float* d_mix, uint32_t flags, void* stream)
{
    int rc = ensure_program(h, flags);
    if (rc != SRACK_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const FlatProgram& P = h.prog;
    const uint32_t V = P.n_voices, T = n_samples, C = (uint32_t)P.hdr.n_channels;
    if (T == 0) return SRACK_OK;
    if (!h.dev) {
        rc = upload_program(h);
        if (rc != SRACK_OK) return rc;
    }
    DeviceState* d = h.dev;
    const uint32_t n_waves = (V + 63) / 64;

    if (P.hdr.n_planes == 0) {  // nothing reaches the output: silence (output.rs:55)
        if (d_mix) {
            size_t n = (size_t)C * T;
            hipLaunchKernelGGL(fill_zero, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_mix, n);
        }
        h.samples_rendered += T;
        return SRACK_OK;
    }
    if (d_mix) {
        size_t need = sizeof(float) * (size_t)P.hdr.n_planes * n_waves * T;
        if (need > d->mixpart_bytes) {
            (void)hipFree(d->d_mixpart);
            d->d_mixpart = nullptr;
            d->mixpart_bytes = 0;
            HIP_TRY(hipMalloc(&d->d_mixpart, need));
            d->mixpart_bytes = need;
        }
    }
    KernelArgs ka{};
    ka.ops = d->d_ops;
    ka.prog = P.hdr;
    ka.table = d->d_table;
    ka.rings = d->d_rings;
    ka.frames = d_frames;
    ka.mixpart = d_mix ? d->d_mixpart : nullptr;
    ka.V = V;
    ka.T = T;
    ka.n_waves = n_waves;
    ka.n0 = h.samples_rendered;

    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto get_event = [&](hipEvent_t& e) -> int {
        if (!d->pool.empty()) {
            e = d->pool.back();
            d->pool.pop_back();
            return SRACK_OK;
        }
        HIP_TRY(hipEventCreate(&e));
        return SRACK_OK;
    };
    if ((rc = get_event(e0)) != SRACK_OK || (rc = get_event(e1)) != SRACK_OK) return rc;
    HIP_TRY(hipEventRecord(e0, st));

    if (P.fused == FUSED_VOICE_CHAIN) {
        ChainRoles roles{};
        uint32_t osc_port = 0, lfo_port = 0, vcf_port = 0;
        for (int i = 0; i < (int)P.ops.size(); i++) {
            const DevOp& op = P.ops[(size_t)i];
            if (op.kind == OP_VCF) { roles.vcf = i; vcf_port = op.flags & (VCF_OUT_LP | VCF_OUT_BP | VCF_OUT_HP); }
            if (op.kind == OP_ADSR) roles.adsr = i;
            if (op.kind == OP_VCA) roles.vca = i;
            if (op.kind == OP_OUT) roles.out = i;
        }
        const Graph& g = h.graph;
        roles.osc_a = P.op_of_module[(size_t)g.modules[(size_t)P.ops[(size_t)roles.vcf].module].in[0].src];
        roles.osc_l = P.op_of_module[(size_t)g.modules[(size_t)P.ops[(size_t)roles.adsr].module].in[0].src];
        osc_port = P.ops[(size_t)roles.osc_a].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
        lfo_port = P.ops[(size_t)roles.osc_l].flags & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
        if (flags & SRACK_RENDER_EXACT_OSC)
            launch_chain_a<true>(osc_port, vcf_port, lfo_port, ka, roles, dim3(n_waves), st);
        else
            launch_chain_a<false>(osc_port, vcf_port, lfo_port, ka, roles, dim3(n_waves), st);
        d->kernel_name = "render_voice_chain";
    } else {
        size_t lds = ((size_t)P.hdr.n_rows + (size_t)P.hdr.n_slots * P.hdr.tile) * 256;
        hipLaunchKernelGGL(render_interp, dim3(n_waves), dim3(64), lds, st, ka);
        d->kernel_name = "render_interp";
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(e1, st));
    d->timings.emplace_back(e0, e1);
    if (d->timings.size() > 4096) {  // nobody is reading them: recycle the oldest
        d->pool.push_back(d->timings.front().first);
        d->pool.push_back(d->timings.front().second);
        d->timings.erase(d->timings.begin());
    }

    if (d_mix) {
        MixArgs m{};
        m.mixpart = d->d_mixpart;
        m.mix = d_mix;
        m.T = T;
        m.n_waves = n_waves;
        m.n_channels = C;
        for (int c = 0; c < 8; c++) m.channel_plane[c] = P.hdr.channel_plane[c];
        hipLaunchKernelGGL(mix_reduce, dim3((T + 255) / 256), dim3(256), 0, st, m);
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

int device_read_rows(PatchHandle& h, int first_row, int n_rows, uint32_t* host_dst)
{
    const size_t V = h.prog.n_voices;
    if (h.dev && h.dev->d_table) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(host_dst, h.dev->d_table + (size_t)first_row * V, sizeof(uint32_t) * V * (size_t)n_rows, hipMemcpyDeviceToHost));
    } else {  // nothing rendered yet: the initial table
        std::memcpy(host_dst, h.prog.table.data() + (size_t)first_row * V, sizeof(uint32_t) * V * (size_t)n_rows);
    }
    return SRACK_OK;
}

const char* device_kernel_name(const PatchHandle& h) { return h.dev ? h.dev->kernel_name : ""; }

}  // namespace srack
