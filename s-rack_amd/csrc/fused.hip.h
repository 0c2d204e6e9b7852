// fused.hip.h — the fused kernels: whole patch shapes with every wire and all state in VGPRs (no LDS except the
// mix-down transpose tile), and the passes that sum the per-wave mix partials.
//   render_voice_chain        patch P1 with every module per voice
//   render_voice_chain_track  P1 after uniform hoisting (the flagship kernel), with the co-scheduled control block
//   render_ctl_gate_env       P1's control program: OSC -> ADSR -> track
//   render_voice_chain_seq    patch P3's shape: sequencer-driven subtractive voice
//   render_fm_pair            patch P2 at buffer_size 1: two-operator FM with a z^-1 feedback edge
//   mix_reduce_groups/final   deterministic sum of the per-wave partials (no atomics)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "interp.hip.h"

#ifndef SRK_FM_UNROLL
#define SRK_FM_UNROLL 8
#endif

namespace srack {

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
    uint32_t port;     // OSC_OUT_* of the gate oscillator, | OSC_EXACT in the exact render mode
    uint32_t* table_out;  // where the state goes (a tick session, render.hip: another copy of the table; null: in place) ...
    uint32_t n_rows;      // ... and how many rows the table has
};

// kExact: the oscillator's outputs are the reference's f64 formulas (osc_step with OSC_EXACT) instead of the default mode's f32
// PolyBLEP.  The phase recurrence is the same f64 add and exact wrap either way, and the segmented envelope performs adsr.rs's own
// operations in both modes.
template <uint32_t kOscPort, bool kExact>
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

    // One voice on one wave is a latency chain, so the block works a tile at a time across lanes (modules.hip.h): the oscillator runs only
    // its two-instruction phase recurrence per sample and lane j evaluates sample j's output; the envelope takes whole runs of samples
    // between the events that can end its segment.  55 ns per sample where the sample-by-sample form (four-sample speculative groups)
    // took 102 — it is the first chunk's track that nothing hides.
    constexpr uint32_t osc_flags = OSC_AA | kOscPort | (kExact ? OSC_EXACT : 0u);
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        const float gate = cosc_tile<kExact>(osc_flags, cl, n);
        const float env = adsr_seg_tile(sd, kd, seg, gate, n);
        if (lane < n) track[t0 + lane] = env;
    }
    adsr_seg_flush(sd, seg);
    uint32_t* const tout = a.table_out ? a.table_out : a.table;
    if (a.table_out) {  // the rows this block does not rewrite travel with the state (lane r's store, then lane 0's below: fenced)
        for (uint32_t rr = (uint32_t)lane; rr < a.n_rows; rr += 64u) a.table_out[rr] = a.table[rr];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    if (lane == 0) {
        tout[ol.state_row + OSC_S_POS_LO] = f64_lo(cl.pos);
        tout[ol.state_row + OSC_S_POS_HI] = f64_hi(cl.pos);
        tout[ol.state_row + OSC_S_SYNC_LAST] = 0u;
        tout[od.state_row + ADSR_S_PHASE] = __float_as_uint(sd.phase);
        tout[od.state_row + ADSR_S_MODE] = (uint32_t)sd.mode;
        tout[od.state_row + ADSR_S_R_VAL] = __float_as_uint(sd.r_val);
        tout[od.state_row + ADSR_S_FROM_A] = __float_as_uint(sd.from_a_val);
        tout[od.state_row + ADSR_S_GATE_LAST] = sd.gate_last ? 1u : 0u;
    }
}


template <bool kExact>
SRK_DEV void ctl_gate_env(const CtlWork& w)
{
    const uint32_t port = w.port & (OSC_OUT_SINE | OSC_OUT_SQUARE | OSC_OUT_SAW);
    if (port == OSC_OUT_SQUARE)
        ctl_gate_env_body<OSC_OUT_SQUARE, kExact>(w);
    else if (port == OSC_OUT_SAW)
        ctl_gate_env_body<OSC_OUT_SAW, kExact>(w);
    else
        ctl_gate_env_body<OSC_OUT_SINE, kExact>(w);
}

__global__ __launch_bounds__(64) void render_ctl_gate_env(CtlWork w)
{
    if (w.port & OSC_EXACT)
        ctl_gate_env<true>(w);
    else
        ctl_gate_env<false>(w);
}

// ---- fused voice chain (patch P1's shape) -------------------------------------------------------------
// OSC_A.<port> -> VCF.<port> -> VCA <- ADSR <- OSC_L.<port>; all wires and all state in VGPRs.

template <uint32_t kOscAPort, uint32_t kOscLPort, uint32_t kVcfPort, bool kExact, int kOut>
__global__ __launch_bounds__(64) void render_voice_chain(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
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
    ka.inv_dt = inv_dt_f32(ka.delta);
    kl.sr = ol.sample_rate;
    kl.val = 0.0;
    kl.delta = ol.delta_row >= 0 ? make_f64(row(ol.delta_row), row(ol.delta_row + 1)) : ol.delta;
    kl.inv_dt = inv_dt_f32(kl.delta);

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
    if (a.T > 0) vcf_coeffs<!kExact>(sv, vcf_frequency(parv(ov, VCF_P_FREQ), 0.0f, parv(ov, VCF_P_EXP)), vcf_resonance(parv(ov, VCF_P_RES)));
    bool sv_fin = !kExact || vcf_nan_free(sv);  // (exact mode: v_med3 clamps while nothing can turn into a NaN, modules.hip.h vcf_run)

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
    FOsc fa_osc;
    // default mode, a saw nothing integrates or thresholds: the host stored its phase and increment as value * 2^64 (OSC_FIXED_PHASE; wave-uniform)
    const bool fixed_a = !kExact && kOscAPort == OSC_OUT_SAW && (oa.flags & OSC_FIXED_PHASE) != 0;
    uint32_t fix_lo = 0u, fix_hi = 0u;  // ... its phase after exactly t samples
    AdsrSeg seg;
    float x = 0.0f, gate = 0.0f;
    double pos_a = sa.pos, pos_l = sl.pos;  // oscillator phases after exactly t samples (the loop runs one sample ahead)
    if (!kExact) {
        if (fixed_a) {
            const uint64_t dbits = (uint64_t)__double_as_longlong(oa.delta);
            fosc_init(fa_osc, row(oa.state_row + OSC_S_POS_LO), row(oa.state_row + OSC_S_POS_HI), oa.delta_row >= 0 ? row(oa.delta_row) : (uint32_t)dbits,
                      oa.delta_row >= 0 ? row(oa.delta_row + 1) : (uint32_t)(dbits >> 32));
            fix_lo = fa_osc.lo;
            fix_hi = fa_osc.hi;
        } else {
            cosc_init(ca, sa.pos, ka.delta);
        }
        cosc_init(cl, sl.pos, kl.delta);
        adsr_seg_enter(sd, kd, seg);
        if (a.T > 0) {  // software pipeline: the oscillators of sample t+1 are evaluated beside the filter of sample t
            x = fixed_a ? fosc_saw(fa_osc) : cosc_step<kOscAPort>(ca);
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
            vcf_run<!kExact>(sv, sv_fin, x, lp, bp, hp);
            const float y = kVcfPort == VCF_OUT_LP ? lp : (kVcfPort == VCF_OUT_BP ? bp : hp);
            if (!kExact) {  // next sample's oscillators: same basic block as the filter chain above => they interleave
                pos_l = cl.pos;
                if (fixed_a) {
                    fix_lo = fa_osc.lo;
                    fix_hi = fa_osc.hi;
                    x_next = fosc_saw(fa_osc);
                } else {
                    pos_a = ca.pos;
                    x_next = cosc_step<kOscAPort>(ca);
                }
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
        emit_flush<kOut>(em, mix_tile, t0, n, V);
    }
    if (!kExact) {
        sa.pos = pos_a;
        sl.pos = pos_l;
        sa.sync_last = sl.sync_last = false;  // sync unconnected: `last` follows the constant 0.0 input
        adsr_seg_flush(sd, seg);
    }

    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(oa.state_row + OSC_S_POS_LO, fixed_a ? fix_lo : f64_lo(sa.pos));
        put(oa.state_row + OSC_S_POS_HI, fixed_a ? fix_hi : f64_hi(sa.pos));
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
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
    // Block 0 of a co-scheduled launch is not a voice wave: it computes the NEXT chunk's envelope track while the
    // voice blocks consume this chunk's (written by the previous launch).  Same stream, no events, no second queue.
    if (blockIdx.x < a.block0) {
        // a latency chain sharing its SIMD with four throughput-bound voice waves: without priority it gets a
        // fifth of the issue slots and can outlast the voice blocks (measured: 1.9 -> 2.6 ms per launch)
        __builtin_amdgcn_s_setprio(3);
        ctl_gate_env<kExact>(co);
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
    ka.inv_dt = inv_dt_f32(ka.delta);

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
    if (a.T > 0) vcf_coeffs<!kExact>(sv, vcf_frequency(parv(ov, VCF_P_FREQ), 0.0f, parv(ov, VCF_P_EXP)), vcf_resonance(parv(ov, VCF_P_RES)));
    bool sv_fin = !kExact || vcf_nan_free(sv);  // (exact mode: v_med3 clamps while nothing can turn into a NaN, modules.hip.h vcf_run)
    const bool negative = parv(oc, VCA_P_NEG) != 0.0f;

    Emit em = make_emit(a, plane, lane);

    // default mode, saw: the phase lives in 64-bit fixed point (modules.hip.h, FOsc); the host stores state and increment so
    constexpr bool kFixed = !kExact && kOscAPort == OSC_OUT_SAW;
    COsc ca;
    FOsc fa_osc;
    float x = 0.0f;
    double pos_a = sa.pos;
    uint32_t fpos_lo = 0u, fpos_hi = 0u;
    if (kFixed) {
        const uint64_t dbits = oa.delta_row >= 0 ? ((uint64_t)row(oa.delta_row + 1) << 32) | row(oa.delta_row) : (uint64_t)__double_as_longlong(oa.delta);
        fosc_init(fa_osc, row(oa.state_row + OSC_S_POS_LO), row(oa.state_row + OSC_S_POS_HI), (uint32_t)dbits, (uint32_t)(dbits >> 32));
        fpos_lo = fa_osc.lo;
        fpos_hi = fa_osc.hi;
        if (a.T > 0) x = fosc_saw(fa_osc);
    } else if (!kExact) {
        cosc_init(ca, sa.pos, ka.delta);
        if (a.T > 0) x = cosc_step<kOscAPort>(ca);
    }
    // The envelope track is wave-uniform data written by an earlier launch: it is read through the scalar unit (constant
    // address space => s_load_dwordx8/x16 straight into SGPRs, a tile at a time), costing no vector instruction at all.
    typedef const __attribute__((address_space(4))) float CFloat;
    CFloat* env_s = (CFloat*)(uintptr_t)env_track;
    // Exact mode, saw: the oscillator writes a whole tile first (modules.hip.h, XSaw: windowless values, then each lane repairs
    // its own PolyBLEP rows with the reference's f64 division), into the rows of the mix tile that the samples' outputs overwrite
    // one by one afterwards.  Its preconditions also make every filter input finite, which licenses the v_med3 clamps.
    XSaw xo;
    const bool xs = kExact && kOscAPort == OSC_OUT_SAW && xsaw_usable(sa.pos, ka.delta) && vcf_nan_free(sv);
    if (xs) xsaw_init(xo, sa.pos, ka.delta);
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        if (xs) {
            xsaw_tile(xo, mix_tile + lane, kMixPitch, n);
            auto sample_x = [&](int i, float xin) {
                const float env = env_s[t0 + (uint32_t)i];
                float lp, bp, hp;
                vcf_step<false, true>(sv, xin, lp, bp, hp);
                const float y = kVcfPort == VCF_OUT_LP ? lp : (kVcfPort == VCF_OUT_BP ? bp : hp);
                const bool cv_pos = (uint32_t)(__float_as_int(env) - 1) < 0x7f800000u;
                const float o = (negative || cv_pos) ? y * env : 0.0f;
                emit_put<kOut>(em, mix_tile, o, i, V);
            };
            if (n == kMixRows) {  // eight rows of the tile in flight per LDS round trip; a row is read before its output overwrites it
#pragma unroll
                for (int i0 = 0; i0 < kMixRows; i0 += 8) {
                    float xin[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) xin[u] = mix_tile[(i0 + u) * kMixPitch + lane];
#pragma unroll
                    for (int u = 0; u < 8; u++) sample_x(i0 + u, xin[u]);
                }
            } else {
                for (int i = 0; i < n; i++) sample_x(i, mix_tile[i * kMixPitch + lane]);
            }
            emit_flush<kOut>(em, mix_tile, t0, n, V);
            continue;
        }
        auto sample = [&](int i) {
            const float env = env_s[t0 + (uint32_t)i];
            if (kExact) {
                float sine = 0.0f, square = 0.0f, saw = 0.0f;
                osc_step(fa, sa, ka, 0.0f, 0.0f, sine, square, saw);
                x = kOscAPort == OSC_OUT_SINE ? sine : (kOscAPort == OSC_OUT_SQUARE ? square : saw);
            }
            float lp, bp, hp;
            vcf_run<!kExact>(sv, sv_fin, x, lp, bp, hp);
            const float y = kVcfPort == VCF_OUT_LP ? lp : (kVcfPort == VCF_OUT_BP ? bp : hp);
            if (kFixed) {
                fpos_lo = fa_osc.lo;
                fpos_hi = fa_osc.hi;
                x = fosc_saw(fa_osc);  // sample t+1
            } else if (!kExact) {
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
#pragma unroll 32
            for (int i = 0; i < kMixRows; i++) sample(i);
        } else {
            for (int i = 0; i < n; i++) sample(i);
        }
        emit_flush<kOut>(em, mix_tile, t0, n, V);
    }
    if (!kExact) {
        sa.pos = pos_a;
        sa.sync_last = false;
    }
    if (xs) {
        sa.pos = xo.pos;
        sa.sync_last = false;  // sync unconnected: `last` follows the constant 0.0 input
    }
    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(oa.state_row + OSC_S_POS_LO, kFixed ? fpos_lo : f64_lo(sa.pos));
        put(oa.state_row + OSC_S_POS_HI, kFixed ? fpos_hi : f64_hi(sa.pos));
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
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
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
    if (!has_cut && a.T > 0) vcf_coeffs<true>(sv, vcf_frequency(vfreq, 0.0f, vexp), vres);
    const bool negative = parv(oc, VCA_P_NEG) != 0.0f;

    Emit em = make_emit(a, plane, lane);
    float* extra_row[4] = {nullptr, nullptr, nullptr, nullptr};   // frame rows of the track-fed planes (wave-uniform)
    float* extra_mp[4] = {nullptr, nullptr, nullptr, nullptr};    // ... and their mix partials
    __amdgpu_buffer_rsrc_t extra_rsrc[4];
    const float* extra_track[4] = {env_track, env_track, env_track, env_track};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        if (e >= r.n_extra) continue;
        extra_track[e] = a.tracks + (size_t)r.extra_trk[e] * a.t_stride;
        if (a.frames) extra_row[e] = a.frames + (size_t)r.extra_plane[e] * a.plane_stride + wm.wave0;
        extra_rsrc[e] = __builtin_amdgcn_make_buffer_rsrc(extra_row[e], 0, 0x7fffffff, 0x00020000);
        if (a.mixpart) extra_mp[e] = a.mixpart + ((size_t)r.extra_plane[e] * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride;
    }

    // Per-sample track values come through the scalar unit (constant address space => s_load into SGPRs); the VGPR tiles
    // below (lane l = sample l of the tile) only serve the once-per-tile "did anything change" ballots and the mix partials.
    typedef const __attribute__((address_space(4))) float CFloat;
    CFloat* pitch_s = (CFloat*)(uintptr_t)pitch_track;
    CFloat* cut_s = (CFloat*)(uintptr_t)cut_track;
    CFloat* env_s = (CFloat*)(uintptr_t)env_track;
    CFloat* extra_s[4];
#pragma unroll
    for (int e = 0; e < 4; e++) extra_s[e] = (CFloat*)(uintptr_t)extra_track[e];
    const uint32_t l32 = (uint32_t)(lane & (kMixRows - 1));
    auto fetch = [&](const float* trk, uint32_t t0) { return trk[min(t0 + l32, a.T - 1)]; };
    float pitch_tile = fetch(pitch_track, 0), cut_tile = fetch(cut_track, 0);
    float extra_tile[4];
#pragma unroll
    for (int e = 0; e < 4; e++) extra_tile[e] = fetch(extra_track[e], 0);
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const float pitch_next = fetch(pitch_track, t0 + kMixRows), cut_next = fetch(cut_track, t0 + kMixRows);
        float extra_next[4];
#pragma unroll
        for (int e = 0; e < 4; e++) extra_next[e] = fetch(extra_track[e], t0 + kMixRows);
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        // A note lasts thousands of samples and an envelope rests in sustain or at zero for long stretches: when a whole
        // tile holds the pitch (the cutoff CV) the oscillator (the coefficients) were last set up for, its samples run
        // without the per-sample "did it change" branches, which lets the compiler interleave oscillator and filter of
        // neighbouring samples.  `steady_pitch` / `steady_cut` are compile-time constants inside each unrolled loop.
        auto sample = [&](int i, bool steady_pitch, bool steady_cut) {
            if (!steady_pitch) {
                const uint32_t pb = __float_as_uint(pitch_s[t0 + (uint32_t)i]);
                if (!have_pitch || pb != seen_pitch) {  // a new note (scalar test): new increment, carried terms rebuilt
                    have_pitch = true;
                    seen_pitch = pb;
                    const float note = __uint_as_float(pb);
                    cv_lane = has_math ? math_step(mflags, note, 0.0f, mconst) : note;
                    const double delta = osc_delta_cold((double)cv_lane + ko.val, ko.sr);  // once per note: the reference's own increment (modules.hip.h)
                    carried = __builtin_amdgcn_ballot_w64(!(delta < 0.25)) == 0;
                    cosc_init(co, co.pos, delta);
                }
            }
            float x;
            if (steady_pitch || carried) {  // (a steady tile is only declared when the carried form holds)
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
            if (has_cut && !steady_cut) {  // vcf_coeffs itself recomputes only for lanes whose (frequency, res) changed
                const float cutv = cut_s[t0 + (uint32_t)i];
                vcf_coeffs<true>(sv, vcf_frequency(vfreq, cutv, vexp), vres);
            }
            float lp, bp, hp;
            vcf_step<true>(sv, x, lp, bp, hp);
            const float y = vport == VCF_OUT_LP ? lp : (vport == VCF_OUT_BP ? bp : hp);
            const float env = env_s[t0 + (uint32_t)i];
            const bool cv_pos = (uint32_t)(__float_as_int(env) - 1) < 0x7f800000u;  // env > 0.0 on the scalar unit
            const float o = (negative || cv_pos) ? y * env : 0.0f;
            emit_put<kOut>(em, mix_tile, o, i, V);
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (e < r.n_extra && extra_row[e]) {  // same tile-relative row offset as the main plane (em.soff was just advanced)
                    const float v = extra_s[e][t0 + (uint32_t)i];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), extra_rsrc[e], em.lane_c * 4, (int)(em.soff - V * 4u), 2);
                }
        };
        const bool in_tile = (int)l32 < n;
        const bool steady_pitch = have_pitch && carried && __builtin_amdgcn_ballot_w64(in_tile && __float_as_uint(pitch_tile) != seen_pitch) == 0;
        const bool steady_cut = !has_cut || (have_cut && __builtin_amdgcn_ballot_w64(in_tile && __float_as_uint(cut_tile) != seen_cut) == 0);
        if (n == kMixRows) {
            if (steady_pitch && steady_cut) {
#pragma unroll 8
                for (int i = 0; i < kMixRows; i++) sample(i, true, true);
            } else if (steady_pitch) {
#pragma unroll 8
                for (int i = 0; i < kMixRows; i++) sample(i, true, false);
            } else {
#pragma unroll 4
                for (int i = 0; i < kMixRows; i++) sample(i, false, false);
            }
        } else {
            for (int i = 0; i < n; i++) sample(i, false, false);
        }
        if (has_cut && !steady_cut) {  // the coefficients now belong to the tile's last cutoff CV
            have_cut = true;
            seen_cut = __float_as_uint(cut_s[t0 + (uint32_t)(n - 1)]);
        }
        emit_flush<kOut>(em, mix_tile, t0, n, V);
#pragma unroll
        for (int e = 0; e < 4; e++)
            if (e < r.n_extra && extra_row[e]) {
                extra_row[e] += (size_t)n * V;
                extra_rsrc[e] = __builtin_amdgcn_make_buffer_rsrc(extra_row[e], 0, 0x7fffffff, 0x00020000);
            }
#pragma unroll
        for (int e = 0; e < 4; e++)  // every voice carries the same sample: the wave's partial is count x sample
            if (e < r.n_extra && extra_mp[e] && lane < n) extra_mp[e][t0 + lane] = (float)em.n_active * extra_tile[e];
        pitch_tile = pitch_next;
        cut_tile = cut_next;
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

// Wave-uniform facts about an FM pair's voices (default mode), from which the kernels below pick their sample loop.  Inactive lanes
// mirror a real voice (WaveMap::vc), so every lane votes.  In the proved loops the increment is scale * 2^cv with scale = 440 / sr * 2^val
// per voice (OSC_VAL_FOLDED), so what matters is |cv| <= |gain| (the fed-back value is a sine: at most 1, checked where it enters).
struct FmFacts {
    bool tame = false;  // |gain| + |val| below 1000 per oscillator, both phases in [0, 1), sample rates >= 1: increments finite and >= 0
    int mod = 0, car = 0;  // per oscillator: 2 = |gain| <= 1/2 (no range reduction), 1 = |gain| <= 2 ((2^(cv/4))^4), 0 = range reduction
};
using dev::fm_gain_class;  // modules.hip.h: shared with the kernels specialised at run time
SRK_DEV FmFacts fm_facts(float c_fb, const dev::OscConst& km, double pos_m, float c_ix, const dev::OscConst& kc, double pos_c)
{
    // the exponent of an increment is val + cv with |cv| <= |gain|: the SUM has to stay clear of 2^x's overflow (val = 600 with gain = 600 would
    // not), and then scale = 440 / sr * 2^val and scale * 2^cv are finite too
    const bool sizes = (double)__builtin_fabsf(c_fb) + __builtin_fabs(km.val) < 1000.0 && (double)__builtin_fabsf(c_ix) + __builtin_fabs(kc.val) < 1000.0;
    const bool phases = pos_m >= 0.0 && pos_m < 1.0 && pos_c >= 0.0 && pos_c < 1.0;
    const bool rates = km.sr >= 1.0 && kc.sr >= 1.0 && sizes && dev::osc_below_rate(c_fb, km.val, km.sr) && dev::osc_below_rate(c_ix, kc.val, kc.sr);  // 440 / sr finite, increments below a cycle per sample (modules.hip.h)
    FmFacts f;
    f.tame = __builtin_amdgcn_ballot_w64(!(sizes && phases && rates)) == 0;  // NaNs vote no
    f.mod = fm_gain_class(c_fb);
    f.car = fm_gain_class(c_ix);
    return f;
}
// the proved sample loops: tile(modulator flags, carrier flags) for the facts' classes
template <uint32_t kMod, uint32_t kCar, class Tile>
SRK_DEV void fm_proved_tile(const FmFacts& f, Tile&& tile)
{
    using std::integral_constant;
    constexpr uint32_t P = OSC_PHASE_TAME | OSC_VAL_FOLDED;
    auto with_car = [&](auto m) {
        constexpr uint32_t M = kMod | P | decltype(m)::value;
        if (f.car == 2)
            tile(integral_constant<uint32_t, M>{}, integral_constant<uint32_t, kCar | P | OSC_CV_SMALL>{});
        else if (f.car == 1)
            tile(integral_constant<uint32_t, M>{}, integral_constant<uint32_t, kCar | P | OSC_CV_QUAD>{});
        else
            tile(integral_constant<uint32_t, M>{}, integral_constant<uint32_t, kCar | P>{});
    };
    if (f.mod == 2)  // (a feedback gain between 1/2 and 2 takes the range reduction: three more copies of every loop for one more instruction pair)
        with_car(integral_constant<uint32_t, OSC_CV_SMALL>{});
    else
        with_car(integral_constant<uint32_t, 0u>{});
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
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
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
    constexpr uint32_t fo_carrier = fo | OSC_SINE_LOOSE;  // the carrier's sine only feeds the OutputModule (the matched shape): nothing integrates it
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
    if (!kExact) {  // (the proved loops: OSC_VAL_FOLDED)
        km.scale = (440.0 / km.sr) * exp2(km.val);
        kc.scale = (440.0 / kc.sr) * exp2(kc.val);
    }
    // both MATH modules are Multiply by a constant (host-checked): in1 * constant (math.rs:152)
    const float c_fb = parv(ofb, MATH_P_CONST), c_ix = parv(oix, MATH_P_CONST);
    float fed = __uint_as_float(row(ring_row));  // OSC_M.sine of the previous tick (0.0 before the first)

    Emit em = make_emit(a, plane, lane);
    float sq = 0.0f, sw = 0.0f;
    float sine_m = 0.0f;
    double pos_m = sm.pos;  // modulator phase after exactly t samples (the loop runs it one sample ahead)
    // What a wave can prove about its own voices once per launch (default mode).  A sine is at most 1 in magnitude, so each oscillator's
    // CV is bounded by its gain: below 1000 (and |val| too) the increment is finite and positive and the wrap is one v_fract
    // (OSC_PHASE_TAME); at most 2 and 2^cv is (2^(cv/4))^4, at most 1/2 and it needs no range reduction at all (fm_facts).  The folded
    // val and the squarings round differently from the literal form (1e-16), so a launch is proved as a whole or not at all — the first
    // modulator step and a ragged last tile included: a render must not depend on where it was split.  The one value that is not a
    // sine is the ring's initial one (the host's): checked here.
    const FmFacts facts = kExact ? FmFacts{} : fm_facts(c_fb, km, sm.pos, c_ix, kc, sc.pos);
    const bool proved = !kExact && facts.tame && __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(fed) <= 1.0f)) == 0;
    if (a.T > 0) {  // modulator of sample 0
        if (proved)
            fm_proved_tile<fo, fo_carrier>(facts, [&](auto m, auto) { osc_step(decltype(m)::value, sm, km, fed * c_fb, 0.0f, sine_m, sq, sw); });
        else
            osc_step(fo, sm, km, fed * c_fb, 0.0f, sine_m, sq, sw);
    }
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        auto tile = [&](auto fm_c, auto fc_c, auto whole) {
            constexpr uint32_t FM = decltype(fm_c)::value, FC = decltype(fc_c)::value;
            auto sample = [&](int i) {
                const float cur = sine_m;  // OSC_M.sine[t]: feeds the carrier now and, through the z^-1 ring, the modulator of t+1
                float out = 0.0f;
                osc_step(FC, sc, kc, cur * c_ix, 0.0f, out, sq, sw);      // carrier of sample t
                pos_m = sm.pos;
                osc_step(FM, sm, km, cur * c_fb, 0.0f, sine_m, sq, sw);   // modulator of sample t+1 (independent of the carrier)
                fed = cur;
                emit_put<kOut>(em, mix_tile, out, i, V);
            };
            if (decltype(whole)::value) {  // straight-line code over several samples: the tail of one sample's chains overlaps the head of the next's
#pragma unroll SRK_FM_UNROLL
                for (int i = 0; i < kMixRows; i++) sample(i);
            } else {
                for (int i = 0; i < n; i++) sample(i);
            }
        };
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        if (proved && n == kMixRows)
            fm_proved_tile<fo, fo_carrier>(facts, [&](auto m, auto c) { tile(m, c, true_type{}); });
        else if (proved)
            fm_proved_tile<fo, fo_carrier>(facts, [&](auto m, auto c) { tile(m, c, false_type{}); });
        else if (n == kMixRows)
            tile(integral_constant<uint32_t, fo>{}, integral_constant<uint32_t, fo_carrier>{}, true_type{});
        else
            tile(integral_constant<uint32_t, fo>{}, integral_constant<uint32_t, fo_carrier>{}, false_type{});
        emit_flush<kOut>(em, mix_tile, t0, n, V);
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

// ---- the z^-1 FM pair on TWO waves per 64 voices ------------------------------------------------------------------------------
// The carrier is not part of the feedback loop: it only consumes the modulator's sine.  Wave 0 of the workgroup runs the modulators of
// its 64 voices (the recurrence), wave 1 their carriers, frames and mix; the sines cross through a double-buffered LDS tile, one
// barrier per 32 samples: while the carrier wave works on tile k - 1 the modulator wave produces tile k.  65 536 voices are then 2048
// waves — two per SIMD instead of one, each with half the f64 work — and the other wave's instructions fill the gaps a lone wave's
// dependent chains leave.  Same arithmetic, same proofs per wave (each about its own oscillator) as render_fm_pair; results are
// bit-identical to it.  MEASURED (config 4, one box, tools/ab_env.sh SRACK_FM_SPLIT "0 1"): exact mode 55.4 -> 44.0 ms per step — its
// correctly rounded 2^cv, library sine and division are long serial chains behind wave-uniform branches —, default mode 7.29 -> 7.57:
// render_fm_pair already interleaves its four chains by hand, and the LDS hand-over and the barrier only add to it (swapping the two
// jobs in every other workgroup, by any bit of its index, to balance the SIMDs: 7.6 - 7.8).  So the launch code takes this kernel in
// exact mode only.
struct OscFacts {
    bool tame = false, small = false;  // as FmFacts, for one oscillator
};
SRK_DEV OscFacts fm_osc_facts(float gain, const dev::OscConst& k, double pos)
{
    const float e = __builtin_fabsf(gain) + __builtin_fabsf((float)k.val);
    OscFacts f;
    f.tame = __builtin_amdgcn_ballot_w64(!(e < 1000.0f && pos >= 0.0 && pos < 1.0 && k.sr >= 1.0 && dev::osc_below_rate(gain, k.val, k.sr))) == 0;
    f.small = __builtin_amdgcn_ballot_w64(!(e <= 0.4999f)) == 0;
    return f;
}
template <bool kExact, int kOut>
__global__ __launch_bounds__(128) void render_fm_pair_split(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    using std::integral_constant;
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
    __shared__ float sines[2][kMixRows * 64];
    const int lane = (int)(threadIdx.x & 63u);
    const bool carrier = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t voice = wm.voice, vc = wm.vc, V = a.V;
    const bool active = wm.active;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };
    auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
    const DevOp& oscop = a.ops[carrier ? r.osc_a : r.osc_l];   // this wave's oscillator (roles as in render_fm_pair)
    const DevOp& mulop = a.ops[carrier ? r.vca : r.adsr];      // the Multiply in front of its CV: x index / x feedback gain
    const int plane = a.ops[r.out].aux;
    const int ring_row = r.track;

    constexpr uint32_t fo = OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE | (kExact ? OSC_EXACT : 0u);
    constexpr uint32_t fo_carrier = fo | OSC_SINE_LOOSE;
    OscRegs s;
    OscConst k;
    s.pos = make_f64(row(oscop.state_row + OSC_S_POS_LO), row(oscop.state_row + OSC_S_POS_HI));
    s.sync_last = false;
    k.sr = oscop.sample_rate;
    k.val = (double)parv(oscop, OSC_P_VAL);
    k.delta = 0.0;
    k.inv_dt = 0.0f;
    const float gain = parv(mulop, MATH_P_CONST);
    float fed = __uint_as_float(row(ring_row));  // (modulator wave) OSC_M.sine of the previous tick
    Emit em = make_emit(a, plane, lane);         // (carrier wave)
    float sq = 0.0f, sw = 0.0f;
    OscFacts facts = kExact ? OscFacts{} : fm_osc_facts(gain, k, s.pos);

    const uint32_t n_tiles = (a.T + (uint32_t)kMixRows - 1u) / (uint32_t)kMixRows;
    for (uint32_t kt = 0; kt <= n_tiles; kt++) {
        if (!carrier) {
            if (kt < n_tiles) {
                const uint32_t t0 = kt * (uint32_t)kMixRows;
                const int n = (int)min((uint32_t)kMixRows, a.T - t0);
                float* const dst = sines[kt & 1u] + lane;
                // the tile's first fed-back value is the host's at the start of a launch, a sine afterwards: the bound is checked where it enters
                const bool proved = !kExact && facts.tame && n == kMixRows && __builtin_amdgcn_ballot_w64(!(__builtin_fabsf(fed) <= 1.0f)) == 0;
                auto tile = [&](auto flags_c) {
                    constexpr uint32_t F = decltype(flags_c)::value;
#pragma unroll SRK_FM_UNROLL
                    for (int i = 0; i < kMixRows; i++) {
                        float sine = 0.0f;
                        osc_step(F, s, k, fed * gain, 0.0f, sine, sq, sw);
                        fed = sine;
                        dst[i * 64] = sine;
                    }
                };
                if (proved && facts.small)
                    tile(integral_constant<uint32_t, fo | OSC_PHASE_TAME | OSC_CV_SMALL>{});
                else if (proved)
                    tile(integral_constant<uint32_t, fo | OSC_PHASE_TAME>{});
                else if (n == kMixRows)
                    tile(integral_constant<uint32_t, fo>{});
                else
                    for (int i = 0; i < n; i++) {
                        float sine = 0.0f;
                        osc_step(fo, s, k, fed * gain, 0.0f, sine, sq, sw);
                        fed = sine;
                        dst[i * 64] = sine;
                    }
                if (!kExact && !proved) facts = fm_osc_facts(gain, k, s.pos);  // the literal forms may have left [0, 1)
            }
        } else if (kt > 0) {
            const uint32_t t0 = (kt - 1u) * (uint32_t)kMixRows;
            const int n = (int)min((uint32_t)kMixRows, a.T - t0);
            const float* const src = sines[(kt - 1u) & 1u] + lane;
            float in[kMixRows];
#pragma unroll
            for (int i = 0; i < kMixRows; i++) in[i] = src[i * 64];  // (rows past a short last tile: stale, unused)
            bool proved = false;
            if (!kExact && facts.tame && n == kMixRows) {  // a sine from a phase outside [0, 1) is not bounded by 1: look
                float m = 0.0f;
#pragma unroll
                for (int i = 0; i < kMixRows; i++) m = __builtin_fmaxf(m, __builtin_fabsf(in[i]));
                proved = __builtin_amdgcn_ballot_w64(!(m <= 1.0f)) == 0;
            }
            auto tile = [&](auto flags_c) {
                constexpr uint32_t F = decltype(flags_c)::value;
#pragma unroll SRK_FM_UNROLL
                for (int i = 0; i < kMixRows; i++) {
                    float out = 0.0f;
                    osc_step(F, s, k, in[i] * gain, 0.0f, out, sq, sw);
                    emit_put<kOut>(em, mix_tile, out, i, V);
                }
            };
            if (proved && facts.small)
                tile(integral_constant<uint32_t, fo_carrier | OSC_PHASE_TAME | OSC_CV_SMALL>{});
            else if (proved)
                tile(integral_constant<uint32_t, fo_carrier | OSC_PHASE_TAME>{});
            else if (n == kMixRows)
                tile(integral_constant<uint32_t, fo_carrier>{});
            else {
#pragma unroll
                for (int i = 0; i < kMixRows; i++)
                    if (i < n) {
                        float out = 0.0f;
                        osc_step(fo_carrier, s, k, in[i] * gain, 0.0f, out, sq, sw);
                        emit_put<kOut>(em, mix_tile, out, i, V);
                    }
            }
            if (!kExact && !proved) facts = fm_osc_facts(gain, k, s.pos);
            emit_flush<kOut, false>(em, mix_tile, t0, n, V);
        }
        __syncthreads();  // tile kt is complete and visible; the carrier is done with the buffer tile kt + 1 will overwrite
    }
    if (active) {
        put(oscop.state_row + OSC_S_POS_LO, f64_lo(s.pos));
        put(oscop.state_row + OSC_S_POS_HI, f64_hi(s.pos));
        put(oscop.state_row + OSC_S_SYNC_LAST, 0u);
        if (!carrier) put(ring_row, __float_as_uint(fed));
    }
}

// ---- the same FM pair with the app's default delay: buffer_size >= 32, the ring in HBM ---------------------------------
// The modulator of sample t reads what it produced buffer_size samples ago (ring[(n0 + t) mod B], [B][V] f32, voice-
// minor) and overwrites it.  A 32-sample tile's reads were all written before the tile began (B >= 32), so they are
// issued together at the tile's start; the modulator no longer depends on its own previous sample, which leaves the
// compiler free to overlap modulator and carrier of neighbouring samples.
template <bool kExact, int kOut>
__global__ __launch_bounds__(64) void render_fm_pair_ring(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
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
    const uint32_t B = (uint32_t)a.prog.buffer_size;
    float* ring = a.rings + (size_t)r.track * B * V;  // r.track: the ring's id

    constexpr uint32_t fo = OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE | (kExact ? OSC_EXACT : 0u);
    constexpr uint32_t fo_carrier = fo | OSC_SINE_LOOSE;  // the carrier's sine only feeds the OutputModule (the matched shape): nothing integrates it
    OscRegs sm, sc;
    OscConst km, kc;
    sm.pos = make_f64(row(om.state_row + OSC_S_POS_LO), row(om.state_row + OSC_S_POS_HI));
    sm.sync_last = false;
    sc.pos = make_f64(row(ocr.state_row + OSC_S_POS_LO), row(ocr.state_row + OSC_S_POS_HI));
    sc.sync_last = false;
    km.sr = om.sample_rate;
    km.val = (double)parv(om, OSC_P_VAL);
    km.delta = 0.0;
    km.inv_dt = 0.0f;
    kc = km;
    kc.sr = ocr.sample_rate;
    kc.val = (double)parv(ocr, OSC_P_VAL);
    if (!kExact) {
        km.scale = (440.0 / km.sr) * exp2(km.val);
        kc.scale = (440.0 / kc.sr) * exp2(kc.val);
    }
    const float c_fb = parv(ofb, MATH_P_CONST), c_ix = parv(oix, MATH_P_CONST);

    Emit em = make_emit(a, plane, lane);
    float sq = 0.0f, sw = 0.0f;
    uint32_t p0 = (uint32_t)(a.n0 % B);  // ring position of the tile's first sample
    auto ring_at = [&](uint32_t p) { return p < B ? p : p - B; };
    auto load_tile = [&](float (&dst)[kMixRows], uint32_t first) {  // (past the end of a short last tile: harmless re-reads of valid ring rows)
#pragma unroll
        for (int i = 0; i < kMixRows; i++) dst[i] = ring[(size_t)ring_at(first + (uint32_t)i) * V + vc];
    };
    // With buffer_size >= 64 the NEXT tile's 32 values are older than this tile too: they are fetched while this tile computes
    // (one wave per SIMD at 65 536 voices: nobody else hides the 32 loads' latency; the app's 1024: 12.7 -> see NOTES.md section 4).
    const bool ahead = B >= 2u * (uint32_t)kMixRows;
    float fed[kMixRows];
    load_tile(fed, p0);
    FmFacts facts = kExact ? FmFacts{} : fm_facts(c_fb, km, sm.pos, c_ix, kc, sc.pos);  // as in render_fm_pair; re-proved below once a tile left them
    for (uint32_t t0 = 0; t0 < a.T; t0 += kMixRows) {
        const int n = (int)min((uint32_t)kMixRows, a.T - t0);
        const bool more = t0 + kMixRows < a.T;
        float nxt[kMixRows];
        if (ahead && more) load_tile(nxt, ring_at(p0 + (uint32_t)kMixRows));
        auto tile = [&](auto fm_c, auto fc_c, auto whole) {
            constexpr uint32_t FM = decltype(fm_c)::value, FC = decltype(fc_c)::value;
            auto sample = [&](int i) {
                float sine_m = 0.0f, out = 0.0f;
                osc_step(FM, sm, km, fed[i] * c_fb, 0.0f, sine_m, sq, sw);   // modulator of sample t
                const uint32_t p = p0 + (uint32_t)i < B ? p0 + (uint32_t)i : p0 + (uint32_t)i - B;
                if (active) ring[(size_t)p * V + voice] = sine_m;
                osc_step(FC, sc, kc, sine_m * c_ix, 0.0f, out, sq, sw);      // carrier of sample t
                emit_put<kOut>(em, mix_tile, out, i, V);
            };
            if (decltype(whole)::value) {
#pragma unroll SRK_FM_UNROLL
                for (int i = 0; i < kMixRows; i++) sample(i);
            } else {
#pragma unroll
                for (int i = 0; i < kMixRows; i++)
                    if (i < n) sample(i);
            }
        };
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        // the ring's first lap holds whatever the host put there (a rack file's saved buffers), not sines: the bound on the CVs (fm_facts)
        // holds for a tile whose fed-back values are at most 1 in magnitude — one v_max per sample, wave-uniform per tile.  (A ragged
        // last tile takes the same arithmetic as a whole one: a render must not depend on where it was split.)
        bool fed_unit = false;
        if (!kExact && facts.tame) {
            float m = 0.0f;
#pragma unroll
            for (int i = 0; i < kMixRows; i++) m = __builtin_fmaxf(m, i < n ? __builtin_fabsf(fed[i]) : 0.0f);
            // (max skips a NaN, and may: a NaN CV makes the increment and then the phase NaN through either form of 2^x and of the wrap)
            fed_unit = __builtin_amdgcn_ballot_w64(!(m <= 1.0f)) == 0;
        }
        if (fed_unit && n == kMixRows)
            fm_proved_tile<fo, fo_carrier>(facts, [&](auto m, auto c) { tile(m, c, true_type{}); });
        else if (fed_unit)
            fm_proved_tile<fo, fo_carrier>(facts, [&](auto m, auto c) { tile(m, c, false_type{}); });
        else if (n == kMixRows)
            tile(integral_constant<uint32_t, fo>{}, integral_constant<uint32_t, fo_carrier>{}, true_type{});
        else
            tile(integral_constant<uint32_t, fo>{}, integral_constant<uint32_t, fo_carrier>{}, false_type{});
        if (!kExact && !fed_unit) facts = fm_facts(c_fb, km, sm.pos, c_ix, kc, sc.pos);  // the literal forms may have left [0, 1)
        emit_flush<kOut>(em, mix_tile, t0, n, V);
        p0 = ring_at(p0 + (uint32_t)n);
        if (more) {
            if (ahead) {
#pragma unroll
                for (int i = 0; i < kMixRows; i++) fed[i] = nxt[i];
            } else {
                load_tile(fed, p0);
            }
        }
    }
    if (active) {
        auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
        put(om.state_row + OSC_S_POS_LO, f64_lo(sm.pos));
        put(om.state_row + OSC_S_POS_HI, f64_hi(sm.pos));
        put(om.state_row + OSC_S_SYNC_LAST, 0u);
        put(ocr.state_row + OSC_S_POS_LO, f64_lo(sc.pos));
        put(ocr.state_row + OSC_S_POS_HI, f64_hi(sc.pos));
        put(ocr.state_row + OSC_S_SYNC_LAST, 0u);
    }
}

// ---- the same FM pair, TIME-PARALLEL: buffer_size 256 ... 1024, the ring in LDS (round 3) ----------------------------------------------
// With a delay of B samples the modulator's pitch CV for the next B samples is already in the ring: beta * sine[t - B].  So for a
// chunk of L <= B samples every modulator increment is known up front, and the modulator's phase at sample i is a PREFIX SUM of
// increments (mod 1) instead of a recurrence: pos[i] = frac(pos[0] + sum_{j<i} delta[j]).  Its sines then follow independently,
// the carrier's increments from them, the carrier's phases by a second prefix sum.  Nothing in a chunk waits for its own past.
//   * One workgroup of 512 threads owns 32 voices for the whole launch: lane (g, s) = voice g of the 32, time slice s of 16.  A chunk
//     is 256 samples; slice s holds samples [16 s, 16 s + 16) of it: 16 increments in registers, a local prefix, the 16 slice totals
//     exchanged through LDS (two barriers per chunk), every lane adds the totals below its own.
//   * 2048 workgroups x 8 waves instead of 1024 lone waves: two waves per SIMD, each with 16 independent evaluations of 2^x and of
//     the sine in flight — the z^-1 kernel's one wave per SIMD is bound by the latency of its own dependency chains (NOTES.md section 4).
//   * The ring ([B][32] f32 = 128 KB at the app's buffer_size 1024) lives in LDS for the whole launch: loaded from HBM once, stored back
//     once.  render_fm_pair_ring reads and writes it in HBM every sample (8 B per voice-sample: 3.0 x the algorithmic bytes); here the
//     launch is the whole render segment and the ring costs 8 B per voice and B samples of it.
//   * Frames leave as 128-byte rows per (sample, workgroup): a wave stores two of them per instruction (its two slices).
//   * Arithmetic: the proved per-oscillator loops of render_fm_pair (modules.hip.h, fm_class_flags: scale * 2^cv with val folded, no
//     range reduction for |cv| <= 1/2, (2^(cv/4))^4 for |cv| <= 2, one-instruction wrap), voted per workgroup and launch; the literal
//     forms otherwise.  Default mode only: a prefix sum associates differently from the reference's recurrence — 1e-16 in a phase, nine
//     orders below the contract, but not bit-identical, and a render split between calls sums in other groups than an unsplit one.
// Reference: oscillator.rs:43-48,132-153 (the two oscillators), math.rs:152 (the two gains), synth.rs:164-192 (which edge is delayed).
constexpr int kBlkVoices = 32, kBlkSlices = 16, kBlkPer = 16, kBlkChunk = kBlkSlices * kBlkPer;  // 256 samples per chunk
SRK_DEV size_t fm_block_lds_bytes(uint32_t B) { return sizeof(float) * B * kBlkVoices + sizeof(double) * 2 * kBlkSlices * kBlkVoices + 16; }

// The polynomials' coefficients as REGISTERS.  A VOP3 f64 fma cannot take a 64-bit literal, and with sixteen evaluations in flight the
// compiler rematerialised every coefficient at every use (s_mov + v_mov: a fifth of the chunk's instructions).  Loaded once per launch and
// made opaque, they stay where they are; the operations are exp2_fast's and sine_fast's own (modules.hip.h), bit for bit.
struct BlkConsts {
    double e[10];  // 2^f: degree 8 (e[9] unused) or, loaded with kSeries9, degree 9 (exp2_fast9's coefficients)
    double q[7];   // sin(2 pi x) / x in x^2, degree 6
};
template <bool kSeries9 = false>
SRK_DEV void blk_consts_load(BlkConsts& K)
{
    const double e8[10] = {1.0000000000000004, 0.6931471805459332, 0.24022650695814518, 0.055504109412108156, 0.009618129159053034,
                           0.0013333450563173552, 0.00015403456082633648, 1.5310080611926545e-05, 1.3255179556479267e-06, 0.0};
    const double e9[10] = {0x1.000000000003dp+0, 0x1.62e42fefa39f7p-1, 0x1.ebfbdff8149f2p-3, 0x1.c6b08d7044119p-5, 0x1.3b2ab72b175eep-7,
                           0x1.5d87fe908f88ap-10, 0x1.43088e257f341p-13, 0x1.ffcb76789860fp-17, 0x1.63ef969a64d3cp-20, 0x1.b6571de2f2351p-24};
    const double q[7] = {6.283185307179272, -41.34170223990684, 81.60524914955879, -76.70584757807868, 42.05813586028645, -15.081496425342264, 3.6659216216293173};
    K.e[9] = 0.0;
#pragma unroll
    for (int i = 0; i < (kSeries9 ? 10 : 9); i++) {
        K.e[i] = kSeries9 ? e9[i] : e8[i];
        asm volatile("" : "+v"(K.e[i]));
    }
#pragma unroll
    for (int i = 0; i < 7; i++) {
        K.q[i] = q[i];
        asm volatile("" : "+v"(K.q[i]));
    }
}
template <bool kReduce>
SRK_DEV double blk_exp2(const BlkConsts& K, double x)  // == dev::exp2_fast<kReduce>(x)
{
    const double n = kReduce ? __builtin_rint(x) : 0.0;
    const double f = kReduce ? x - n : x;
    const double f2 = f * f;
    const double a01 = __builtin_fma(K.e[1], f, K.e[0]);
    const double a23 = __builtin_fma(K.e[3], f, K.e[2]);
    const double a45 = __builtin_fma(K.e[5], f, K.e[4]);
    const double a67 = __builtin_fma(K.e[7], f, K.e[6]);
    const double f4 = f2 * f2;
    const double b0 = __builtin_fma(a23, f2, a01);
    const double b1 = __builtin_fma(a67, f2, a45);
    const double b2 = __builtin_fma(K.e[8], f4, b1);
    const double p = __builtin_fma(b2, f4, b0);
    return kReduce ? __builtin_ldexp(p, (int)n) : p;
}
template <bool kReduce>
SRK_DEV double blk_exp2_9(const BlkConsts& K, double x)  // == dev::exp2_fast9<kReduce>(x); K loaded with kSeries9
{
    const double n = kReduce ? __builtin_rint(x) : 0.0;
    const double f = kReduce ? x - n : x;
    const double f2 = f * f;
    const double a01 = __builtin_fma(K.e[1], f, K.e[0]);
    const double a23 = __builtin_fma(K.e[3], f, K.e[2]);
    const double a45 = __builtin_fma(K.e[5], f, K.e[4]);
    const double a67 = __builtin_fma(K.e[7], f, K.e[6]);
    const double a89 = __builtin_fma(K.e[9], f, K.e[8]);
    const double f4 = f2 * f2;
    const double b0 = __builtin_fma(a23, f2, a01);
    const double b1 = __builtin_fma(a67, f2, a45);
    const double b2 = __builtin_fma(a89, f4, b1);
    const double p = __builtin_fma(b2, f4, b0);
    return kReduce ? __builtin_ldexp(p, (int)n) : p;
}
SRK_DEV float blk_sine(const BlkConsts& K, double pos)  // == dev::sine_fast(pos)
{
    uint32_t sign;
    const double x = dev::sine_fold(pos, sign);
    const double z = x * x;
    const double a01 = __builtin_fma(K.q[1], z, K.q[0]);
    const double a23 = __builtin_fma(K.q[3], z, K.q[2]);
    const double a45 = __builtin_fma(K.q[5], z, K.q[4]);
    const double z2 = z * z;
    const double b0 = __builtin_fma(a23, z2, a01);
    const double b1 = __builtin_fma(K.q[6], z2, a45);
    const double z4 = z2 * z2;
    const double p = __builtin_fma(b1, z4, b0);
    return __uint_as_float(__float_as_uint((float)(p * x)) ^ sign);
}
// The carrier's sine — it only feeds the OutputModule (the matched shape), nothing integrates it — folded in f32: the phase's 6e-8 of
// f32 rounding is 4e-7 of sine, beside the polynomial's 2e-7 (sine_loose folds in f64: three f64-rate instructions more per sample).
SRK_DEV float blk_sine_loose(double pos)
{
    const float qn = 0.5f - (float)pos;
    const uint32_t sign = __float_as_uint(qn) & 0x80000000u;
    const float x = 0.25f - __builtin_fabsf(__builtin_fabsf(qn) - 0.25f);
    const float z = x * x;
    const float a01 = __builtin_fmaf(-41.34168243408203f, z, 6.2831854820251465f);
    const float a23 = __builtin_fmaf(-76.58116912841797f, z, 81.60247802734375f);
    const float z2 = z * z;
    const float p = __builtin_fmaf(__builtin_fmaf(39.75982666015625f, z2, a23), z2, a01);
    return __uint_as_float(__float_as_uint(p * x) ^ sign);
}
// the phase increment of one sample, as osc_step spells it for the proved flags (default mode)
template <uint32_t kFlags>
SRK_DEV double fm_increment(const BlkConsts& K, const dev::OscConst& c, float cv)
{
    static_assert((kFlags & OSC_VAL_FOLDED) != 0, "the time-parallel pair runs proved loops only");
    double p;
    constexpr bool d9 = (kFlags & OSC_CV_SERIES9) != 0;  // (the degree-9 series: K loaded with kSeries9)
    if (kFlags & OSC_CV_QUAD) {
        p = d9 ? blk_exp2_9<false>(K, (double)(cv * 0.25f)) : blk_exp2<false>(K, (double)(cv * 0.25f));
        p = p * p;
        p = p * p;
    } else if (kFlags & OSC_CV_SMALL) {
        p = d9 ? blk_exp2_9<false>(K, (double)cv) : blk_exp2<false>(K, (double)cv);
    } else {
        p = d9 ? blk_exp2_9<true>(K, (double)cv) : blk_exp2<true>(K, (double)cv);
    }
    return c.scale * p;
}
// lane i's value from lane i ^ kMask of its row of 16, through the DPP network (no LDS crossbar): the masks a halving butterfly over 16
// lanes can use are 1 and 2 (quad permutes), 7 (row_half_mirror: i <-> 7 - i) and 15 (row_mirror: i <-> 15 - i) — together they span the row
template <int kMask>
SRK_DEV float row_xchg(float v)
{
    constexpr int ctrl = kMask == 1 ? 0xB1 : kMask == 2 ? 0x4E : kMask == 7 ? 0x141 : 0x140;
    static_assert(kMask == 1 || kMask == 2 || kMask == 7 || kMask == 15, "not a single DPP control");
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false));
}

// (two waves per SIMD is all the LDS allows — one workgroup per CU —, so the kernel may as well use their 256 registers each: the
// sixteen evaluations per lane and the polynomials' constants stay in registers instead of being rematerialised)
template <int kOut>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void render_fm_pair_block(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    extern __shared__ __attribute__((aligned(16))) float blk_lds[];
    const uint32_t B = (uint32_t)a.prog.buffer_size, V = a.V;
    float* const ring_l = blk_lds;                                                   // [B][32]
    double* const sums = (double*)(blk_lds + (size_t)B * kBlkVoices);                 // [2][16][32]: per oscillator, per slice, per voice
    uint32_t* const flag = (uint32_t*)(sums + 2 * kBlkSlices * kBlkVoices);
    const int tid = (int)threadIdx.x, g = tid & 31, s = tid >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave holds slices 2 w and 2 w + 1 of all 32 voices
    const bool odd = (s & 1) != 0;
    const uint32_t voice0 = (blockIdx.x - a.block0) * (uint32_t)kBlkVoices;
    const uint32_t n_act = min((uint32_t)kBlkVoices, V - voice0);
    const bool active = (uint32_t)g < n_act;
    const uint32_t voice = voice0 + (uint32_t)g, vc = active ? voice : voice0 + n_act - 1u;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };

    const DevOp& ofb = a.ops[r.adsr];    // roles as in render_fm_pair_ring
    const DevOp& om = a.ops[r.osc_l];
    const DevOp& oix = a.ops[r.vca];
    const DevOp& ocr = a.ops[r.osc_a];
    const int plane = a.ops[r.out].aux;
    float* const ring = a.rings + (size_t)r.track * B * V;

    constexpr uint32_t fo = OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE;
    constexpr uint32_t fo_carrier = fo | OSC_SINE_LOOSE;
    OscConst km, kc;
    double pos_m = make_f64(row(om.state_row + OSC_S_POS_LO), row(om.state_row + OSC_S_POS_HI));
    double pos_c = make_f64(row(ocr.state_row + OSC_S_POS_LO), row(ocr.state_row + OSC_S_POS_HI));
    km.sr = om.sample_rate;
    km.val = (double)parv(om, OSC_P_VAL);
    km.delta = 0.0;
    km.inv_dt = 0.0f;
    kc = km;
    kc.sr = ocr.sample_rate;
    kc.val = (double)parv(ocr, OSC_P_VAL);
    km.scale = (440.0 / km.sr) * exp2(km.val);
    kc.scale = (440.0 / kc.sr) * exp2(kc.val);
    const float c_fb = parv(ofb, MATH_P_CONST), c_ix = parv(oix, MATH_P_CONST);
    BlkConsts K;
    blk_consts_load(K);

    // the ring: HBM -> LDS, looking at every value on the way (the host's may be anything; a sine is at most 1)
    if (tid == 0) *flag = 0u;
    __syncthreads();
    bool unit = true;
    for (uint32_t p = (uint32_t)s; p < B; p += (uint32_t)kBlkSlices) {
        const float x = ring[(size_t)p * V + vc];
        ring_l[p * kBlkVoices + (uint32_t)g] = x;
        unit = unit && __builtin_fabsf(x) <= 1.0f;
    }
    if (!unit) atomicOr(flag, 1u);
    __syncthreads();
    // every wave holds all 32 voices (two slices of each): its ballots speak for the workgroup
    const FmFacts facts = fm_facts(c_fb, km, pos_m, c_ix, kc, pos_c);
    // A prefix sum adds a chunk's increments BEFORE it wraps: fine while they are of ordinary size (256 x 2^20 cycles still leaves 2^-24 of
    // a cycle), wrong for the huge ones a gain of hundreds of octaves produces — the recurrence wraps after every step and carries on from
    // the fraction, a sum would swallow every later increment.  A workgroup with such a voice (or with host values above 1 in its ring, or
    // a phase outside [0, 1)) renders through the recurrence itself, below: one lane per voice, the literal forms, as render_fm_pair_ring.
    const double biggest = __builtin_fmax(km.scale * exp2((double)__builtin_fabsf(c_fb)), kc.scale * exp2((double)__builtin_fabsf(c_ix)));
    const bool sane = facts.tame && *flag == 0u && __builtin_amdgcn_ballot_w64(!(biggest <= 1048576.0)) == 0;
    uint32_t p0 = (uint32_t)(a.n0 % B);
    const uint32_t i0 = (uint32_t)s * (uint32_t)kBlkPer;  // this lane's first sample of a chunk
    float* const frame_base = a.frames ? a.frames + (size_t)plane * a.plane_stride + voice0 : nullptr;
    float* const mp = a.mixpart ? a.mixpart + ((size_t)plane * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride : nullptr;
    const bool frames = kOut == 0 ? frame_base != nullptr : (kOut & 1) != 0;
    const bool mix = kOut == 0 ? mp != nullptr : (kOut & 2) != 0;

    // Where a lane's slice starts: the phase at the chunk's first sample + the totals of the slices below it.  The 16 slice totals of a
    // voice sit in 8 waves, two per wave: each wave leaves one total per voice (its two slices') in LDS, a lane adds the waves below
    // its own — a wave-uniform count: scalar branches, no selects — and the odd slice its even neighbour's, 32 lanes away.
    auto slice_base = [&](int osc, double mine, double start, double& base, double& total) {
        const double pair = mine + __shfl_xor(mine, 32);
        if (!odd) sums[(osc * 8 + w) * kBlkVoices + g] = pair;
        __syncthreads();
        base = start;
        total = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < 8; w2++) {
            const double v = sums[(osc * 8 + w2) * kBlkVoices + g];
            total += v;
            if (w2 < w) base += v;
        }
        if (odd) base += pair - mine;  // (the even neighbour's total; exact enough: the same additions in another order are not bit-identical anyway)
    };

    auto run = [&](auto fm_c, auto fc_c) {
        constexpr uint32_t FM = decltype(fm_c)::value, FC = decltype(fc_c)::value;
        auto chunk = [&](uint32_t t0, uint32_t n, auto fast_c) {
            // kFast: all 256 samples are the render's, every slice's 16 ring rows are contiguous (no wrap inside a slice: the ring's position
            // and length are multiples of 16) and all 32 voices are real — no masks, no selects, ring addresses as instruction offsets
            constexpr bool kFast = decltype(fast_c)::value;
            uint32_t idx[kBlkPer];
            double pre[kBlkPer];
            // ---- modulator: increments from what the ring holds, local prefix ----
            double acc = 0.0;
            if (kFast) {
                uint32_t p = p0 + i0;
                p = p >= B ? p - B : p;
                const uint32_t first = p * (uint32_t)kBlkVoices + (uint32_t)g;
#pragma unroll
                for (int k = 0; k < kBlkPer; k++) idx[k] = first + (uint32_t)(k * kBlkVoices);
            } else {
#pragma unroll
                for (int k = 0; k < kBlkPer; k++) {
                    uint32_t p = p0 + i0 + (uint32_t)k;
                    p = p >= B ? p - B : p;
                    idx[k] = p * (uint32_t)kBlkVoices + (uint32_t)g;
                }
            }
            float fed[kBlkPer];
#pragma unroll
            for (int k = 0; k < kBlkPer; k++) fed[k] = ring_l[idx[k]];
#pragma unroll
            for (int k = 0; k < kBlkPer; k++) {
                double d = fm_increment<FM>(K, km, fed[k] * c_fb);
                if (!kFast) {
                    asm("" : "+v"(d));  // (computed in every lane: a select, not sixteen branches that would fence the evaluations off from each other)
                    d = i0 + (uint32_t)k < n ? d : 0.0;
                }
                pre[k] = acc;
                acc += d;
            }
            double base, total;
            slice_base(0, acc, pos_m, base, total);
            pos_m = __builtin_amdgcn_fract(pos_m + total);
            // ---- modulator sines -> ring; carrier increments, local prefix ----
            acc = 0.0;
#pragma unroll
            for (int k = 0; k < kBlkPer; k++) {
                const double pm = __builtin_amdgcn_fract(base + pre[k]);
                const float sine = blk_sine(K, pm);
                double d = fm_increment<FC>(K, kc, sine * c_ix);
                if (kFast) {
                    ring_l[idx[k]] = sine;
                } else {
                    asm("" : "+v"(d));
                    const bool live = i0 + (uint32_t)k < n;
                    if (live) ring_l[idx[k]] = sine;
                    d = live ? d : 0.0;
                }
                pre[k] = acc;
                acc += d;
            }
            slice_base(1, acc, pos_c, base, total);
            pos_c = __builtin_amdgcn_fract(pos_c + total);
            // ---- carrier sines: frames and the workgroup's mix partial ----
            float out[kBlkPer];
#pragma unroll
            for (int k = 0; k < kBlkPer; k++) out[k] = blk_sine_loose(__builtin_amdgcn_fract(base + pre[k]));
            if (frames && (kFast || active)) {
                // one descriptor per wave and chunk: its two slices' first rows; lane offset = voice (+ 16 rows for the odd slice), scalar offset = row k
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(frame_base + (size_t)(t0 + (uint32_t)w * 2u * (uint32_t)kBlkPer) * V, 0, 0x7fffffff, 0x00020000);
                const uint32_t voff = ((uint32_t)g + (odd ? (uint32_t)kBlkPer * V : 0u)) * 4u;
#pragma unroll
                for (int k = 0; k < kBlkPer; k++)
                    if (kFast || i0 + (uint32_t)k < n) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[k]), rsrc, (int)voff, (int)((uint32_t)k * V * 4u), SRK_FRAME_AUX);
            }
            if (mix) {
                // sum over the 32 voices of each of this slice's 16 samples: a halving butterfly — 16 -> 8 -> 4 -> 2 -> 1 values per lane
                // while the lanes pair up (i ^ 15, i ^ 7, i ^ 2, i ^ 1: single DPP controls) — then the two rows of 16 lanes: lane g ends with sample (g & 15)
                float v[kBlkPer];
#pragma unroll
                for (int k = 0; k < kBlkPer; k++) v[k] = (kFast || active) ? out[k] : 0.0f;
                auto halve = [&](auto mask_c, int bit, int h) {  // lanes i and i ^ mask share the work: the one with `bit` set keeps the upper half
                    const bool up = (g & bit) != 0;
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (j < h) {
                            const float send = up ? v[j] : v[j + h], keep = up ? v[j + h] : v[j];
                            v[j] = keep + row_xchg<decltype(mask_c)::value>(send);
                        }
                };
                halve(std::integral_constant<int, 15>{}, 8, 8);
                halve(std::integral_constant<int, 7>{}, 4, 4);
                halve(std::integral_constant<int, 2>{}, 2, 2);
                halve(std::integral_constant<int, 1>{}, 1, 1);
                const float sum = v[0] + __shfl_xor(v[0], 16);
                if (g < kBlkPer && (kFast || i0 + (uint32_t)g < n)) mp[t0 + i0 + (uint32_t)g] = sum;
            }
            p0 += n;
            p0 = p0 >= B ? p0 - B : p0;
        };
        const bool fast = (B % (uint32_t)kBlkPer) == 0u && (p0 % (uint32_t)kBlkPer) == 0u && n_act == (uint32_t)kBlkVoices;  // (uniform; p0 stays a multiple of 16 over full chunks)
        uint32_t t0 = 0;
        if (fast)
            for (; t0 + (uint32_t)kBlkChunk <= a.T; t0 += (uint32_t)kBlkChunk) chunk(t0, (uint32_t)kBlkChunk, std::true_type{});
        for (; t0 < a.T; t0 += (uint32_t)kBlkChunk) chunk(t0, min((uint32_t)kBlkChunk, a.T - t0), std::false_type{});
    };
    if (sane) {
        fm_proved_tile<fo, fo_carrier>(facts, [&](auto m, auto c) { run(m, c); });
    } else if (s == 0) {  // the recurrence, sample by sample, one lane per voice (the other fifteen slices have nothing to do in this launch)
        OscRegs sm, sc;
        sm.pos = pos_m;
        sc.pos = pos_c;
        sm.sync_last = sc.sync_last = false;
        float sq = 0.0f, sw = 0.0f;
        for (uint32_t t = 0; t < a.T; t++) {
            const uint32_t at = p0 * (uint32_t)kBlkVoices + (uint32_t)g;
            float sine_m = 0.0f, out = 0.0f;
            osc_step(fo, sm, km, ring_l[at] * c_fb, 0.0f, sine_m, sq, sw);
            ring_l[at] = sine_m;
            osc_step(fo_carrier, sc, kc, sine_m * c_ix, 0.0f, out, sq, sw);
            if (frames && active) frame_base[(size_t)t * V + (uint32_t)g] = out;
            if (mix) {
                float v = active ? out : 0.0f;
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
                if (g == 0) mp[t] = v;
            }
            p0 = p0 + 1u == B ? 0u : p0 + 1u;
        }
        pos_m = sm.pos;
        pos_c = sc.pos;
    }
    __syncthreads();  // the last chunk's ring writes
    if (active) {
        for (uint32_t p = (uint32_t)s; p < B; p += (uint32_t)kBlkSlices) ring[(size_t)p * V + voice] = ring_l[p * kBlkVoices + (uint32_t)g];
        if (s == 0) {
            auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
            put(om.state_row + OSC_S_POS_LO, f64_lo(pos_m));
            put(om.state_row + OSC_S_POS_HI, f64_hi(pos_m));
            put(om.state_row + OSC_S_SYNC_LAST, 0u);
            put(ocr.state_row + OSC_S_POS_LO, f64_lo(pos_c));
            put(ocr.state_row + OSC_S_POS_HI, f64_hi(pos_c));
            put(ocr.state_row + OSC_S_SYNC_LAST, 0u);
        }
    }
}

// ---- the same, with the MODULATOR EXACT: time lanes for everything but two instructions (round 6) -----------------------------------------
// Default mode renders config 4 with the modulator exact as a whole (csrc/approx.cpp: its feedback loop runs through a pitch, where only the
// reference's own bits follow the reference for longer than seconds) — 2^cv by the host libm's pow operation for operation, the reference's
// sine —, and until round 6 that program was the general path's: one lane per voice, 48 000 samples in a row, one wave per SIMD, the ring
// through HBM both ways (22.2 ms per step, 3.0 x the algorithmic bytes).  But with a delay of B >= 256 samples the modulator's pitch CV of a
// whole chunk is in the ring when the chunk starts, exactly as for the kernel above; what an EXACT oscillator may not do is add its
// increments in another order.  So only that stays serial:
//     pos[t + 1] = (pos[t] + delta[t]) % 1.0            oscillator.rs:152-153 — one v_add_f64 and one v_fract_f64 per voice-sample,
// the reference's operations in the reference's order, and everything around it is evaluated across TIME lanes, each value by the
// reference's own expression: delta[t] = 440 * 2^(f64(cv[t]) + f64(val)) / sr (oscillator.rs:43-48,132: exp2_libm, div_rn) before the scan,
// sine[t] = (pos[t] * PI * 2).sin() as f32 (oscillator.rs:133: sine_exact) after it.  Bit-identical to osc_step's exact flavour, hence to
// the CPU tick.
//   * A workgroup of 512 threads owns 32 voices for the launch, lane (g, s) = voice g, time slice s of 16; a chunk is 64 samples, slice s
//     holds samples [4 s, 4 s + 4).  The ring ([B][32] f32, 128 KB at the app's 1024) lives in LDS for the whole launch: HBM sees the frames.
//   * The scan runs in the first 32 lanes of wave 0, through one [64][32] f64 buffer in LDS (increments in, phases out, in place), while
//     all eight waves do the rest — a three-stage pipeline over chunks, two barriers per chunk:
//         iteration k:   phases of chunk k -> registers, increments of chunk k + 1 -> buffer (a lane overwrites what it has just read) | barrier |
//                        scan(k + 1)  ||  carrier frames of chunk k - 1, sines of chunk k (-> ring, carrier increments), increments of chunk k + 2 -> registers | barrier
//   * The carrier behind the loop keeps the default forms of the kernel above (bounded-CV classes, prefix sum of its increments, f32
//     sine): nothing feeds it back.  Its slice totals cross between the waves one iteration late, on the pipeline's own barriers.
//   * A workgroup whose carrier cannot be proved tame (gains of hundreds of octaves, a phase outside [0, 1)) renders sample by sample
//     through osc_step, one lane per voice, like the kernel above; increments that are not finite send the scan through fmod1.
constexpr int kXPer = 4, kXChunk = kBlkSlices * kXPer;  // 64 samples per chunk
__host__ __device__ inline size_t fm_block_x_lds_bytes(uint32_t B) { return sizeof(float) * B * kBlkVoices + sizeof(double) * kXChunk * kBlkVoices + sizeof(double) * 2 * 8 * kBlkVoices + sizeof(uint64_t) * 256 + 16; }
struct LibmTabLds {
    const uint64_t* t;
    SRK_DEV uint64_t operator()(uint32_t i) const { return t[i]; }
};
// The exact forms' constants as REGISTERS (as BlkConsts above: an f64 VOP3 cannot take a 64-bit literal, and with four evaluations in flight the
// compiler rematerialised every one at every use — 95 of the chunk loop's 544 vector instructions were moves).  The operations below are
// dev::exp2_libm_plain_t's and dev::sine_exact_plain's own (modules.hip.h), operation for operation: same values, same roundings.
struct XConsts {
    double lhi, llo, inv_ln2n, shift, nln2hi, nln2lo, c2, c3, c4, c5, k440;  // 2^e as the host libm's pow
    double q[7], quarter, rel, abs_;                                          // the sine's polynomial, its fold, the decision's interval
};
SRK_DEV void x_consts_load(XConsts& X)
{
    X.lhi = 0x1.62e42fefa39efp-1, X.llo = 0x1.abc9e3b398000p-56, X.inv_ln2n = 0x1.71547652b82fep+7, X.shift = 0x1.8p52;
    X.nln2hi = -0x1.62e42fefa0000p-8, X.nln2lo = -0x1.cf79abc9e3b3ap-47;
    X.c2 = 0x1.ffffffffffdbdp-2, X.c3 = 0x1.555555555543cp-3, X.c4 = 0x1.55555cf172b91p-5, X.c5 = 0x1.1111167a4d017p-7, X.k440 = 440.0;
    const double q[7] = {6.283185307179272, -41.34170223990684, 81.60524914955879, -76.70584757807868, 42.05813586028645, -15.081496425342264, 3.6659216216293173};
#pragma unroll
    for (int i = 0; i < 7; i++) X.q[i] = q[i];
    X.quarter = 0.25, X.rel = 1.0e-13, X.abs_ = 2.0e-15;
    double* const all = &X.lhi;
    static_assert(sizeof(XConsts) == 21 * sizeof(double), "XConsts is 21 doubles in a row");
#pragma unroll
    for (int i = 0; i < 21; i++) asm volatile("" : "+v"(all[i]));
}
template <class Tab>
SRK_DEV double x_exp2_libm_plain(const XConsts& X, double e, bool& cold, const Tab& tab)  // == dev::exp2_libm_plain_t(e, cold, tab)
{
    const double ehi = e * X.lhi;
    const double elo = __builtin_fma(e, X.llo, __builtin_fma(X.lhi, e, -ehi));
    const uint32_t abstop = ((uint32_t)__double2hiint(ehi) >> 20) & 0x7ffu;
    cold = cold || !(abstop - 0x3c9u <= 0x3eu);
    const double kds = __builtin_fma(ehi, X.inv_ln2n, X.shift);
    const uint64_t ki = (uint64_t)__double_as_longlong(kds);
    const double kd = kds - X.shift;
    double r = __builtin_fma(kd, X.nln2lo, __builtin_fma(kd, X.nln2hi, ehi));
    r = elo + r;
    const uint32_t idx = 2u * ((uint32_t)ki & 127u);
    const double tail = __longlong_as_double((long long)tab(idx));
    const uint64_t sbits = tab(idx + 1u) + (ki << 45);
    const double r2 = r * r;
    const double a = __builtin_fma(r, X.c3, X.c2);
    const double b = r + tail;
    const double c = __builtin_fma(r, X.c5, X.c4);
    double tmp = __builtin_fma(a, r2, b);
    tmp = __builtin_fma(c, r2 * r2, tmp);
    const double scale = __longlong_as_double((long long)sbits);
    return __builtin_fma(tmp, scale, scale);
}
template <bool kRange>
SRK_DEV float x_sine_exact_plain(const XConsts& X, double pos, bool& cold)  // == dev::sine_exact_plain<kRange>(pos, cold)
{
    const double qn = 0.5 - pos;
    const uint32_t sign = (uint32_t)__double2hiint(qn) & 0x80000000u;
    const double t = __builtin_fabs(qn) - X.quarter;
    const double x = X.quarter - __builtin_fabs(t);
    const double z = x * x;
    const double a01 = __builtin_fma(X.q[1], z, X.q[0]);
    const double a23 = __builtin_fma(X.q[3], z, X.q[2]);
    const double a45 = __builtin_fma(X.q[5], z, X.q[4]);
    const double z2 = z * z;
    const double b0 = __builtin_fma(a23, z2, a01);
    const double b1 = __builtin_fma(X.q[6], z2, a45);
    const double z4 = z2 * z2;
    const double y = __builtin_fma(b1, z4, b0) * x;
    const double d = __builtin_fma(X.rel, y, X.abs_);
    const float r = (float)(y - d), r2 = (float)(y + d);
    bool sure = __float_as_uint(r) == __float_as_uint(r2);
    if (kRange) sure = sure && pos >= 0.0 && pos < 1.0;
    cold = cold || !sure;
    return __uint_as_float(__float_as_uint(r) ^ sign);
}

template <int kOut>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void render_fm_pair_block_x(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    extern __shared__ __attribute__((aligned(16))) float blk_lds[];
    const uint32_t B = (uint32_t)a.prog.buffer_size, V = a.V;
    float* const ring_l = blk_lds;                                                   // [B][32]
    double* const buf = (double*)(blk_lds + (size_t)B * kBlkVoices);                 // [64][32]: a chunk's modulator increments, then its phases
    double* const sums = buf + kXChunk * kBlkVoices;                                  // [2][8][32]: the carrier's slice totals, per chunk parity, per wave, per voice
    uint64_t* const tab = (uint64_t*)(sums + 2 * 8 * kBlkVoices);                     // the libm's 2^(k/128) table
    uint32_t* const flag = (uint32_t*)(tab + 256);                                    // [0]: some modulator increment is not an ordinary number; [1]: the ring came with a value above 1
    const int tid = (int)threadIdx.x, g = tid & 31, s = tid >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // this wave holds slices 2 w and 2 w + 1 of all 32 voices
    const bool odd = (s & 1) != 0;
    const uint32_t voice0 = (blockIdx.x - a.block0) * (uint32_t)kBlkVoices;
    const uint32_t n_act = min((uint32_t)kBlkVoices, V - voice0);
    const bool active = (uint32_t)g < n_act;
    const uint32_t voice = voice0 + (uint32_t)g, vc = active ? voice : voice0 + n_act - 1u;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };

    const DevOp& ofb = a.ops[r.adsr];    // roles as in render_fm_pair_ring
    const DevOp& om = a.ops[r.osc_l];
    const DevOp& oix = a.ops[r.vca];
    const DevOp& ocr = a.ops[r.osc_a];
    const int plane = a.ops[r.out].aux;
    float* const ring = a.rings + (size_t)r.track * B * V;

    constexpr uint32_t fo = OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE;
    constexpr uint32_t fo_mod = fo | OSC_EXACT, fo_carrier = fo | OSC_SINE_LOOSE;
    OscConst km, kc;
    double pos_m = make_f64(row(om.state_row + OSC_S_POS_LO), row(om.state_row + OSC_S_POS_HI));  // (scan lanes: the phase at the start of the next chunk to scan)
    double pos_c = make_f64(row(ocr.state_row + OSC_S_POS_LO), row(ocr.state_row + OSC_S_POS_HI));
    km.sr = om.sample_rate;
    km.val = (double)parv(om, OSC_P_VAL);
    km.delta = 0.0;
    km.inv_dt = 0.0f;
    kc = km;
    kc.sr = ocr.sample_rate;
    kc.val = (double)parv(ocr, OSC_P_VAL);
    kc.scale = (440.0 / kc.sr) * exp2(kc.val);
    const float c_fb = parv(ofb, MATH_P_CONST), c_ix = parv(oix, MATH_P_CONST);
    BlkConsts K;
    blk_consts_load<true>(K);  // (the carrier's 2^cv at degree 9: OSC_CV_SERIES9)
    XConsts X;
    x_consts_load(X);
    const LibmTabLds libm{tab};

    if (tid == 0) flag[0] = flag[1] = 0u;
    for (int k = tid; k < 256; k += kBlkVoices * kBlkSlices) tab[k] = kLibmExpTab[k];
    __syncthreads();
    bool unit = true;  // the ring: HBM -> LDS, looking at every value on the way (the host's may be anything; a sine is at most 1)
    for (uint32_t p = (uint32_t)s; p < B; p += (uint32_t)kBlkSlices) {
        const float x = ring[(size_t)p * V + vc];
        ring_l[p * kBlkVoices + (uint32_t)g] = x;
        unit = unit && __builtin_fabsf(x) <= 1.0f;
    }
    if (!unit) atomicOr(flag + 1, 1u);
    __syncthreads();
    // every wave holds all 32 voices (two slices of each): its ballots speak for the workgroup
    const OscFacts cf = fm_osc_facts(c_ix, kc, pos_c);
    const int car_class = fm_gain_class(c_ix);
    const double biggest = kc.scale * exp2((double)__builtin_fabsf(c_ix));  // (a prefix sum adds a chunk's increments before it wraps: see the kernel above)
    const bool sane = cf.tame && __builtin_amdgcn_ballot_w64(!(biggest <= 1048576.0)) == 0;
    // What the modulator's arithmetic may skip once it is PROVED for the launch: |cv| <= |gain| (the ring holds sines: looked at above) and
    // |gain| + |val| <= 800 put every exponent within pow's plain range on the large side and every increment, 440 * 2^e / sr, within the
    // normal range and below 2^52 — no test of the quotient, no look at the increments, a phase that stays in [0, 1): the wrap is the one
    // instruction, the sine needs no range check.  (The SMALL side of pow's plain range — |e ln 2| < 2^-54, e = 0 among them: a silent ring —
    // is an argument's own business and stays tested per sample.)
    const bool mod_ok = (double)__builtin_fabsf(c_fb) + __builtin_fabs(km.val) <= 800.0 && km.sr >= 1.0 && km.sr <= 65535.0 && pos_m >= 0.0 && pos_m < 1.0;
    const bool proved = flag[1] == 0u && __builtin_amdgcn_ballot_w64(!mod_ok) == 0;
    if (!(pos_m >= 0.0 && pos_m < 1.0)) atomicOr(flag, 1u);  // (only a host can store such a phase: the scan then wraps with fmod1)
    uint32_t p0 = (uint32_t)(a.n0 % B);
    const uint32_t i0 = (uint32_t)s * (uint32_t)kXPer;  // this lane's first sample of a chunk
    float* const frame_base = a.frames ? a.frames + (size_t)plane * a.plane_stride + voice0 : nullptr;
    float* const mp = a.mixpart ? a.mixpart + ((size_t)plane * a.n_waves + (blockIdx.x - a.block0)) * a.t_stride : nullptr;
    const bool frames = kOut == 0 ? frame_base != nullptr : (kOut & 1) != 0;
    const bool mix = kOut == 0 ? mp != nullptr : (kOut & 2) != 0;
    const double inv_sr = 1.0 / km.sr;

    // kFast: the modulator proved (above), every chunk whole (the launch's length a multiple of 64), no slice wraps inside the ring (its length
    // and position multiples of 4) and all 32 voices real — no masks, no selects, ring addresses as instruction offsets
    auto run = [&](auto fc_c, auto fast_c) {
        constexpr uint32_t FC = decltype(fc_c)::value;
        constexpr bool kFast = decltype(fast_c)::value;
        const uint32_t n_chunks = (a.T + (uint32_t)kXChunk - 1u) / (uint32_t)kXChunk;
        // the ring's word for this lane's sample j of a chunk whose first sample of this slice sits at ring position `at` (below B; B >= 4)
        auto ring_at = [&](uint32_t at, int j) {
            if (kFast) return (at + (uint32_t)j) * (uint32_t)kBlkVoices + (uint32_t)g;
            const uint32_t p = at + (uint32_t)j;
            return (p >= B ? p - B : p) * (uint32_t)kBlkVoices + (uint32_t)g;
        };
        auto ring_step = [&](uint32_t at) {  // ... one chunk on (B >= 256 > 64)
            const uint32_t p = at + (uint32_t)kXChunk;
            return p >= B ? p - B : p;
        };
        auto live = [&](uint32_t chunk, int j) { return kFast || chunk * (uint32_t)kXChunk + i0 + (uint32_t)j < a.T; };
        // the modulator's increments of a chunk, from what the ring holds: 440 * 2^(f64(cv) + f64(val)) / sr (oscillator.rs:43-48,132)
        // (`redo`: the lanes whose plain forms could not decide — the reference's expression itself, the same value wherever they had.  The
        // question "could some lane not decide" is asked ONCE per iteration, after its arithmetic, for increments and sines together: a branch
        // in the middle fences the iteration's two halves off from each other — 13.7 against 12.7 ms per step without any question)
        auto increments = [&](uint32_t chunk, uint32_t at, double (&d)[kXPer], bool& cold, bool redo) {
            float fed[kXPer];
#pragma unroll
            for (int j = 0; j < kXPer; j++) fed[j] = ring_l[ring_at(at, j)];
            double e[kXPer];
#pragma unroll
            for (int j = 0; j < kXPer; j++) {
                e[j] = (double)(fed[j] * c_fb) + km.val;
                if (redo) {
                    d[j] = osc_delta_exact_cold(e[j], km.sr);
                } else {
                    const double pw = X.k440 * x_exp2_libm_plain(X, e[j], cold, libm);
                    d[j] = kFast ? div_rn_proved(pw, km.sr, inv_sr) : div_rn_plain(pw, km.sr, cold);
                }
            }
            if (!kFast) {
                bool ordinary = true;
#pragma unroll
                for (int j = 0; j < kXPer; j++) {
                    ordinary = ordinary && d[j] >= 0.0 && d[j] < 4503599627370496.0;
                    d[j] = live(chunk, j) ? d[j] : 0.0;  // (past the launch's last sample: the phase stands still)
                }
                if (!ordinary) atomicOr(flag, 1u);
            }
        };
        // the scan: chunk's increments in `buf` -> its phases, in place; lanes 0 .. 31 of wave 0, one voice each
        auto scan = [&]() {
#ifdef SRK_X_NOSCAN
            return;  // (timing experiment only)
#endif
            if (tid >= kBlkVoices) return;
            double p = pos_m;
            double* const col = buf + g;
            if (kFast || flag[0] == 0u) {
#pragma unroll 1
                for (int t0 = 0; t0 < kXChunk; t0 += 16) {
                    double d[16];
#pragma unroll
                    for (int t = 0; t < 16; t++) d[t] = col[(t0 + t) * kBlkVoices];
#pragma unroll
                    for (int t = 0; t < 16; t++) {
                        col[(t0 + t) * kBlkVoices] = p;
                        p = __builtin_amdgcn_fract(p + d[t]);   // pos += delta; pos %= 1.0 — exact for ordinary increments and a phase in [0, 1)
                    }
                }
            } else {
                for (int t = 0; t < kXChunk; t++) {
                    const double d = col[t * kBlkVoices];
                    col[t * kBlkVoices] = p;
                    p = fmod1(p + d);
                }
            }
            pos_m = p;
        };
        // where a lane's slice starts (carrier): the phase at the chunk's first sample + the totals of the slices below it
        // (returns the OTHER slice of this wave's pair: the odd slice starts behind the even one's total — taken as it is, not as pair - mine: a
        // voice whose modulator has just overflowed has a NaN in ITS slice's total, and the samples before it must not inherit it)
        auto slice_totals = [&](int parity, double mine) {
            const double other = __shfl_xor(mine, 32);
            if (!odd) sums[(parity * 8 + w) * kBlkVoices + g] = mine + other;
            return other;
        };
        auto slice_base = [&](int parity, double mine, double other, double start, double& base, double& total) {
            // (the waves below this one: a wave-uniform count.  A product with a scalar 0 / 1 — one fma per wave where an addition behind a
            // select costs three instructions, two scalar-count loops more still (measured: 13.7 / 14.5 / 14.6 ms per step) — unless a total is
            // a NaN: a voice whose modulator has overflowed hands them on, and they must not reach the slices BEFORE them: then the selects)
            base = start;
            total = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < 8; w2++) {
                const double v = sums[(parity * 8 + w2) * kBlkVoices + g];
                total += v;
                base = __builtin_fma(v, w2 < w ? 1.0 : 0.0, base);
            }
            if (__builtin_amdgcn_ballot_w64(total != total) != 0) {
                base = start;
#pragma unroll
                for (int w2 = 0; w2 < 8; w2++) {
                    const double v = sums[(parity * 8 + w2) * kBlkVoices + g];
                    base = w2 < w ? base + v : base;
                }
            }
            (void)mine;
            if (odd) base += other;
        };
        // frames and mix partial of a chunk whose carrier phases are known
        auto emit = [&](uint32_t chunk, const double (&pre)[kXPer], double base) {
            const uint32_t t0 = chunk * (uint32_t)kXChunk;
            float out[kXPer];
#pragma unroll
            for (int j = 0; j < kXPer; j++) out[j] = blk_sine_loose(__builtin_amdgcn_fract(base + pre[j]));
            if (frames && (kFast || active)) {
                const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(frame_base + (size_t)(t0 + (uint32_t)w * 2u * (uint32_t)kXPer) * V, 0, 0x7fffffff, 0x00020000);
                const uint32_t voff = ((uint32_t)g + (odd ? (uint32_t)kXPer * V : 0u)) * 4u;
#pragma unroll
                for (int j = 0; j < kXPer; j++)
                    if (live(chunk, j)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(out[j]), rsrc, (int)voff, (int)((uint32_t)j * V * 4u), SRK_FRAME_AUX);
            }
            if (mix) {
                // sum over the 32 voices of each of this slice's 4 samples: 4 -> 2 -> 1 values per lane while lanes pair up (g ^ 1, g ^ 2), then
                // plain butterflies over the remaining voice bits: lane g ends with sample (g & 3) of its slice
                float v[kXPer];
#pragma unroll
                for (int j = 0; j < kXPer; j++) v[j] = (kFast || active) ? out[j] : 0.0f;
                {
                    const bool up = (g & 1) != 0;
                    const float s0 = up ? v[0] : v[2], s1 = up ? v[1] : v[3], k0 = up ? v[2] : v[0], k1 = up ? v[3] : v[1];
                    v[0] = k0 + row_xchg<1>(s0);
                    v[1] = k1 + row_xchg<1>(s1);
                }
                {
                    const bool up = (g & 2) != 0;
                    const float send = up ? v[0] : v[1], keep_ = up ? v[1] : v[0];
                    v[0] = keep_ + row_xchg<2>(send);
                }
                float sum = v[0];
                sum += __shfl_xor(sum, 4);
                sum += __shfl_xor(sum, 8);
                sum += __shfl_xor(sum, 16);
                const int mine_j = ((g & 1) ? 2 : 0) + ((g & 2) ? 1 : 0);  // which of the slice's samples this lane ended up with
                if (g < 4 && (kFast || t0 + i0 + (uint32_t)mine_j < a.T)) mp[t0 + i0 + (uint32_t)mine_j] = sum;
            }
        };

        // ---- prologue: increments of chunks 0 and 1, phases of chunk 0 ----
        double dM[kXPer];
        uint32_t at_k = (p0 + i0) % B;                     // ring position of this slice's first sample of chunk k ...
        uint32_t at_inc = at_k;                            // ... and of the chunk whose increments are computed next
        {
            bool cold = false;
            increments(0u, at_inc, dM, cold, false);
            if (__builtin_amdgcn_ballot_w64(cold) != 0) increments(0u, at_inc, dM, cold, true);
        }
        at_inc = ring_step(at_inc);
#pragma unroll
        for (int j = 0; j < kXPer; j++) buf[(i0 + (uint32_t)j) * kBlkVoices + (uint32_t)g] = dM[j];
        __syncthreads();
        scan();
        if (n_chunks > 1u) {
            bool cold = false;
            increments(1u, at_inc, dM, cold, false);
            if (__builtin_amdgcn_ballot_w64(cold) != 0) increments(1u, at_inc, dM, cold, true);
        }
        at_inc = ring_step(at_inc);
        __syncthreads();
        double preC[kXPer] = {0.0, 0.0, 0.0, 0.0}, pairC = 0.0, mineC = 0.0;  // the carrier's local prefix / totals of the chunk whose frames are still to come
        for (uint32_t k = 0; k < n_chunks; k++) {
            double pM[kXPer];
#pragma unroll
            for (int j = 0; j < kXPer; j++) pM[j] = buf[(i0 + (uint32_t)j) * kBlkVoices + (uint32_t)g];
            // (no barrier here: a lane overwrites exactly the entries it has just read; the scan, which reads everybody's, is behind the next one)
            if (k + 1u < n_chunks) {
#pragma unroll
                for (int j = 0; j < kXPer; j++) buf[(i0 + (uint32_t)j) * kBlkVoices + (uint32_t)g] = dM[j];
            }
            __syncthreads();
            if (k + 1u < n_chunks) scan();
            if (k > 0u) {  // the carrier's frames of chunk k - 1 (its slice totals crossed on the barriers above)
                double base, total;
                slice_base((int)((k - 1u) & 1u), mineC, pairC, pos_c, base, total);
                emit(k - 1u, preC, base);
                pos_c = __builtin_amdgcn_fract(pos_c + total);
            }
            // sines of chunk k -> ring; the carrier's increments, their local prefix and slice totals; the modulator's increments of chunk k + 2
            auto second_half = [&](bool& cold, bool redo) {
                float sine[kXPer];
#pragma unroll
                for (int j = 0; j < kXPer; j++) {
                    if (redo) {
                        double unused = 0.0;
                        osc_exact_cold(0.0, 1.0, pM[j], false, unused, sine[j]);
                    } else {
                        sine[j] = x_sine_exact_plain<!kFast>(X, pM[j], cold);  // (pos * PI * 2).sin() as f32, oscillator.rs:133
                    }
                }
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < kXPer; j++) {
                    double d = fm_increment<FC>(K, kc, sine[j] * c_ix);
                    const bool lv = live(k, j);
                    if (lv) ring_l[ring_at(at_k, j)] = sine[j];
                    if (!kFast) d = lv ? d : 0.0;
                    preC[j] = acc;
                    acc += d;
                }
                mineC = acc;
                pairC = slice_totals((int)(k & 1u), acc);
                if (k + 2u < n_chunks) increments(k + 2u, at_inc, dM, cold, redo);
            };
            bool cold = false;
            second_half(cold, false);
#ifndef SRK_X_NOCOLD
            if (__builtin_amdgcn_ballot_w64(cold) != 0) second_half(cold, true);  // (3.4e-6 of the samples: this wave's part of the iteration again, every value the reference's own)
#endif
            at_k = ring_step(at_k);
            at_inc = ring_step(at_inc);
            __syncthreads();
        }
        {   // the last chunk's frames
            double base, total;
            slice_base((int)((n_chunks - 1u) & 1u), mineC, pairC, pos_c, base, total);
            emit(n_chunks - 1u, preC, base);
            pos_c = __builtin_amdgcn_fract(pos_c + total);
        }
    };
    const bool fast = proved && (a.T % (uint32_t)kXChunk) == 0u && (B % (uint32_t)kXPer) == 0u && (p0 % (uint32_t)kXPer) == 0u && n_act == (uint32_t)kBlkVoices;
    if (a.T == 0u) {
        // nothing to render: the ring goes back as it came
    } else if (sane) {
        using std::integral_constant;
        using std::true_type;
        using std::false_type;
        constexpr uint32_t P = OSC_PHASE_TAME | OSC_VAL_FOLDED | OSC_CV_SERIES9;
        // (the fast copy for the class config 4's draw takes — index 0.5 ... 1.5: (2^(cv/4))^4 —; the general one for every class)
        if (car_class == 2)
            run(integral_constant<uint32_t, fo_carrier | P | OSC_CV_SMALL>{}, false_type{});
        else if (car_class == 1 && fast)
            run(integral_constant<uint32_t, fo_carrier | P | OSC_CV_QUAD>{}, true_type{});
        else if (car_class == 1)
            run(integral_constant<uint32_t, fo_carrier | P | OSC_CV_QUAD>{}, false_type{});
        else
            run(integral_constant<uint32_t, fo_carrier | P>{}, false_type{});
    } else if (s == 0) {  // the recurrence, sample by sample, one lane per voice
        OscRegs sm, sc;
        sm.pos = pos_m;
        sc.pos = pos_c;
        sm.sync_last = sc.sync_last = false;
        float sq = 0.0f, sw = 0.0f;
        for (uint32_t t = 0; t < a.T; t++) {
            const uint32_t at = p0 * (uint32_t)kBlkVoices + (uint32_t)g;
            float sine_m = 0.0f, out = 0.0f;
            osc_step(fo_mod, sm, km, ring_l[at] * c_fb, 0.0f, sine_m, sq, sw);
            ring_l[at] = sine_m;
            osc_step(fo_carrier, sc, kc, sine_m * c_ix, 0.0f, out, sq, sw);
            if (frames && active) frame_base[(size_t)t * V + (uint32_t)g] = out;
            if (mix) {
                float v = active ? out : 0.0f;
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m);
                if (g == 0) mp[t] = v;
            }
            p0 = p0 + 1u == B ? 0u : p0 + 1u;
        }
        pos_m = sm.pos;
        pos_c = sc.pos;
    }
    __syncthreads();  // the last chunk's ring writes
    if (active) {
        for (uint32_t p = (uint32_t)s; p < B; p += (uint32_t)kBlkSlices) ring[(size_t)p * V + voice] = ring_l[p * kBlkVoices + (uint32_t)g];
        if (s == 0) {
            auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
            put(om.state_row + OSC_S_POS_LO, f64_lo(pos_m));   // (slice 0 = the scan lanes)
            put(om.state_row + OSC_S_POS_HI, f64_hi(pos_m));
            put(om.state_row + OSC_S_SYNC_LAST, 0u);
            put(ocr.state_row + OSC_S_POS_LO, f64_lo(pos_c));
            put(ocr.state_row + OSC_S_POS_HI, f64_hi(pos_c));
            put(ocr.state_row + OSC_S_SYNC_LAST, 0u);
        }
    }
}

// ---- the z^-1 FM pair with the MODULATOR EXACT (config 4 as default mode renders it; round 6) -----------------------------------------------------
// buffer_size 1: the fed-back sine is last sample's, in a register, and nothing can be evaluated across time — the modulator of sample t + 1 needs
// the sine of sample t.  Until round 6 this program was the general path's (a kernel specialised at run time: 18.5 ms per step, 16.4 with the
// cheaper sine decision).  Written by hand, on TWO waves per 64 voices like render_fm_pair_split: wave 0 of the workgroup runs the modulators (the
// recurrence), wave 1 their carriers, frames and mix; the sines cross through a double-buffered LDS tile, one barrier per 32 samples.  65 536
// voices are then 2048 waves — two per SIMD, each with half the work — and one wave's instructions fill the gaps the other's dependent chains
// leave (a lone wave per SIMD issued 54 % of the time: 14.1 ms per step on one wave against this kernel's two).  Per modulator sample two chains
// that do not wait for each other — the increment from the fed-back sine (the libm's 2^e, the correctly rounded quotient), the sine from the
// phase —, the exact forms' constants in registers, the libm's table in LDS, ONE test per sample for "some lane could not decide", and what a wave
// can prove once per launch (|gain| + |val| <= 800, a fed-back value of at most 1: no test of the quotient's range, a phase that stays in [0, 1)).
// The modulator's arithmetic is osc_step's exact flavour operation for operation (x_exp2_libm_plain, div_rn_proved / div_rn_plain,
// x_sine_exact_plain, v_fract / fmod1): bit-identical to it and to the CPU tick; the carrier's is render_fm_pair's default loop.
// Reference: oscillator.rs:43-48,124-153, math.rs:152.
// (a tile with an undecided sample, again through osc_step itself — out of line: inlined, its cold arms' calls sat in the kernel's hot
// function, and their register conventions with them)
__device__ __attribute__((noinline)) void fm_x_tile_again(double pos0, float fed0, double sr, double val, float gain, float* dst, double& pos_out, float& fed_out)
{
    using namespace dev;
    OscRegs s;
    s.pos = pos0;
    s.sync_last = false;
    OscConst k;
    k.sr = sr;
    k.val = val;
    k.delta = 0.0;
    k.inv_dt = 0.0f;
    float fed = fed0, sq = 0.0f, sw = 0.0f;
    for (int i = 0; i < kMixRows; i++) {
        float sine = 0.0f;
        osc_step(OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE | OSC_EXACT, s, k, fed * gain, 0.0f, sine, sq, sw);
        fed = sine;
        dst[i * 64] = sine;
    }
    pos_out = s.pos;
    fed_out = fed;
}

template <int kOut>
__global__ __launch_bounds__(128) void render_fm_pair_x(KernelArgs a, ChainRoles r)
{
    using namespace dev;
    using std::integral_constant;
    __shared__ __attribute__((aligned(16))) float mix_tile[kMixTile];
    __shared__ float sines[2][kMixRows * 64];
    __shared__ uint64_t tab[256];
    const int lane = (int)(threadIdx.x & 63u);
    const bool carrier = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) != 0;
    const WaveMap wm = wave_map(a, lane);
    const uint32_t voice = wm.voice, vc = wm.vc, V = a.V;
    const bool active = wm.active;
    auto row = [&](int rr) { return a.table[(size_t)rr * V + vc]; };
    auto parv = [&](const DevOp& op, int k) { return op.par_row[k] >= 0 ? __uint_as_float(row(op.par_row[k])) : op.par_val[k]; };
    auto put = [&](int rr, uint32_t v) { a.table[(size_t)rr * V + voice] = v; };
    const DevOp& oscop = a.ops[carrier ? r.osc_a : r.osc_l];   // this wave's oscillator (roles as in render_fm_pair)
    const DevOp& mulop = a.ops[carrier ? r.vca : r.adsr];      // the Multiply in front of its CV: x index / x feedback gain
    const int plane = a.ops[r.out].aux;
    const int ring_row = r.track;

    for (int k = (int)threadIdx.x; k < 256; k += 128) tab[k] = kLibmExpTab[k];
    constexpr uint32_t fo = OSC_HAS_CV | OSC_CV_AUDIO_RATE | OSC_AA | OSC_OUT_SINE;
    constexpr uint32_t fo_mod = fo | OSC_EXACT, fo_carrier = fo | OSC_SINE_LOOSE;
    OscRegs s;
    OscConst k;
    s.pos = make_f64(row(oscop.state_row + OSC_S_POS_LO), row(oscop.state_row + OSC_S_POS_HI));
    s.sync_last = false;
    k.sr = oscop.sample_rate;
    k.val = (double)parv(oscop, OSC_P_VAL);
    k.delta = 0.0;
    k.inv_dt = 0.0f;
    k.scale = (440.0 / k.sr) * exp2(k.val);  // (the carrier's proved loops)
    const float gain = parv(mulop, MATH_P_CONST);
    float fed = __uint_as_float(row(ring_row));  // (modulator wave) OSC_M.sine of the previous tick
    Emit em = make_emit(a, plane, lane);         // (carrier wave)
    float sq = 0.0f, sw = 0.0f;
    XConsts X;
    x_consts_load(X);
    const LibmTabLds libm{tab};
    const double inv_sr = 1.0 / k.sr;
    OscFacts facts = fm_osc_facts(gain, k, s.pos);  // (carrier wave: its class; re-proved once a tile has left it)
    int car_class = fm_gain_class(gain);
    // (modulator wave) what the exact forms may skip once it is proved for the launch — see render_fm_pair_block_x
    const bool mod_ok = (double)__builtin_fabsf(gain) + __builtin_fabs(k.val) <= 800.0 && k.sr >= 1.0 && k.sr <= 65535.0 && s.pos >= 0.0 && s.pos < 1.0 && __builtin_fabsf(fed) <= 1.0f;
    const bool mod_proved = __builtin_amdgcn_ballot_w64(!mod_ok) == 0;
    __syncthreads();  // the table

    const uint32_t n_tiles = (a.T + (uint32_t)kMixRows - 1u) / (uint32_t)kMixRows;
    for (uint32_t kt = 0; kt <= n_tiles; kt++) {
        if (!carrier) {
            if (kt < n_tiles) {
                const uint32_t t0 = kt * (uint32_t)kMixRows;
                const int n = (int)min((uint32_t)kMixRows, a.T - t0);
                float* const dst = sines[kt & 1u] + lane;
                if (mod_proved && n == kMixRows) {
                    // A tile of 32 samples SPECULATIVELY: osc_step's exact flavour (merged form) with its tests proved away and its one remaining
                    // question — "could some lane not decide?" (a sine within 1e-13 of an f32 rounding boundary, an exponent below pow's plain range:
                    // 3.4e-6 of the samples) — asked ONCE, after the tile.  Asked per sample it is a wave-uniform branch between every two samples of a
                    // recurrence that runs at one or two waves per SIMD: 14.5 ms per step against 9.2 without the question (tools/gpu_r6.sh ab).  A tile
                    // with an undecided sample (0.7 % of them) is rendered again from its first sample through osc_step itself — every value the
                    // reference's, the decided ones the same bits as before.
                    const double pos0 = s.pos;
                    const float fed0 = fed;
                    bool cold = false;
#pragma unroll SRK_FM_UNROLL
                    for (int i = 0; i < kMixRows; i++) {
                        const double e = (double)(fed * gain) + k.val;
                        const double delta = div_rn_proved(X.k440 * x_exp2_libm_plain(X, e, cold, libm), k.sr, inv_sr);
                        const float sn = x_sine_exact_plain<false>(X, s.pos, cold);
                        s.pos = __builtin_amdgcn_fract(s.pos + delta);
                        fed = sn;
                        dst[i * 64] = sn;
                    }
                    if (__builtin_expect(__builtin_amdgcn_ballot_w64(cold) != 0, 0)) fm_x_tile_again(pos0, fed0, k.sr, k.val, gain, dst, s.pos, fed);
                } else {  // a ragged last tile, a modulator that is not proved: osc_step itself (the same values)
                    for (int i = 0; i < n; i++) {
                        float sine = 0.0f;
                        osc_step(fo_mod, s, k, fed * gain, 0.0f, sine, sq, sw);
                        fed = sine;
                        dst[i * 64] = sine;
                    }
                }
            }
        } else if (kt > 0) {
            const uint32_t t0 = (kt - 1u) * (uint32_t)kMixRows;
            const int n = (int)min((uint32_t)kMixRows, a.T - t0);
            const float* const src = sines[(kt - 1u) & 1u] + lane;
            float in[kMixRows];
#pragma unroll
            for (int i = 0; i < kMixRows; i++) in[i] = src[i * 64];  // (rows past a short last tile: stale, unused)
            bool proved = false;
            if (facts.tame) {  // a sine from a phase outside [0, 1), or a NaN, is not bounded by 1: look (a ragged last tile takes the same arithmetic as a whole one)
                float m = 0.0f;
#pragma unroll
                for (int i = 0; i < kMixRows; i++) m = __builtin_fmaxf(m, i < n ? __builtin_fabsf(in[i]) : 0.0f);
                proved = __builtin_amdgcn_ballot_w64(!(m <= 1.0f)) == 0;
            }
            auto tile = [&](auto flags_c) {
                constexpr uint32_t F = decltype(flags_c)::value;
                if (n == kMixRows) {
#pragma unroll SRK_FM_UNROLL
                    for (int i = 0; i < kMixRows; i++) {
                        float out = 0.0f;
                        osc_step(F, s, k, in[i] * gain, 0.0f, out, sq, sw);
                        emit_put<kOut>(em, mix_tile, out, i, V);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < kMixRows; i++)
                        if (i < n) {
                            float out = 0.0f;
                            osc_step(F, s, k, in[i] * gain, 0.0f, out, sq, sw);
                            emit_put<kOut>(em, mix_tile, out, i, V);
                        }
                }
            };
            constexpr uint32_t P = OSC_PHASE_TAME | OSC_VAL_FOLDED | OSC_CV_SERIES9;
            if (proved && car_class == 2)
                tile(integral_constant<uint32_t, fo_carrier | P | OSC_CV_SMALL>{});
            else if (proved && car_class == 1)
                tile(integral_constant<uint32_t, fo_carrier | P | OSC_CV_QUAD>{});
            else if (proved)
                tile(integral_constant<uint32_t, fo_carrier | P>{});
            else
                tile(integral_constant<uint32_t, fo_carrier>{});
            if (!proved) {  // the literal forms may have left [0, 1)
                facts = fm_osc_facts(gain, k, s.pos);
                car_class = fm_gain_class(gain);
            }
            emit_flush<kOut, false>(em, mix_tile, t0, n, V);
        }
        __syncthreads();  // tile kt is complete and visible; the carrier is done with the buffer tile kt + 1 will overwrite
    }
    if (active) {
        put(oscop.state_row + OSC_S_POS_LO, f64_lo(s.pos));
        put(oscop.state_row + OSC_S_POS_HI, f64_hi(s.pos));
        put(oscop.state_row + OSC_S_SYNC_LAST, 0u);
        if (!carrier) put(ring_row, __float_as_uint(fed));
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
    float* mix;             // [channels][mix_stride], T samples of it
    uint32_t T, n_waves, n_channels, n_planes;
    uint32_t mix_stride;    // samples between channels of `mix` (the whole render's length; T is one segment of it)
    uint32_t t_begin, t_end;  // mix_reduce_groups: the samples of the segment this launch sums (a chunk)
    int32_t channel_plane[8];
};

__global__ __launch_bounds__(256) void mix_reduce_groups(MixArgs m)
{
    const uint32_t i = m.t_begin + blockIdx.x * 256u + threadIdx.x;  // samples [t_begin, t_end) of the segment: one chunk's, or all
    if (i >= m.t_end) return;
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
    if (i >= m.t_end) return;  // (m.T is the row pitch of the partials, not the segment's length)
    for (uint32_t c = 0; c < m.n_channels; c++) {
        const int plane = m.channel_plane[c];
        float s = 0.0f;
        if (plane >= 0)
            for (uint32_t y = 0; y < kMixSplit; y++) s += m.mixgroup[((size_t)plane * kMixSplit + y) * m.T + i];
        m.mix[(size_t)c * m.mix_stride + i] = s;
    }
}

__global__ void fill_zero(float* p, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0.0f;
}

}  // namespace srack
