// program.hpp — the flattened op list: what the host planner hands to the HIP kernels.
//
// One DevOp per planned module (reference: one `calc()` call per plan entry, synth.rs:97-101).
// Wires are numbered slots (an f32 per voice per sample); a slot lives in an LDS tile in the
// tile interpreter and in a VGPR in the fused kernels.  Plain C structs: shared by host C++ and
// device code, copied to the device verbatim.
#pragma once
#ifndef __HIPCC_RTC__
#include <cstdint>
#else  // the run-time compiler (hiprtc) has no standard headers; its fixed-width types live in a namespace of its own
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
typedef unsigned long uintptr_t;
#endif

namespace srack {

enum OpKind : int32_t {
    OP_NONE = 0,
    OP_OSC = 1,        // OscillatorModule::calc      oscillator.rs:108-158
    OP_VCF = 2,        // MoogFilterModule::calc      filter.rs:182-221
    OP_ADSR = 3,       // ADSRModule::calc            adsr.rs:134-217
    OP_VCA = 4,        // VCAModule::calc             vca.rs:117-148
    OP_MIX = 5,        // MonoMixerModule::calc       mixer.rs:101-122
    OP_MATH = 6,       // MathModule::calc            math.rs:139-160
    OP_OUT = 7,        // OutputModule::calc          output.rs:46-60 (+ frame store and mix-down)
    OP_DELAY_RD = 8,   // broken feedback edge: ring[(n - B) mod B] -> slot   (sink side)
    OP_DELAY_WR = 9,   // slot -> ring[n mod B]                               (source side)
    // (10 was OP_TRACK_RD: control tracks are read in place now — an input slot >= kTrackSlot — not copied into a wire)
    OP_GRIDSEQ = 11,   // GridSequencerModule::calc    sequencer.rs:190-246
    OP_PATSEQ = 12,    // PatternSequencerModule::calc sequencer.rs:482-533
    OP_NONLIN = 13,    // NonLinearModule::calc        math.rs:291-311
    OP_SAMPLE = 14,    // SampleModule::calc           sample.rs:192-240
    OP_NOISE = 15,     // NoiseModule::calc            oscillator.rs:381-387 (the draw: srack_hip.h, srack_patch_set_noise_seed)
    OP_FREEVERB = 16,  // FreeverbModule::calc         freeverb.rs:208-270 + the freeverb crate's tick (restated, modules.hip.h)
    kOpKinds = 17      // (arrays indexed by kind are sized with this: an n_kind[16] overran when OP_FREEVERB arrived — found under ASan)
};

// per-kind flag bits -----------------------------------------------------------------------------
enum : uint32_t {
    // OP_OSC
    OSC_HAS_CV = 1u << 0,
    OSC_HAS_SYNC = 1u << 1,
    OSC_AA = 1u << 2,         // antialiasing (PolyBLEP) on
    OSC_OUT_SINE = 1u << 3,   // which ports anything reads; dead ports are not computed
    OSC_OUT_SQUARE = 1u << 4,
    OSC_OUT_SAW = 1u << 5,
    OSC_EXACT = 1u << 6,      // f64 PolyBLEP / sin / pow exactly as the reference spells them: every oscillator of an exact-flavour render, or — default
                              // flavour — the single oscillators approx.cpp finds behind an unbounded error gain (a loop through a pitch, a gate ...)
    OSC_CONST_FAST = 1u << 7, // host-proved: no CV, no sync, one live port, PolyBLEP on, every voice's delta < 0.25
                              // => the carried-phase oscillator (modules.hip.h, COsc) may be used
    OSC_CV_AUDIO_RATE = 1u << 8,  // the CV SWEEPS (flatten.cpp `sweeps`: an oscillator, a filter, noise, a sample player upstream): 2^cv by polynomial every sample.
                                  // Without it the CV HOLDS (envelopes, sequencers, arithmetic on them): the increment is recomputed when the value changes, with the
                                  // reference's own 2^cv (modules.hip.h, osc_delta_cold) — a polynomial's error would be a constant per held value, i.e. a phase drift
    OSC_FIXED_PHASE = 1u << 10,   // pos rows and delta (rows or DevOp::delta's bit pattern) hold phase * 2^64 as u64, not f64 (fused voice kernels, default mode, saw)
    OSC_CV_STEPWISE = 1u << 9,    // host-proved: the CV is a sequencer's note CV (plus constants): constant between steps
    OSC_SINE_LOOSE = 1u << 12,    // host-proved: the sine port's value cannot reach a pitch input (an oscillator's or the sample player's CV), so
                                  // nothing integrates its rounding: default mode may evaluate it in f32 after the exact f64 fold
    OSC_EXACT_BLEP = 1u << 13,    // host-derived, default mode only: the f32 PolyBLEP's 2.4e-7 times the gain from this oscillator's saw / square to some output
                                  // (through a pitch input, an event input, a cutoff CV, a loop ...: approx.cpp) does not fit the error budget — its PolyBLEP is evaluated as in exact mode (f64, true division); everything else about it
                                  // (2^cv, sine, the rest of the patch) stays in the default arithmetic
    OSC_CONST_SMALL = 1u << 11,   // host-proved, whatever the render mode: no CV, no sync, one live port, PolyBLEP on, every voice's delta < 0.25
                                  // (OSC_CONST_FAST = this and not OSC_EXACT)
    // the next two are never set by the flattener: a kernel that has PROVED them for a stretch of samples (wave-uniform tests on its
    // own inputs, see render_fm_pair) passes them as compile-time constants; default mode only
    OSC_CV_SMALL = 1u << 14,      // |f64(cv) + f64(val)| <= 1/2 (|cv| <= 1/2 with OSC_VAL_FOLDED): 2^x needs no range reduction
    OSC_PHASE_TAME = 1u << 15,    // 0 <= pos < 1 and the increment is finite and >= 0: `pos %= 1.0` is one v_fract_f64
    OSC_VAL_FOLDED = 1u << 16,    // the increment is OscConst::scale * 2^cv with scale = 440 / sr * 2^val computed once per launch (2^(cv + val) =
                                  // 2^val 2^cv: one rounding more than the sum's, 1e-16 against the polynomial's 1e-12), so the bound is on |cv| alone
    OSC_CV_QUAD = 1u << 17,       // with OSC_VAL_FOLDED, |cv| <= 2: 2^cv = (2^(cv / 4))^4 — cv / 4 is exact in f32 and needs no range reduction; two
                                  // squarings instead of v_rndne, subtract, v_cvt_i32 and v_ldexp (the relative error is four times the polynomial's)
    OSC_CV_SERIES9 = 1u << 18,    // with OSC_VAL_FOLDED: the polynomial 2^f is the degree-9 interpolant (1.9e-14; exp2_fast9) instead of the degree-8 one (1.1e-12):
                                  // the degree-8 error is a smooth function of the CV, and a sine through a gain does not average it away — 1.3e-13 relative
                                  // stays, 1.2e-8 cycles of phase per minute at config 4's carrier (profiles/r06_horizon.json).  Every proved class carries it
                                  // except inside the fast FM kernels a host asks for by name (SRACK_RENDER_KEEP_DEFAULT), whose modulators drift by more
    // OP_VCF
    VCF_HAS_AUDIO = 1u << 0,
    VCF_HAS_CV = 1u << 1,
    VCF_OUT_LP = 1u << 3,
    VCF_OUT_BP = 1u << 4,
    VCF_OUT_HP = 1u << 5,
    VCF_LITERAL = 1u << 6,    // default mode only: the contracted ladder's epsilon times the gain from its ports to some output does not fit the error budget, its
                              // cutoff jumps at audio rate, or it is near self-oscillation (approx.cpp) — the ladder runs the reference's operations one by one (no fma contraction)
    // OP_ADSR
    ADSR_HAS_GATE = 1u << 0,
    // OP_VCA
    VCA_HAS_AUDIO = 1u << 0,
    VCA_HAS_CV = 1u << 1,
    // OP_MIX: bit k = input k connected
    // OP_MATH
    MATH_HAS_IN1 = 1u << 0,
    MATH_HAS_IN2 = 1u << 1,
    MATH_OP_SHIFT = 4,        // bits 4..5 = SRACK_MATH_ADD / SUBTRACT / MULTIPLY
    // OP_NONLIN: MATH_HAS_IN1 / MATH_HAS_IN2, and
    NONLIN_EXACT = 1u << 8,   // exact render mode: ocml's f64 log2 inside the power (default: a table-driven one, modules.hip.h)
    NONLIN_LOOSE = 1u << 9,   // host-proved, default mode only: the output can reach neither a pitch input nor a threshold, so nothing integrates or
                              // amplifies its rounding — the power goes through the f32 transcendental unit (v_log_f32, v_exp_f32: 1e-7 |b log2 x|)
    // OP_SAMPLE
    SMP_HAS_GATE = 1u << 0,
    SMP_HAS_CV = 1u << 1,
    // OP_GRIDSEQ / OP_PATSEQ: bit k = output port k is read; SEQ_HAS_* in bits 16..17
    SEQ_HAS_STEP = 1u << 16,
    SEQ_HAS_SYNC = 1u << 17,
    // OP_DELAY_*
    DELAY_RING_GLOBAL = 1u << 0  // ring in HBM ([B][V] f32); otherwise B consecutive LDS rows
};

// OP_FREEVERB: 24 delay lines per module (8 combs + 4 allpasses, x 2 channels), f64, one sample per voice per slot, in
// a per-program HBM block of rows of V doubles: rows [0, 16) the combs' filter states, then the lines back to back.
// Line j = 2 * unit + channel (units 0..7 combs, 8..11 allpasses).  DevOp::aux = dword offset of the op's table in
// seqtab: [24] line lengths, [24] first rows of the lines (relative to row 16), then 7 doubles (lo, hi): comb
// feedback, comb dampening, 1 - dampening, wet_gains.0, wet_gains.1, dry, input_gain.  DevOp::delta_row = the
// block's first row in KernelArgs::fv.
constexpr int kFvLines = 24, kFvStates = 16, kFvTableDwords = 2 * kFvLines + 14;

constexpr int kMaxIn = 8;  // OutputModule: one input per channel (u8 in the reference; capped at 8 here)
constexpr int kMaxOut = 9;  // PatternSequencerModule: 8 gates + sync
constexpr int kMaxPar = 8;
constexpr int kMaxTracksRead = 16;  // control tracks one program may read

// parameter indices into DevOp::par_row / par_val, per kind
enum { OSC_P_VAL = 0 };                                            // f32 `val` (used when CV is wired)
enum { VCF_P_FREQ = 0, VCF_P_RES = 1, VCF_P_EXP = 2 };
enum { ADSR_P_A = 0, ADSR_P_D = 1, ADSR_P_S = 2, ADSR_P_R = 3, ADSR_P_SR = 4 };
enum { VCA_P_NEG = 0 };
enum { MIX_P_GAIN0 = 0 };
enum { MATH_P_CONST = 0 };
enum { GRIDSEQ_P_SPO = 0 };  // steps_per_octave as f32
enum { NONLIN_P_CONST = 0 };
enum { SMP_P_SR = 0, SMP_P_WAVE_SR = 1 };

// state rows per kind (row offsets from DevOp::state_row), all 32-bit rows of the voice table
enum { OSC_S_POS_LO = 0, OSC_S_POS_HI = 1, OSC_S_SYNC_LAST = 2, OSC_S__N = 3 };
enum { VCF_S_F = 0, VCF_S_P = 1, VCF_S_Q = 2, VCF_S_B0 = 3, VCF_S_FREQ = 8, VCF_S_RES = 9, VCF_S__N = 10 };
enum { SEQ_S_CURRENT = 0, SEQ_S_STEP_LAST = 1, SEQ_S_SYNC_LAST = 2, GRIDSEQ_S_LAST = 3, GRIDSEQ_S__N = 4, PATSEQ_S__N = 3 };
enum { SMP_S_POS = 0, SMP_S_PLAYING = 1, SMP_S_GATE_LAST = 2, SMP_S__N = 3 };
enum { ADSR_S_PHASE = 0, ADSR_S_MODE = 1, ADSR_S_R_VAL = 2, ADSR_S_FROM_A = 3, ADSR_S_GATE_LAST = 4, ADSR_S__N = 5 };

struct DevOp {
    int32_t kind;
    uint32_t flags;
    int32_t module;             // index in all_modules (diagnostics, state read-back)
    int32_t in_slot[kMaxIn];    // wire slot per input port, -1 = None (unconnected)
    int32_t out_slot[kMaxOut];  // wire slot per output port, -1 = nobody reads it
    int32_t state_row;          // first state row in the voice table (-1: stateless)
    int32_t par_row[kMaxPar];   // >= 0: per-voice row in the voice table; -1: uniform => par_val
    float par_val[kMaxPar];
    // OP_OSC without CV: delta = 440 * 2^val / sample_rate, hoisted to the host in f64
    // (bit-equal to the reference's per-sample value, oscillator.rs:43-48,132)
    int32_t delta_row;          // >= 0: two per-voice rows (lo, hi); -1: uniform => delta.  OP_FREEVERB: first row of its block in KernelArgs::fv
    int32_t aux;                // OP_OUT: plane; OP_DELAY_*: ring id / first LDS row; sequencers: dword offset of the 64 cells in seqtab; OP_SAMPLE: dword offset of the wave in seqtab
    int32_t seq_row;            // sequencers: LDS row the 64 cells are staged in (shared by the wave, indexed by step)
    int32_t seq_len;            // sequencers: sequence length (1..64); OP_SAMPLE: wave length in samples
    double delta;               // OP_NOISE: the bit pattern of the module's u64 base key
    double sample_rate;         // OP_OSC: f64(sample_rate), the divisor of oscillator.rs:132; OP_NOISE: the bit pattern of the u64 first_voice
};

struct DevProgram {
    int32_t n_ops;
    int32_t n_slots;        // wire slots (tile interpreter: LDS tiles)
    int32_t n_rows;         // rows of the voice table: state rows, then per-voice parameter rows
    int32_t n_state_rows;   // rows [0, n_state_rows) are written back at the end of a render
    int32_t n_planes;       // distinct wires feeding the OutputModule
    int32_t n_channels;
    int32_t channel_plane[8];  // per output channel: plane index or -1
    int32_t buffer_size;    // B: length of a broken edge's delay
    int32_t n_rings;        // global rings ([B][V] f32 each)
    int32_t tile;           // samples per tile the interpreter uses (<= B when rings exist)
    int32_t n_tracks;       // control tracks this program reads (each gets one LDS row per tile) ...
    int32_t track_id[kMaxTracksRead];  // ... and which rows of the track buffer they are (input slot kTrackSlot + k reads track_id[k])
};

constexpr int kMaxOps = 96;

// An input slot >= kTrackSlot is not an LDS wire but control track (slot - kTrackSlot): a wave-uniform stream in HBM
// written by the control program, read in place (stride 1 per sample, no lane offset).
constexpr int kTrackSlot = 0x4000;

}  // namespace srack
