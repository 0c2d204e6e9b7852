/*
 * srack_hip.h — C ABI of the MI355X batch-render path for s-rack's module-graph evaluator.
 *
 * The reference (s-rack v0.3.1, Rust) has no FFI; its operator API for this path is the trait
 * `SynthModule` (src/synth.rs:222-263) plus the free functions `plan_execution`
 * (src/synth.rs:128-212) and `execute` (src/synth.rs:97-101).  This header is the flat,
 * value-typed mirror of exactly that surface: what a Rust host would bind with `extern "C"`
 * in place of `execute(&plan)` (the binding is shown in INTEGRATION.md).
 *
 * Conventions
 *   - every function returns `int`: 0 = ok, < 0 = error (the reference's `Err(())` / panic);
 *     `srack_last_error()` gives a thread-local message for the last failure.
 *   - the caller owns every host/device buffer it passes in; the library owns the handle and
 *     the device-side voice state hanging off it.
 *   - a handle is not thread-safe: one render at a time per handle (the reference holds the plan
 *     `Mutex` during `execute`, src/main.rs:60).
 *   - no callbacks into the host during a render; no torch / C++ types in any signature.
 *   - module indices are positions in the workspace's `all_modules` list (src/ui.rs:54): the
 *     planner's result depends on that order (src/synth.rs:193-211), so modules must be added
 *     in list order.
 */
#ifndef SRACK_HIP_H
#define SRACK_HIP_H

#ifndef __HIPCC_RTC__ /* (device code specialised at run time sees this header through hiprtc: no libc headers there) */
#include <stddef.h>
#include <stdint.h>
#else
using __hip_internal::int32_t;
using __hip_internal::int64_t;
using __hip_internal::uint32_t;
using __hip_internal::uint64_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define SRACK_ABI_VERSION 2 /* 2: srack_dist_unique_id / init / comm_count / destroy; save_srk rejects a short buffer */

/* ---- status codes ------------------------------------------------------------------------ */
enum {
    SRACK_OK              = 0,
    SRACK_ERR_INVALID     = -1, /* bad handle / argument (reference: panic or Err(())) */
    SRACK_ERR_PORT        = -2, /* port index out of range: `Err(())` of get_input/set_input/get_output */
    SRACK_ERR_NO_OUTPUT   = -3, /* no OutputModule in the list: find_output() Err, src/ui.rs:84-96 */
    SRACK_ERR_SELF_LOOP   = -4, /* module wired to itself: RwLock self-deadlock in the reference (src/synth.rs:99,251) */
    SRACK_ERR_STATE       = -5, /* call order violated (render before voices are configured, ...) */
    SRACK_ERR_UNSUPPORTED = -6, /* module type outside the hot-path scope */
    SRACK_ERR_DEVICE      = -7, /* HIP runtime error; text in srack_last_error() */
    SRACK_ERR_NOMEM       = -8
};

/* ---- module types: the ★ rows of the scope table ------------------------------------------ */
enum {
    SRACK_MOD_OUTPUT      = 0, /* output::OutputModule        src/synth/output.rs:7-60      */
    SRACK_MOD_OSCILLATOR  = 1, /* oscillator::OscillatorModule src/synth/oscillator.rs:9-158 */
    SRACK_MOD_MOOG_FILTER = 2, /* filter::MoogFilterModule    src/synth/filter.rs:11-221    */
    SRACK_MOD_ADSR        = 3, /* adsr::ADSRModule            src/synth/adsr.rs:7-217       */
    SRACK_MOD_VCA         = 4, /* vca::VCAModule              src/synth/vca.rs:6-148        */
    SRACK_MOD_MONO_MIXER  = 5, /* mixer::MonoMixerModule      src/synth/mixer.rs:6-122      */
    SRACK_MOD_MATH        = 6, /* math::MathModule            src/synth/math.rs:13-160      */
    /* scope table (f) rank 1: the clocked note sources that sit in front of the path in real patches */
    SRACK_MOD_GRID_SEQUENCER    = 7, /* sequencer::GridSequencerModule    src/synth/sequencer.rs:12-246   */
    SRACK_MOD_PATTERN_SEQUENCER = 8, /* sequencer::PatternSequencerModule src/synth/sequencer.rs:336-533 */
    /* scope table (f) rank 4: the remaining per-voice modules */
    SRACK_MOD_NONLINEAR   = 9,  /* math::NonLinearModule       src/synth/math.rs:176-311     */
    SRACK_MOD_SAMPLE      = 10, /* sample::SampleModule        src/synth/sample.rs:72-240    */
    /* White noise: out = (rand::random::<f32>() - 0.5) * 2.0 per sample (oscillator.rs:381-387).  The reference draws from
     * rand 0.8's thread-local, OS-seeded ChaCha12: no two runs of it agree, so only the DISTRIBUTION can be matched — the
     * 2^24 equally likely values k * 2^-23 - 1 of rand's Standard f32 — and the draw itself is this library's: a
     * counter-based splitmix64 stream per (seed, module, voice), see srack_patch_set_noise_seed. */
    SRACK_MOD_NOISE       = 11, /* oscillator::NoiseModule     src/synth/oscillator.rs:308-393 */
    /* Stereo reverb: the module (freeverb.rs:8-274) routes six parameters into, and ticks, `freeverb::Freeverb` of the
     * freeverb crate 0.1.0 (Cargo.lock:1479-1482), which is NOT vendored in the reference tree.  The crate is restated
     * from its published algorithm — Jezar's Freeverb: eight parallel lowpass-feedback combs and four series allpasses per
     * channel, the classic tunings scaled by sample_rate / 44100, f64 throughout — see oracle/srack_oracle.c for the
     * restatement and its status ("parity unpinned": no reference test or vector exercises it). */
    SRACK_MOD_FREEVERB    = 12, /* freeverb::FreeverbModule    src/synth/freeverb.rs:8-274     */
    SRACK_MOD__COUNT      = 13
};

/* ---- ports (u8 in the reference) --------------------------------------------------------- */
enum { SRACK_OSC_IN_CV = 0, SRACK_OSC_IN_SYNC = 1 };                        /* oscillator.rs:164-170 */
enum { SRACK_OSC_OUT_SINE = 0, SRACK_OSC_OUT_SQUARE = 1, SRACK_OSC_OUT_SAW = 2 }; /* oscillator.rs:90-97 */
enum { SRACK_VCF_IN_AUDIO = 0, SRACK_VCF_IN_CV = 1 };                       /* filter.rs:113-119 */
enum { SRACK_VCF_OUT_LOWPASS = 0, SRACK_VCF_OUT_BANDPASS = 1, SRACK_VCF_OUT_HIGHPASS = 2 }; /* filter.rs:166-172 */
enum { SRACK_ADSR_IN_GATE = 0 };                                            /* adsr.rs:77-82 */
enum { SRACK_VCA_IN_AUDIO = 0, SRACK_VCA_IN_CV = 1 };                       /* vca.rs:50-56 */
enum { SRACK_SAMPLE_IN_GATE = 0, SRACK_SAMPLE_IN_CV = 1 };                  /* sample.rs:166-172 */
enum { SRACK_FREEVERB_IN_LEFT = 0, SRACK_FREEVERB_IN_RIGHT = 1 };           /* freeverb.rs:139-145 */
enum { SRACK_FREEVERB_OUT_LEFT = 0, SRACK_FREEVERB_OUT_RIGHT = 1 };         /* freeverb.rs:192-198 */
enum { SRACK_SEQ_IN_STEP = 0, SRACK_SEQ_IN_SYNC = 1 };                      /* sequencer.rs:253-259, 543-549 */
enum { SRACK_GRIDSEQ_OUT_CV = 0, SRACK_GRIDSEQ_OUT_GATE = 1, SRACK_GRIDSEQ_OUT_SYNC = 2 }; /* sequencer.rs:291-298 */
enum { SRACK_PATSEQ_OUT_GATE0 = 0, SRACK_PATSEQ_OUT_SYNC = 8 };             /* gates 0..7, then sync (sequencer.rs:578-586) */

/* ---- fields: the serialisable struct members of each module (params AND runtime state) ---- */
/* Values travel as double (exact for f32, f64, bool, small ints). `mode`: see SRACK_ADSR_MODE_*. */
enum { /* OscillatorModule, oscillator.rs:10-24 */
    SRACK_OSC_VAL = 0,          /* f32, 1 V/oct offset; 0 => 440 Hz */
    SRACK_OSC_ANTIALIASING = 1, /* bool */
    SRACK_OSC_POS = 2,          /* f64 phase in [0,1) (state) */
    SRACK_OSC_SYNC_LAST = 3,    /* bool, TransitionDetector.last (state; starts true, synth.rs:283) */
    SRACK_OSC__NFIELDS = 4
};
enum { /* MoogFilterModule + InternalMoogFilterState, filter.rs:12-56 */
    SRACK_VCF_FREQ = 0, SRACK_VCF_RES = 1, SRACK_VCF_EXP_AMT = 2,
    SRACK_VCF_ST_F = 3, SRACK_VCF_ST_P = 4, SRACK_VCF_ST_Q = 5,
    SRACK_VCF_ST_B0 = 6, SRACK_VCF_ST_B1 = 7, SRACK_VCF_ST_B2 = 8, SRACK_VCF_ST_B3 = 9, SRACK_VCF_ST_B4 = 10,
    SRACK_VCF_ST_FREQ = 11, SRACK_VCF_ST_RES = 12,
    SRACK_VCF__NFIELDS = 13
};
enum { /* ADSRModule, adsr.rs:8-24 */
    SRACK_ADSR_A_SEC = 0, SRACK_ADSR_D_SEC = 1, SRACK_ADSR_S_VAL = 2, SRACK_ADSR_R_SEC = 3,
    SRACK_ADSR_PHASE = 4, SRACK_ADSR_MODE = 5, SRACK_ADSR_R_VAL = 6, SRACK_ADSR_FROM_A_VAL = 7,
    SRACK_ADSR_SAMPLE_RATE = 8, /* f32 copy taken at new(); NOT refreshed by set_audio_config (adsr.rs:69-71) */
    SRACK_ADSR_GATE_LAST = 9,   /* TransitionDetector.last */
    SRACK_ADSR__NFIELDS = 10
};
enum { SRACK_ADSR_MODE_ATTACK = 0, SRACK_ADSR_MODE_DECAY = 1, SRACK_ADSR_MODE_SUSTAIN = 2,
       SRACK_ADSR_MODE_RELEASE = 3, SRACK_ADSR_MODE_NONE = 4 };             /* adsr.rs:27-33 */
enum { SRACK_VCA_NEGATIVE = 0, SRACK_VCA__NFIELDS = 1 };                    /* vca.rs:14 */
enum { SRACK_MIX_GAIN0 = 0, SRACK_MIX_GAIN1 = 1, SRACK_MIX_GAIN2 = 2, SRACK_MIX_GAIN3 = 3,
       SRACK_MIX__NFIELDS = 4 };                                            /* mixer.rs:11 */
enum { SRACK_MATH_CONSTANT = 0, SRACK_MATH_OPERATION = 1, SRACK_MATH__NFIELDS = 2 }; /* math.rs:21-22 */
enum { SRACK_MATH_ADD = 0, SRACK_MATH_SUBTRACT = 1, SRACK_MATH_MULTIPLY = 2 };       /* math.rs:7-11 */
enum { /* GridSequencerModule, sequencer.rs:13-30 (the sequence itself: srack_patch_set_step) */
    SRACK_GRIDSEQ_STEPS_PER_OCTAVE = 0, /* u16, default 12 */
    SRACK_GRIDSEQ_OCTAVES = 1,          /* u8, UI only */
    SRACK_GRIDSEQ_LENGTH = 2,           /* sequence.len(), 1..64, default 64 */
    SRACK_GRIDSEQ_CURRENT_STEP = 3,     /* u16 (state) */
    SRACK_GRIDSEQ_STEP_LAST = 4,        /* transition_detector.last (state) */
    SRACK_GRIDSEQ_SYNC_LAST = 5,        /* sync_transition_detector.last (state) */
    SRACK_GRIDSEQ_LAST = 6,             /* f32: the CV held over empty steps (state) */
    SRACK_GRIDSEQ__NFIELDS = 7
};
enum { /* PatternSequencerModule, sequencer.rs:337-349 */
    SRACK_PATSEQ_LENGTH = 0,            /* sequence[0].len(), 1..64, default 64 */
    SRACK_PATSEQ_CURRENT_STEP = 1, SRACK_PATSEQ_STEP_LAST = 2, SRACK_PATSEQ_SYNC_LAST = 3,
    SRACK_PATSEQ__NFIELDS = 4
};
enum { SRACK_NONLIN_CONSTANT = 0 /* f32 exponent used when In2 is unconnected, default 1.0 */,
       SRACK_NONLIN__NFIELDS = 1 };                                         /* math.rs:177-185 */
enum { /* SampleModule + WaveBox, sample.rs:15-20, 72-85 (the samples themselves: srack_patch_set_wave) */
    SRACK_SAMPLE_SAMPLE_RATE = 0,      /* f32 copy of the audio sample rate */
    SRACK_SAMPLE_WAVE_SAMPLE_RATE = 1, /* f32 wavebox.sample_rate; 0.0 until a wave is set */
    SRACK_SAMPLE_WAVE_NEW = 2,         /* wavebox.new: the next render starts with pos = 0, playing = false */
    SRACK_SAMPLE_POS = 3,              /* f32 read position in wave samples (state) */
    SRACK_SAMPLE_PLAYING = 4,          /* bool (state) */
    SRACK_SAMPLE_GATE_LAST = 5,        /* transition_detector.last (state) */
    SRACK_SAMPLE__NFIELDS = 6
};
enum { /* FreeverbModule, freeverb.rs:19-30: the *_ctl members (what the sliders hold; calc() copies them into the reverb, :88-114).
        * All f64 except FREEZE (bool).  Per-voice overrides are not supported for this module. */
    SRACK_FREEVERB_DAMPENING = 0,      /* default 0.5, slider 0..2 */
    SRACK_FREEVERB_FREEZE = 1,         /* default false */
    SRACK_FREEVERB_WET = 2,            /* default 1.0 */
    SRACK_FREEVERB_WIDTH = 3,          /* default 0.5 */
    SRACK_FREEVERB_ROOM_SIZE = 4,      /* default 0.5 */
    SRACK_FREEVERB_DRY = 5,            /* default 0.0 */
    SRACK_FREEVERB__NFIELDS = 6
};
/* step contents for srack_patch_set_step.
 * Grid: sequence[step] = None | Some((value, hold)) (sequencer.rs:19);  Pattern: sequence[channel][step] = None | Some(false) | Some(true). */
enum { SRACK_STEP_NONE = 0, SRACK_STEP_ON = 1 /* Some((v,false)) / Some(false): gate follows the clock */,
       SRACK_STEP_HOLD = 2 /* Some((v,true)) / Some(true): gate held at 1.0 */ };

/* ---- render flags ------------------------------------------------------------------------- */
enum {
    SRACK_RENDER_DEFAULT    = 0,
    /* oscillator PolyBLEP in f64 with true division and f64 sin/pow, as the reference spells it
     * (oscillator.rs:43-67,132-152): saw/square bit-identical to the CPU tick, at ~2x VALU cost. */
    SRACK_RENDER_EXACT_OSC  = 1u << 0,
    /* never pick one of the hand-matched fused chain kernels: the patch takes the general path — a kernel specialised
     * for the flattened program at run time (below), or the tile interpreter */
    SRACK_RENDER_NO_FUSION  = 1u << 1,
    /* do not hoist voice-invariant sub-graphs into the control track; evaluate them per lane */
    SRACK_RENDER_NO_UNIFORM_HOIST = 1u << 2,
    /* evaluate the control program as one stage instead of a pipeline of dependency depths */
    SRACK_RENDER_NO_CTL_STAGES = 1u << 3,
    /* The general path.  A patch that matches none of the fused shapes is rendered by a kernel SPECIALISED for its flattened
     * program: the op list is turned into one straight-line HIP kernel over the same per-module device functions (wires and
     * module state in registers, control tracks through scalar loads, frames through buffer stores) and compiled for gfx950
     * with hiprtc, once per program structure (parameter values are not part of it), when the program is first used
     * (srack_render_reserve / first render: ~1 s).  By default that happens for renders of 4096 voices or more; smaller
     * ones, programs with the one module the generator does not cover (FreeverbModule) and hosts without hiprtc use the tile
     * interpreter, which executes the same device functions
     * tile by tile with the wires in LDS.  Both are parity-tested against the oracle. */
    SRACK_RENDER_NO_SPECIALIZE = 1u << 4, /* always the tile interpreter */
    SRACK_RENDER_SPECIALIZE    = 1u << 5, /* specialise whatever the voice count (fails loudly if it cannot) */
    /* Default mode decides per patch which of its cheaper forms each module takes, from a first-order error bound against the 1e-5
     * contract over a ten-minute render (csrc/approx.cpp; srack_render_info: "approx[bound 2.3e-06]").  Where the graph has an unbounded
     * error gain — a loop that amplifies, a ladder near self-oscillation, a loop through a gate or a pitch (BASELINE config 4's FM
     * feedback) — the oscillators behind it are evaluated exactly as the reference spells them ("; exact osc 0,3"), and a patch whose
     * VALUES have no bound, or with an unbounded gain behind a module without an exact form of its own (the sample player's pitch, a
     * NonLinear), is rendered in the exact flavour altogether ("approx[exact: ...]").  This flag keeps the default forms where the reason
     * is an unbounded GAIN (the f32 PolyBLEP / contracted ladder are still denied module by module where the bound asks for it): faster —
     * config 4: 7 ms per second of audio against 18.5 — and inside the contract for renders of seconds, not minutes ("approx[kept
     * default: ...]").  It does NOT waive "unbounded values": there the default forms' clamps treat a NaN differently from the
     * reference's min / max and the render is wrong from the first overflow on — such a patch stays exact. */
    SRACK_RENDER_KEEP_DEFAULT  = 1u << 6
};

typedef struct srack_patch srack_patch; /* opaque: the workspace's module list + plan + device voice state */

/* ---- library ------------------------------------------------------------------------------ */
int         srack_abi_version(void);
const char* srack_last_error(void);

/* ---- patch graph (host only; no GPU needed) ----------------------------------------------- */
/* AudioConfig{sample_rate:u16, buffer_size:usize, channels:u8}, synth.rs:20-25.
 * sample_rate must fit the reference's u16 (1..65535); buffer_size >= 1 is also the length of a
 * broken feedback edge's delay; channels = number of OutputModule inputs. */
int srack_patch_create(uint32_t sample_rate, uint32_t buffer_size, uint32_t channels, srack_patch** out);
int srack_patch_destroy(srack_patch* p);

/* `Module::new(&audio_config)` with the reference defaults, appended to all_modules.
 * Returns the module index (>= 0) or an error (< 0). */
int srack_patch_add_module(srack_patch* p, int module_type);
int srack_patch_num_modules(const srack_patch* p);
int srack_patch_module_type(const srack_patch* p, int module);
int srack_module_num_inputs(const srack_patch* p, int module);  /* SynthModule::get_num_inputs  */
int srack_module_num_outputs(const srack_patch* p, int module); /* SynthModule::get_num_outputs */

/* field access = the struct members egui sliders / serde touch.  Uniform across voices.
 * NOTE on state: the voices' running state lives on the device between renders.  Any edit of the patch after a render — a module,
 * a connection, a field (srack_patch_set_field, set_step, set_wave, set_noise_seed), a per-voice field, or a different `flags` argument
 * to srack_render — makes the next render re-flatten the patch and START AGAIN from the state stored in the patch (the state fields
 * as last set; the modules' defaults otherwise), sample counter 0.  Rendering with unchanged patch and flags continues seamlessly.
 * (The reference's sliders change a parameter without touching module state: srack_patch_keep_state below carries the modules'
 * state across an edit.  By hand it is: read the state fields back with srack_voices_get_field BEFORE the edit and set them as
 * per-voice fields — both tested bit for bit in tests/test_gpu_parity.py.) */
int srack_patch_set_field(srack_patch* p, int module, int field, double value);
int srack_patch_get_field(const srack_patch* p, int module, int field, double* value);
/* keep != 0: from now on an edit between renders no longer restarts the voices.  Before the patch is re-flattened, every
 * module's state fields (phases, filter states, envelope phases and modes, detector bits, sequencer steps, sampler positions) are
 * read back from the device, per voice, and become the starting state of the re-flattened program; the delay rings of feedback
 * edges and the reverbs' delay lines move device to device into the ring / reverb of the same module; the sample counter runs on.
 * (A ring or reverb whose module moves between the per-voice program and the voice-invariant control program in the edit is
 * broadcast to every voice, respectively taken from voice 0.)  srack_voices_configure always starts afresh.
 * With keep, every module of the plan is evaluated, as the reference's execute() does — by default modules that cannot influence
 * any output are skipped and their state stays as stored — so that a module wired into the audible graph later has run all along.
 * Default: off (an edit restarts the voices, as documented above). */
int srack_patch_keep_state(srack_patch* p, int keep);

/* Sequencer grid cells (the egui grid editors write these, sequencer.rs:137-184, 437-478).
 * Grid sequencer: channel must be 0, `value` is the note index (u16); pattern sequencer: channel 0..7, value ignored. */
int srack_patch_set_step(srack_patch* p, int module, int channel, int step, int state, int value);
int srack_patch_get_step(const srack_patch* p, int module, int channel, int step, int* state, int* value);

/* SampleModule's WaveBox as WaveBox::load leaves it (sample.rs:31-69): the first channel of the file as
 * f32 samples (copied), the file's sample rate, and new = true.  Shared by every voice.
 * get_wave copies up to `cap` samples and returns the wave's length. */
int srack_patch_set_wave(srack_patch* p, int module, const float* samples, uint32_t n_samples, float sample_rate);
int srack_patch_get_wave(const srack_patch* p, int module, float* samples, uint32_t cap, float* sample_rate);

/* SynthModule::set_input / disconnect_input / get_input. */
int srack_patch_connect(srack_patch* p, int src_module, int src_port, int sink_module, int sink_port);
int srack_patch_disconnect(srack_patch* p, int sink_module, int sink_port);
int srack_patch_get_input(const srack_patch* p, int sink_module, int sink_port, int* src_module /* -1 = None */, int* src_port);

/* plan_execution (synth.rs:128-212) driven as the workspace does (ui.rs:63-82): output = first
 * OutputModule in list order.  Writes the execution order (module indices) to `order` (capacity
 * `cap`, may be NULL) and returns its length.  Re-run automatically by render when the graph changed. */
int srack_patch_plan(srack_patch* p, int* order, int cap);
/* plan_execution(output, &all_modules, &mut plan) with an explicit, possibly shuffled module list —
 * what the reference's own test drives (synth.rs:561-568).  Does not affect later renders. */
int srack_patch_plan_list(srack_patch* p, int output, const int* all_modules, int n_all, int* order, int cap);
/* The scheduler edges phase 2 of the last plan removed (synth.rs:176-191): up to `cap` pairs
 * (from, module) meaning "`from` no longer waits for `module`"; returns the count. */
int srack_patch_removed_edges(srack_patch* p, int* pairs, int cap);
/* The wires that end up as block delays: every wire whose source runs after its sink in the plan, so
 * the sink reads the previous block's buffer.  Up to `cap` quadruples
 * (src_module, src_port, sink_module, sink_port); returns the count. */
int srack_patch_delayed_edges(srack_patch* p, int* quads, int cap);

/* ---- .srk rack files (scope table (f) rank 2) ------------------------------------------------- */
/* FileFormat{modules, connections, positions} (ui.rs:578-586) in rmp-serde 1.3.0's compact MessagePack form.
 * load = SynthModuleWorkspaceImpl::deserialize (ui.rs:116-135) against the host's AudioConfig: the module list comes
 * out in REVERSE file order (ui.rs:654-660), V0 variants migrate, saved buffers survive only at the same buffer_size,
 * connections with unknown ids or bad ports are dropped.  Every SynthModuleType variant of the reference loads.
 * save = serialize (ui.rs:98-114): writes at most `cap` bytes to `buf` (may be NULL) and the full size to *n_bytes.  The state
 * members written are the ones stored in the patch; with srack_patch_keep_state, after a render, they are the RUNNING state of
 * voice 0 (the app saves the rack as it plays; a rack file is one instance). */
int srack_patch_load_srk(const void* bytes, size_t n_bytes, uint32_t sample_rate, uint32_t buffer_size, uint32_t channels, srack_patch** out);
int srack_patch_save_srk(const srack_patch* p, void* buf, size_t cap, size_t* n_bytes);
/* SynthModule::get_id: the UUID string that keys a file's connection list.  Returns its length. */
int srack_patch_module_id(const srack_patch* p, int module, char* buf, size_t cap);
/* Workspace position of a module's window (FileFormat.positions); get returns 1 if the module has one, else 0. */
int srack_patch_set_module_position(srack_patch* p, int module, float x, float y);
int srack_patch_get_module_position(const srack_patch* p, int module, float* x, float* y);
/* Contents of one output buffer before the first tick (what a loaded file carries): `n` = buffer_size samples, or 0
 * for a fresh, zeroed buffer.  Only the sink of a broken feedback edge ever observes it. */
int srack_patch_set_output_buffer(srack_patch* p, int module, int port, const float* samples, uint32_t n);
/* SynthModule::get_output (synth.rs:243) as far as a host can observe it before the first tick: the block the patch holds for this
 * port — what srack_patch_set_output_buffer or a loaded file put there.  Copies up to `cap` samples, returns how many the port holds
 * (0: a fresh, zeroed buffer).  Err(()) for a bad port is SRACK_ERR_PORT. */
int srack_patch_get_output_buffer(const srack_patch* p, int module, int port, float* dst, uint32_t cap);
/* Noise modules (SRACK_MOD_NOISE).  With sm(x) = splitmix64's output function applied to x + 0x9E3779B97F4A7C15,
 *   base(module)  = sm(seed ^ sm(module))                       module = its index in the patch
 *   key(voice)    = sm(base ^ (first_voice + voice))            voice  = its index in this handle
 *   sample n      = ((sm(key + n * 0x9E3779B97F4A7C15) >> 40) * 2^-24 - 0.5) * 2.0
 * i.e. sample n is output n of a splitmix64 generator seeded with key; n counts the samples rendered since
 * srack_voices_configure.  Independent of chunking, of the render flags and of how voices are sharded: a rank that owns
 * the global voices [r*V, (r+1)*V) passes first_voice = r*V.  Defaults: seed 0, first_voice 0. */
int srack_patch_set_noise_seed(srack_patch* p, uint64_t seed, uint64_t first_voice);

/* ---- voices: N independent instances of the patch ----------------------------------------- */
/* Fix the number of voices (lanes) and drop any earlier per-voice data and device state. */
int srack_voices_configure(srack_patch* p, uint32_t n_voices);
/* Per-voice override of one field (the only concept with no reference counterpart: the reference
 * has one instance of each parameter).  `values` has n_voices entries, host memory. */
int srack_voices_set_field_f32(srack_patch* p, int module, int field, const float* values);
int srack_voices_set_field_f64(srack_patch* p, int module, int field, const double* values);

/* ---- render (needs the GPU) ---------------------------------------------------------------- */
/* Number of distinct wires feeding the OutputModule's channels and, per channel, which plane of
 * `d_frames` it is (-1 = unconnected => silence).  P1/P2 wire both channels to one source => 1 plane. */
int srack_render_planes(srack_patch* p, int* channel_plane, int cap);

/* Offline render of `n_samples` ticks for every voice, continuing from the current voice state
 * (first call: the modules' initial state), like calling execute() ceil(n/B) times and keeping
 * the first n frames (main.rs:59-90).
 *   d_frames : device, f32 [planes][n_samples][n_voices]  (voice-minor: one wave store = 256 B)
 *              may be NULL (mix only).
 *   d_mix    : device, f32 [channels][n_samples] = sum over voices of the channel's frames
 *              (the N-voice generalisation of MonoMixerModule's gain-1 sum, mixer.rs:109-118);
 *              may be NULL.
 *   stream   : hipStream_t (NULL = default stream).  The call is asynchronous w.r.t. the host.  The library enqueues on the stream
 *              during the call and never touches it afterwards (a later call, edit or state read-back that depends on this call's
 *              work waits for an event of the library's own): the host may destroy the stream whenever it could for its own work.
 */
int srack_render(srack_patch* p, uint32_t n_samples, float* d_frames, float* d_mix,
                 uint32_t flags, void* stream);

/* Optional: do everything a later srack_render(p, <= n_samples, ..., flags) would do on first use — flatten the graph,
 * upload the programs and the voice table, size the scratch buffers (mix partials when want_mix, control tracks) — so
 * that the first render costs what every render costs.  Renders nothing and leaves the voice state untouched. */
int srack_render_reserve(srack_patch* p, uint32_t n_samples, int want_mix, uint32_t flags);

/* Diagnostics of the specialised kernel (tests, tools): the HIP source generated for the voice program of the current patch
 * under `flags` (returns its length, copies at most cap - 1 characters; SRACK_ERR_UNSUPPORTED if the program holds an op the
 * generator does not cover), and its compilation for gfx950 without rendering anything (needs hiprtc, but no GPU). */
int srack_render_kernel_source(srack_patch* p, uint32_t flags, char* buf, size_t cap);
int srack_render_kernel_compile(srack_patch* p, uint32_t flags);

/* Scratch the render needs for the mix-down partials etc. is owned by the handle; this reports it. */
/* Human-readable: the programs (ops, rows, units), "approx[...]" — the default mode's error bound or why the flavour is exact —, any
 * SRACK_* tuning variable the process carries ("knobs=[...]": such a process does not render what was tested), where the kernel came from,
 * and last "kernel=<name>".  Returns the length; copies at most cap - 1 characters (buf may be NULL to ask for the length). */
int srack_render_info(srack_patch* p, char* buf, size_t cap);

/* Average duration in ms of the dominant render kernel over the renders since the last call with
 * reset != 0, measured with HIP events on the render's own stream.  Returns the number of launches
 * in *n_launches. */
int srack_render_kernel_ms(srack_patch* p, double* avg_ms, int* n_launches, int reset);

/* Read back one field of every voice's state after a render (host buffer, n_voices doubles). */
int srack_voices_get_field(srack_patch* p, int module, int field, double* values);

/* ---- the cache of run-time specialised kernels ------------------------------------------------------------
 * A general patch (anything the hand-matched kernels do not cover) renders through a kernel compiled for its STRUCTURE with hiprtc
 * the first time that structure is seen (0.3 - 2 s).  Code objects are kept in memory (the last SRACK_KERNEL_CACHE_MAX structures,
 * default 64, least recently used first out; a module is unloaded once no patch renders with it) and ON DISK, so the next start of
 * the host — and every other rank of a multi-GPU job — loads instead of compiling (ranks starting together serialise on a file
 * lock: one compiles).  The directory: srack_kernel_cache_set_dir(path); "off" disables the disk level; NULL restores the default
 * resolution — $SRACK_KERNEL_CACHE_DIR, else `.srack_kernel_cache` next to this library if writable, else $XDG_CACHE_HOME/srack_hip
 * or ~/.cache/srack_hip.  srack_render_info names where a patch's kernel came from: jit=compiled(N ms) | disk-cache | memory-cache |
 * unavailable(reason).  No counterpart in the reference (its modules are compiled Rust, src/synth.rs:300-317). */
typedef struct srack_kernel_cache_info {
    uint64_t compiled;               /* hiprtc compilations by this process */
    uint64_t disk_hits, memory_hits;
    uint64_t modules_loaded;
    uint64_t code_evictions, module_evictions;
    uint64_t resident_code_objects, resident_modules;
    double compile_ms;               /* total hiprtc time */
    char directory[512];             /* the disk cache in use; "" = none */
} srack_kernel_cache_info;
int srack_kernel_cache_set_dir(const char* dir);
int srack_kernel_cache_stats(srack_kernel_cache_info* out);

/* ---- device helpers for hosts without a HIP binding ---------------------------------------- */
int srack_device_count(int* n);
int srack_device_set(int device);
/* The calling thread's current device — the one srack_render launches on — and its PCI bus id ("0000:05:00.0"), so that a
 * multi-rank host can check that its ranks really are on different GPUs.  Either pointer may be NULL. */
int srack_device_get(int* device, char* pci_bus_id, size_t cap);
int srack_device_alloc(void** d_ptr, size_t bytes);
int srack_device_free(void* d_ptr);
/* Waits for `stream`, then copies through pinned buffers of the library's own (the caller's memory may be pageable; per device, double-buffered,
 * large copies on several helper threads: 51 GB/s for 50 GB on the GPU box); returns when h_dst holds the bytes. */
int srack_device_to_host(void* h_dst, const void* d_src, size_t bytes, void* stream);
int srack_device_sync(void* stream);

/* ---- multi-GPU mix-down --------------------------------------------------------------------- */
/* Voices shard across ranks (one process per GPU) with no exchange during the render; the only collective is the sum of
 * the per-rank partial mixes over RCCL / xGMI.  The reference has no counterpart: it is single-threaded
 * (src/main.rs:59-63); SURVEY 8(b) item 8 / 8(e) define this surface.
 *
 *   rank 0:      srack_dist_unique_id(id)                  ncclGetUniqueId; the host carries the SRACK_DIST_ID_BYTES to
 *                                                          every rank over any side channel it has
 *   every rank:  srack_device_set(local_rank); srack_dist_init(id, n_ranks, rank, &comm)     ncclCommInitRank
 *                srack_render(...); srack_dist_reduce_mix(comm, d_mix, channels * n_samples, 0, stream)
 *                srack_dist_destroy(comm)
 *
 * `comm` is a plain RCCL `ncclComm_t`: a host that already owns one may pass it to srack_dist_reduce_mix directly.
 * srack_dist_reduce_mix is ncclReduce(sum, f32, root) on `stream`, in place on `d_mix`. */
#define SRACK_DIST_ID_BYTES 128
int srack_dist_unique_id(void* id_out /* SRACK_DIST_ID_BYTES */);
int srack_dist_init(const void* id /* SRACK_DIST_ID_BYTES */, int n_ranks, int rank, void** comm_out);
int srack_dist_comm_count(void* comm, int* n_ranks); /* ncclCommCount: the ranks the communicator really spans */
int srack_dist_destroy(void* comm);
int srack_dist_reduce_mix(void* comm, float* d_mix, size_t count, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRACK_HIP_H */
