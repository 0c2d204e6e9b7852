/*
 * srack_oracle.c — CPU restatement of s-rack's tick loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (s-rack_amd/) never links, imports or calls it.
 *
 * The reference (s-rack v0.3.1) is Rust and cannot be built in this image (no rustc/cargo,
 * 478 crates.io dependencies, no network), so this is a restatement, not the reference binary.
 * PARITY PINNING: the reference's own tests pin only (1) the oscillator's sine port at
 * sr=1760/B=17 (src/synth/oscillator.rs:284-305) and (2) planner ordering incl. one 2-cycle
 * (src/synth.rs:537-613); both are replayed against this file in tests/test_oracle.py.  For
 * saw/square/PolyBLEP, the ladder filter, ADSR, VCA, mixer, math and output the reference holds
 * no known-answer vector, so for those "parity unpinned" by the reference: fidelity rests on
 * this file and the independent NumPy restatement (oracle/srack_numpy.py) agreeing bit for bit.
 * The same holds for the two sequencers (src/synth/sequencer.rs:190-246, 482-533), the non-linear
 * waveshaper (src/synth/math.rs:176-205, 291-311) and the sample player (src/synth/sample.rs:192-240),
 * added as the scope table's "next" rows.
 *
 * Structure follows the reference, not the GPU path: one object graph, per-port block buffers of
 * `buffer_size` f32 (zero-initialised, synth.rs:31-33), module-major execute() (synth.rs:97-101),
 * inputs resolved by reading the source module's buffer at calc() time (synth.rs:249-254) — so a
 * feedback edge the planner broke is a buffer_size-sample delay with no extra code, exactly as in
 * the reference.
 *
 * Rust -> C mapping that keeps bits equal on x86_64-linux-gnu: f64::powf -> pow,
 * f64::sin -> sin, f64 % -> fmod, f32::powi(3) -> x*x*x, f32::min/max -> fminf/fmaxf,
 * `as f32` -> (float).  Build with -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/srack_hip.h" /* enum vocabulary only (module types, fields, ports) */

#define OR_MAX_IN 8
#define OR_MAX_OUT 9  /* PatternSequencerModule: 8 gates + sync */

typedef struct {
    int src; /* module index, -1 = None */
    int port;
} or_input;

/* synth.rs:276-298 */
typedef struct {
    int last;
} or_transition_detector;

static int or_is_transition(or_transition_detector* d, float val)
{
    int above = val > 0.0f;              /* synth.rs:286-288 */
    int is_transition = above && !d->last; /* synth.rs:294 */
    d->last = above;
    return is_transition;
}

typedef struct {
    /* oscillator.rs:10-24 */
    float val;
    uint16_t sample_rate;
    double pos;
    int antialiasing;
    or_transition_detector sync_detector;
} or_osc;

typedef struct {
    /* filter.rs:21-24, 49-56 */
    float freq, res, exp_amt;
    float f, p, q, b[5], sfreq, sres;
} or_vcf;

typedef struct {
    /* adsr.rs:8-24 */
    float a_sec, d_sec, s_val, r_sec, phase;
    int mode;
    float r_val, from_a_val, sample_rate;
    or_transition_detector td;
} or_adsr;

/* sequencer.rs:13-30 (grid) and 337-349 (pattern); grid uses channel 0 of `present/hold` plus `val` */
typedef struct {
    int length;            /* sequence.len() / sequence[0].len() */
    int steps_per_octave, octaves;
    uint16_t val[64];      /* grid: Some((val, _)) */
    uint8_t present[8][64];/* != None */
    uint8_t hold[8][64];   /* grid: Some((_, hold)); pattern: Some(true) */
    uint16_t current_step;
    or_transition_detector td, sync_td;
    float last;
} or_seq;

/* sample.rs:72-85 + WaveBox (sample.rs:15-20).  The reference shares one WaveBox between clones of a
 * module (Arc); here every clone owns a copy, including the `new` flag, so each voice sees the load. */
typedef struct {
    or_transition_detector td;
    float pos;
    int playing;
    float sample_rate;   /* audio_config.sample_rate as f32 */
    float* samples;      /* wavebox.samples */
    uint32_t n_samples;
    float wave_sample_rate;
    int wave_new;
} or_sample;

/* ---- freeverb crate 0.1.0 (Cargo.lock:1479-1482; registry checksum 68732b37...8343e), NOT vendored under /root/reference -------
 * PARITY UNPINNED: restated from the crate's published algorithm — Jezar's public-domain Freeverb as ported to Rust by
 * Ian Hobson (github.com/irh/freeverb-rs, src/freeverb.rs, comb.rs, all_pass.rs, delay_line.rs) — not from a source file
 * in this tree; no reference test or golden vector exercises the module, so nothing here is checked against the real crate.
 * What the restatement fixes: eight parallel combs and four series allpasses per channel; tunings 1116 1188 1277 1356 1422
 * 1491 1557 1617 / 556 441 341 225 (+23 for the right channel), each scaled as `(len as f64 * sr as f64 / 44100.0) as usize`;
 * FIXED_GAIN 0.015, SCALE_WET 3, SCALE_DAMPENING 0.4, SCALE_ROOM 0.28, OFFSET_ROOM 0.7, allpass feedback 0.5; all f64.
 * The module's own part (freeverb.rs) IS in the tree and is followed line by line (or_calc_freeverb). */
#define OR_FV_COMBS 8
#define OR_FV_ALLPASSES 4
typedef struct {
    double* buffer;
    uint32_t len, index;
} or_fv_delay; /* DelayLine: read() = buffer[index]; write_and_advance(v): buffer[index] = v, index = index + 1 wrapped */

typedef struct {
    or_fv_delay delay;
    double feedback, filter_state, dampening, dampening_inverse;
} or_fv_comb;

typedef struct {
    int initialised;      /* freeverb: Option<Freeverb> is Some (freeverb.rs:17, 209-214) */
    or_fv_comb comb[OR_FV_COMBS][2];
    or_fv_delay allpass[OR_FV_ALLPASSES][2];
    double wet_gain0, wet_gain1, wet, width, dry, input_gain, dampening, room_size;
    int frozen;
    /* the module's members, freeverb.rs:18-30 */
    uint32_t sample_rate;
    double p_dampening, p_dampening_ctl, p_wet, p_wet_ctl, p_width, p_width_ctl, p_room_size, p_room_size_ctl, p_dry, p_dry_ctl;
    int p_freeze, p_freeze_ctl;
} or_freeverb;

struct or_patch;
struct or_module;
/* A test's replacement for ONE module's calc() (or_set_calc_hook).  NULL — in every user of libsrack_oracle.so — is the reference's calc().
 * tests/cpp/forms_emu.c, which compiles this file into a library of its own, uses it to put the GPU default mode's cheaper forms (the f32
 * PolyBLEP, the fma-contracted ladder ...) into single modules of an otherwise untouched tick: a CPU check of csrc/approx.cpp's error bound. */
typedef void (*or_calc_hook)(struct or_patch* p, struct or_module* m);

typedef struct or_module {
    int type;
    int n_in, n_out;
    or_input in[OR_MAX_IN];
    float* out[OR_MAX_OUT]; /* n_out buffers of B floats; OutputModule keeps its `bufs` here too */
    or_calc_hook hook;      /* NULL: the reference's calc() */
    uint32_t hook_word;     /* the hook's own (copied with the module by or_patch_clone) */
    union {
        or_osc osc;
        or_vcf vcf;
        or_adsr adsr;
        struct { int negative; } vca;
        struct { float gain[4]; } mix;
        struct { float constant; int operation; } math;
        struct { float constant; } nonlin;
        struct { uint64_t n; } noise; /* samples drawn so far */
        or_freeverb fv;
        or_sample smp;
        or_seq seq;
    } u;
} or_module;

typedef struct or_patch {
    uint32_t sample_rate, buffer_size, channels;
    int n_modules, cap_modules;
    or_module* modules;
    int* plan;
    int n_plan;
    int plan_valid;
    int output; /* index of the OutputModule used by the last plan, -1 if none */
    /* removed scheduler edges recorded by the planner: (from, module) pairs */
    int* removed;
    int n_removed;
    /* NoiseModule streams (no reference counterpart: rand::random is OS-seeded): or_set_noise_seed */
    uint64_t noise_seed, noise_voice;
} or_patch;

/* ------------------------------------------------------------------------------------------ */

or_patch* or_patch_new(uint32_t sample_rate, uint32_t buffer_size, uint32_t channels)
{
    or_patch* p = (or_patch*)calloc(1, sizeof(or_patch));
    p->sample_rate = sample_rate;
    p->buffer_size = buffer_size;
    p->channels = channels;
    p->output = -1;
    return p;
}

static void or_fv_release(or_freeverb* f)
{
    if (!f->initialised) return;
    for (int k = 0; k < OR_FV_COMBS; k++)
        for (int c = 0; c < 2; c++) free(f->comb[k][c].delay.buffer);
    for (int k = 0; k < OR_FV_ALLPASSES; k++)
        for (int c = 0; c < 2; c++) free(f->allpass[k][c].buffer);
    f->initialised = 0;
}

void or_patch_free(or_patch* p)
{
    if (!p) return;
    for (int i = 0; i < p->n_modules; i++) {
        for (int k = 0; k < OR_MAX_OUT; k++) free(p->modules[i].out[k]);
        if (p->modules[i].type == SRACK_MOD_SAMPLE) free(p->modules[i].u.smp.samples);
        if (p->modules[i].type == SRACK_MOD_FREEVERB) or_fv_release(&p->modules[i].u.fv);
    }
    free(p->modules);
    free(p->plan);
    free(p->removed);
    free(p);
}

static float* or_new_buffer(const or_patch* p)
{
    return (float*)calloc(p->buffer_size, sizeof(float)); /* AudioBuffer::new, synth.rs:31-33 */
}

/* Module::new(&audio_config) with the reference defaults. */
int or_add_module(or_patch* p, int type)
{
    if (p->n_modules == p->cap_modules) {
        p->cap_modules = p->cap_modules ? p->cap_modules * 2 : 16;
        p->modules = (or_module*)realloc(p->modules, sizeof(or_module) * (size_t)p->cap_modules);
    }
    or_module* m = &p->modules[p->n_modules];
    memset(m, 0, sizeof(*m));
    m->type = type;
    for (int k = 0; k < OR_MAX_IN; k++) m->in[k].src = -1;
    switch (type) {
    case SRACK_MOD_OUTPUT: /* output.rs:15-23 */
        if (p->channels > OR_MAX_IN) return -1;
        m->n_in = (int)p->channels;
        m->n_out = 0;
        for (uint32_t c = 0; c < p->channels; c++) m->out[c] = or_new_buffer(p); /* bufs */
        break;
    case SRACK_MOD_OSCILLATOR: /* oscillator.rs:27-41 */
        m->n_in = 2;
        m->n_out = 3;
        m->u.osc.val = 0.0f;
        m->u.osc.sample_rate = (uint16_t)p->sample_rate;
        m->u.osc.pos = 0.0;
        m->u.osc.antialiasing = 1;
        m->u.osc.sync_detector.last = 1; /* synth.rs:283 */
        break;
    case SRACK_MOD_MOOG_FILTER: /* filter.rs:28-41, state Default => zeros */
        m->n_in = 2;
        m->n_out = 3;
        m->u.vcf.freq = 0.2f;
        m->u.vcf.res = 0.5f;
        m->u.vcf.exp_amt = 0.5f;
        break;
    case SRACK_MOD_ADSR: /* adsr.rs:36-53 */
        m->n_in = 1;
        m->n_out = 1;
        m->u.adsr.a_sec = 0.0f;
        m->u.adsr.d_sec = 0.5f;
        m->u.adsr.s_val = 0.25f;
        m->u.adsr.r_sec = 0.5f;
        m->u.adsr.phase = 0.0f;
        m->u.adsr.mode = SRACK_ADSR_MODE_NONE;
        m->u.adsr.r_val = 0.0f;
        m->u.adsr.from_a_val = 0.0f;
        m->u.adsr.sample_rate = (float)(uint16_t)p->sample_rate;
        m->u.adsr.td.last = 1;
        break;
    case SRACK_MOD_VCA: /* vca.rs:18-26 */
        m->n_in = 2;
        m->n_out = 1;
        m->u.vca.negative = 0;
        break;
    case SRACK_MOD_MONO_MIXER: /* mixer.rs:16-23 */
        m->n_in = 4;
        m->n_out = 1;
        for (int k = 0; k < 4; k++) m->u.mix.gain[k] = 1.0f;
        break;
    case SRACK_MOD_MATH: /* math.rs:26-35 */
        m->n_in = 2;
        m->n_out = 1;
        m->u.math.constant = 0.0f;
        m->u.math.operation = SRACK_MATH_ADD;
        break;
    case SRACK_MOD_NONLINEAR: /* math.rs:186-196 */
        m->n_in = 2;
        m->n_out = 1;
        m->u.nonlin.constant = 1.0f;
        break;
    case SRACK_MOD_FREEVERB: /* freeverb.rs:60-82 */
        m->n_in = 2;
        m->n_out = 2;
        m->u.fv.sample_rate = p->sample_rate;
        m->u.fv.p_dampening = m->u.fv.p_dampening_ctl = 0.5;
        m->u.fv.p_wet = m->u.fv.p_wet_ctl = 1.0;
        m->u.fv.p_width = m->u.fv.p_width_ctl = 0.5;
        m->u.fv.p_room_size = m->u.fv.p_room_size_ctl = 0.5;
        break;
    case SRACK_MOD_NOISE: /* oscillator.rs:314-320 */
        m->n_in = 0;
        m->n_out = 1;
        m->u.noise.n = 0;
        break;
    case SRACK_MOD_SAMPLE: /* sample.rs:88-101; WaveBox::default() => no samples, sample_rate 0.0, new false */
        m->n_in = 2;
        m->n_out = 1;
        m->u.smp.td.last = 1;
        m->u.smp.sample_rate = (float)(uint16_t)p->sample_rate;
        break;
    case SRACK_MOD_GRID_SEQUENCER: /* sequencer.rs:33-50 */
        m->n_in = 2;
        m->n_out = 3;
        m->u.seq.length = 64;
        m->u.seq.octaves = 2;
        m->u.seq.steps_per_octave = 12;
        m->u.seq.td.last = 1;
        m->u.seq.sync_td.last = 1;
        break;
    case SRACK_MOD_PATTERN_SEQUENCER: /* sequencer.rs:352-368 */
        m->n_in = 2;
        m->n_out = 9;
        m->u.seq.length = 64;
        m->u.seq.td.last = 1;
        m->u.seq.sync_td.last = 1;
        break;
    default:
        return -1;
    }
    if (type != SRACK_MOD_OUTPUT)
        for (int k = 0; k < m->n_out; k++) m->out[k] = or_new_buffer(p);
    p->plan_valid = 0;
    return p->n_modules++;
}

int or_num_modules(const or_patch* p) { return p->n_modules; }

/* SynthModule::set_input: Err(()) on a bad port index. */
int or_connect(or_patch* p, int src, int src_port, int sink, int sink_port)
{
    if (src < 0 || src >= p->n_modules || sink < 0 || sink >= p->n_modules) return -1;
    or_module* m = &p->modules[sink];
    if (sink_port < 0 || sink_port >= m->n_in) return -2;
    /* the reference stores any src_port and only fails later in get_output (panic); reject early */
    if (src_port < 0 || src_port >= p->modules[src].n_out) return -2;
    m->in[sink_port].src = src;
    m->in[sink_port].port = src_port;
    p->plan_valid = 0;
    return 0;
}

int or_disconnect(or_patch* p, int sink, int sink_port)
{
    if (sink < 0 || sink >= p->n_modules) return -1;
    or_module* m = &p->modules[sink];
    if (sink_port < 0 || sink_port >= m->n_in) return -2;
    m->in[sink_port].src = -1;
    p->plan_valid = 0;
    return 0;
}

static double* or_field_f64(or_module* m, int field)
{
    if (m->type == SRACK_MOD_OSCILLATOR && field == SRACK_OSC_POS) return &m->u.osc.pos;
    if (m->type == SRACK_MOD_FREEVERB) switch (field) { /* the sliders' members: calc() copies them into the reverb (freeverb.rs:88-114) */
        case SRACK_FREEVERB_DAMPENING: return &m->u.fv.p_dampening_ctl;
        case SRACK_FREEVERB_WET: return &m->u.fv.p_wet_ctl;
        case SRACK_FREEVERB_WIDTH: return &m->u.fv.p_width_ctl;
        case SRACK_FREEVERB_ROOM_SIZE: return &m->u.fv.p_room_size_ctl;
        case SRACK_FREEVERB_DRY: return &m->u.fv.p_dry_ctl;
        default: break;
        }
    return NULL;
}

static float* or_field_f32(or_module* m, int field)
{
    switch (m->type) {
    case SRACK_MOD_OSCILLATOR:
        if (field == SRACK_OSC_VAL) return &m->u.osc.val;
        break;
    case SRACK_MOD_MOOG_FILTER: {
        or_vcf* v = &m->u.vcf;
        switch (field) {
        case SRACK_VCF_FREQ: return &v->freq;
        case SRACK_VCF_RES: return &v->res;
        case SRACK_VCF_EXP_AMT: return &v->exp_amt;
        case SRACK_VCF_ST_F: return &v->f;
        case SRACK_VCF_ST_P: return &v->p;
        case SRACK_VCF_ST_Q: return &v->q;
        case SRACK_VCF_ST_B0: case SRACK_VCF_ST_B1: case SRACK_VCF_ST_B2: case SRACK_VCF_ST_B3: case SRACK_VCF_ST_B4:
            return &v->b[field - SRACK_VCF_ST_B0];
        case SRACK_VCF_ST_FREQ: return &v->sfreq;
        case SRACK_VCF_ST_RES: return &v->sres;
        }
        break;
    }
    case SRACK_MOD_ADSR: {
        or_adsr* a = &m->u.adsr;
        switch (field) {
        case SRACK_ADSR_A_SEC: return &a->a_sec;
        case SRACK_ADSR_D_SEC: return &a->d_sec;
        case SRACK_ADSR_S_VAL: return &a->s_val;
        case SRACK_ADSR_R_SEC: return &a->r_sec;
        case SRACK_ADSR_PHASE: return &a->phase;
        case SRACK_ADSR_R_VAL: return &a->r_val;
        case SRACK_ADSR_FROM_A_VAL: return &a->from_a_val;
        case SRACK_ADSR_SAMPLE_RATE: return &a->sample_rate;
        }
        break;
    }
    case SRACK_MOD_MONO_MIXER:
        if (field >= SRACK_MIX_GAIN0 && field <= SRACK_MIX_GAIN3) return &m->u.mix.gain[field];
        break;
    case SRACK_MOD_MATH:
        if (field == SRACK_MATH_CONSTANT) return &m->u.math.constant;
        break;
    case SRACK_MOD_GRID_SEQUENCER:
        if (field == SRACK_GRIDSEQ_LAST) return &m->u.seq.last;
        break;
    case SRACK_MOD_NONLINEAR:
        if (field == SRACK_NONLIN_CONSTANT) return &m->u.nonlin.constant;
        break;
    case SRACK_MOD_SAMPLE:
        if (field == SRACK_SAMPLE_POS) return &m->u.smp.pos;
        if (field == SRACK_SAMPLE_SAMPLE_RATE) return &m->u.smp.sample_rate;
        if (field == SRACK_SAMPLE_WAVE_SAMPLE_RATE) return &m->u.smp.wave_sample_rate;
        break;
    }
    return NULL;
}

static int* or_field_int(or_module* m, int field)
{
    switch (m->type) {
    case SRACK_MOD_OSCILLATOR:
        if (field == SRACK_OSC_ANTIALIASING) return &m->u.osc.antialiasing;
        if (field == SRACK_OSC_SYNC_LAST) return &m->u.osc.sync_detector.last;
        break;
    case SRACK_MOD_ADSR:
        if (field == SRACK_ADSR_MODE) return &m->u.adsr.mode;
        if (field == SRACK_ADSR_GATE_LAST) return &m->u.adsr.td.last;
        break;
    case SRACK_MOD_VCA:
        if (field == SRACK_VCA_NEGATIVE) return &m->u.vca.negative;
        break;
    case SRACK_MOD_MATH:
        if (field == SRACK_MATH_OPERATION) return &m->u.math.operation;
        break;
    case SRACK_MOD_GRID_SEQUENCER:
        if (field == SRACK_GRIDSEQ_STEPS_PER_OCTAVE) return &m->u.seq.steps_per_octave;
        if (field == SRACK_GRIDSEQ_OCTAVES) return &m->u.seq.octaves;
        if (field == SRACK_GRIDSEQ_LENGTH) return &m->u.seq.length;
        if (field == SRACK_GRIDSEQ_STEP_LAST) return &m->u.seq.td.last;
        if (field == SRACK_GRIDSEQ_SYNC_LAST) return &m->u.seq.sync_td.last;
        break;
    case SRACK_MOD_FREEVERB:
        if (field == SRACK_FREEVERB_FREEZE) return &m->u.fv.p_freeze_ctl;
        break;
    case SRACK_MOD_SAMPLE:
        if (field == SRACK_SAMPLE_PLAYING) return &m->u.smp.playing;
        if (field == SRACK_SAMPLE_GATE_LAST) return &m->u.smp.td.last;
        if (field == SRACK_SAMPLE_WAVE_NEW) return &m->u.smp.wave_new;
        break;
    case SRACK_MOD_PATTERN_SEQUENCER:
        if (field == SRACK_PATSEQ_LENGTH) return &m->u.seq.length;
        if (field == SRACK_PATSEQ_STEP_LAST) return &m->u.seq.td.last;
        if (field == SRACK_PATSEQ_SYNC_LAST) return &m->u.seq.sync_td.last;
        break;
    }
    return NULL;
}

static int or_is_current_step_field(const or_module* m, int field)
{
    return (m->type == SRACK_MOD_GRID_SEQUENCER && field == SRACK_GRIDSEQ_CURRENT_STEP) ||
           (m->type == SRACK_MOD_PATTERN_SEQUENCER && field == SRACK_PATSEQ_CURRENT_STEP);
}

/* a grid cell: state 0 = None, 1 = Some((value,false)) / Some(false), 2 = Some((value,true)) / Some(true) */
int or_set_step(or_patch* p, int module, int channel, int step, int state, int value)
{
    if (module < 0 || module >= p->n_modules) return -1;
    or_module* m = &p->modules[module];
    int grid = m->type == SRACK_MOD_GRID_SEQUENCER;
    if (!grid && m->type != SRACK_MOD_PATTERN_SEQUENCER) return -1;
    if (step < 0 || step >= 64 || channel < 0 || channel >= (grid ? 1 : 8) || state < 0 || state > 2) return -1;
    m->u.seq.present[channel][step] = state != 0;
    m->u.seq.hold[channel][step] = state == 2;
    if (grid) m->u.seq.val[step] = (uint16_t)value;
    return 0;
}

/* What WaveBox::load leaves behind (sample.rs:31-69): the first channel's samples as f32, the file's
 * sample rate, new = true.  (Decoding the .wav itself is the host's job and off the path.) */
int or_set_wave(or_patch* p, int module, const float* samples, uint32_t n, float wave_sample_rate)
{
    if (module < 0 || module >= p->n_modules || p->modules[module].type != SRACK_MOD_SAMPLE) return -1;
    or_sample* s = &p->modules[module].u.smp;
    free(s->samples);
    s->samples = n ? (float*)malloc(sizeof(float) * n) : NULL;
    if (n) memcpy(s->samples, samples, sizeof(float) * n);
    s->n_samples = n;
    s->wave_sample_rate = wave_sample_rate;
    s->wave_new = 1;
    return 0;
}

/* The contents of an output buffer before the first tick, as a loaded .srk leaves them (AudioBuffer is serialised
 * with its samples, synth.rs:27-28).  Only a broken feedback edge's sink observes them. */
int or_set_output_buffer(or_patch* p, int module, int port, const float* samples)
{
    if (module < 0 || module >= p->n_modules) return -1;
    or_module* m = &p->modules[module];
    if (m->type == SRACK_MOD_OUTPUT || port < 0 || port >= m->n_out) return -2;
    memcpy(m->out[port], samples, sizeof(float) * p->buffer_size);
    return 0;
}

int or_set_field(or_patch* p, int module, int field, double value)
{
    if (module < 0 || module >= p->n_modules) return -1;
    or_module* m = &p->modules[module];
    double* d = or_field_f64(m, field);
    if (d) { *d = value; return 0; }
    float* f = or_field_f32(m, field);
    if (f) { *f = (float)value; return 0; }
    int* i = or_field_int(m, field);
    if (i) { *i = (int)value; return 0; }
    if (or_is_current_step_field(m, field)) { m->u.seq.current_step = (uint16_t)value; return 0; }
    return -1;
}

int or_get_field(or_patch* p, int module, int field, double* value)
{
    if (module < 0 || module >= p->n_modules) return -1;
    or_module* m = &p->modules[module];
    double* d = or_field_f64(m, field);
    if (d) { *value = *d; return 0; }
    float* f = or_field_f32(m, field);
    if (f) { *value = (double)*f; return 0; }
    int* i = or_field_int(m, field);
    if (i) { *value = (double)*i; return 0; }
    if (or_is_current_step_field(m, field)) { *value = (double)m->u.seq.current_step; return 0; }
    return -1;
}

/* ------------------------------------------------------------------------------------------ */
/* plan_execution, synth.rs:107-212.  Modules are identified by index instead of Arc address.   */

typedef struct {
    int* v;
    int n, cap;
} ivec;

static void iv_push(ivec* a, int x)
{
    if (a->n == a->cap) {
        a->cap = a->cap ? a->cap * 2 : 8;
        a->v = (int*)realloc(a->v, sizeof(int) * (size_t)a->cap);
    }
    a->v[a->n++] = x;
}

static void iv_remove(ivec* a, int idx)
{
    memmove(a->v + idx, a->v + idx + 1, sizeof(int) * (size_t)(a->n - idx - 1));
    a->n--;
}

/* synth.rs:107-126.  Returns the module `from` that lists `module` as a dependency, or -1. */
static int or_is_loop(int module, const ivec* edges, int n_modules)
{
    ivec to_search = {0}, to_add = {0};
    char* visited = (char*)calloc((size_t)n_modules, 1);
    int result = -1;
    iv_push(&to_search, module);
    for (;;) {
        /* to_search.iter().find(|m| visited.get(m).is_none()) : first unvisited in list order */
        int current = -1;
        for (int i = 0; i < to_search.n; i++)
            if (!visited[to_search.v[i]]) { current = to_search.v[i]; break; }
        if (current < 0) break;
        visited[current] = 1;
        for (int k = 0; k < edges[current].n; k++) {
            int dependency = edges[current].v[k];
            if (dependency == module) { result = current; goto done; }
            iv_push(&to_add, dependency);
        }
        for (int k = 0; k < to_add.n; k++) iv_push(&to_search, to_add.v[k]); /* append(&mut to_add) */
        to_add.n = 0;
    }
done:
    free(to_search.v);
    free(to_add.v);
    free(visited);
    return result;
}

/* `all_modules` is an explicit list of module indices (the test shuffles it, synth.rs:567). */
int or_plan_list(or_patch* p, int output, const int* all_modules, int n_all)
{
    int n = p->n_modules;
    ivec* edges = (ivec*)calloc((size_t)n, sizeof(ivec)); /* K: sink, V: sources */
    char* visited = (char*)calloc((size_t)n, 1);
    ivec to_search = {0};
    /* phase 1: create all edges (synth.rs:134-163) */
    for (int i = 0; i < n_all; i++) iv_push(&to_search, all_modules[i]);
    iv_push(&to_search, output);
    while (to_search.n) {
        int module = to_search.v[--to_search.n];
        if (visited[module]) continue;
        visited[module] = 1;
        const or_module* m = &p->modules[module];
        for (int k = 0; k < m->n_in; k++) { /* get_inputs: input-index order, None filtered */
            if (m->in[k].src < 0) continue;
            iv_push(&to_search, m->in[k].src);
            iv_push(&edges[module], m->in[k].src);
        }
    }
    /* phase 2: remove cycles (synth.rs:164-192) */
    free(p->removed);
    p->removed = NULL;
    p->n_removed = 0;
    ivec removed = {0};
    memset(visited, 0, (size_t)n);
    for (int i = 0; i < n_all; i++) iv_push(&to_search, all_modules[i]);
    iv_push(&to_search, output);
    while (to_search.n) {
        int module = to_search.v[--to_search.n];
        if (visited[module]) continue;
        visited[module] = 1;
        for (int k = 0; k < edges[module].n; k++) iv_push(&to_search, edges[module].v[k]);
        int from;
        while ((from = or_is_loop(module, edges, n)) >= 0) {
            for (;;) { /* remove every `module` entry from edges[from] (synth.rs:179-190) */
                int idx = -1;
                for (int k = 0; k < edges[from].n; k++)
                    if (edges[from].v[k] == module) { idx = k; break; }
                if (idx < 0) break;
                iv_remove(&edges[from], idx);
            }
            iv_push(&removed, from);
            iv_push(&removed, module);
        }
    }
    /* phase 3: repeatedly take the first unvisited list entry whose dependencies are all
     * visited (synth.rs:193-211) */
    memset(visited, 0, (size_t)n);
    free(p->plan);
    p->plan = (int*)malloc(sizeof(int) * (size_t)(n_all > 0 ? n_all : 1));
    p->n_plan = 0;
    for (;;) {
        int node = -1;
        for (int i = 0; i < n_all && node < 0; i++) {
            int m = all_modules[i];
            if (visited[m]) continue;
            int ready = 1;
            for (int k = 0; k < edges[m].n; k++)
                if (!visited[edges[m].v[k]]) { ready = 0; break; }
            if (ready) node = m;
        }
        if (node < 0) break;
        visited[node] = 1;
        p->plan[p->n_plan++] = node;
    }
    p->removed = removed.v;
    p->n_removed = removed.n / 2;
    p->output = output;
    p->plan_valid = 1;
    for (int i = 0; i < n; i++) free(edges[i].v);
    free(edges);
    free(visited);
    free(to_search.v);
    return p->n_plan;
}

/* The workspace's plan(): output = first OutputModule in list order (ui.rs:63-96). */
int or_plan(or_patch* p)
{
    int output = -1;
    for (int i = 0; i < p->n_modules; i++)
        if (p->modules[i].type == SRACK_MOD_OUTPUT) { output = i; break; }
    if (output < 0) { /* ui.rs:75-79: plan cleared */
        p->n_plan = 0;
        p->output = -1;
        p->plan_valid = 1;
        return 0;
    }
    int* all = (int*)malloc(sizeof(int) * (size_t)p->n_modules);
    for (int i = 0; i < p->n_modules; i++) all[i] = i;
    int r = or_plan_list(p, output, all, p->n_modules);
    free(all);
    return r;
}

int or_get_plan(const or_patch* p, int* order, int cap)
{
    for (int i = 0; i < p->n_plan && i < cap; i++) order[i] = p->plan[i];
    return p->n_plan;
}

/* (from, module) pairs: every wire module.out -> from.in became a block delay. */
int or_get_removed_edges(const or_patch* p, int* pairs, int cap)
{
    for (int i = 0; i < p->n_removed && i < cap; i++) {
        pairs[2 * i] = p->removed[2 * i];
        pairs[2 * i + 1] = p->removed[2 * i + 1];
    }
    return p->n_removed;
}

/* ------------------------------------------------------------------------------------------ */
/* calc() bodies                                                                                */

/* resolve_input (synth.rs:249-254): the source's output buffer as it is *now*, or None. */
static const float* or_resolve(const or_patch* p, const or_module* m, int k)
{
    if (m->in[k].src < 0) return NULL;
    return p->modules[m->in[k].src].out[m->in[k].port];
}

/* oscillator.rs:50-67 */
static double or_poly_blep(double t, double dt)
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

/* oscillator.rs:108-158 */
static void or_calc_osc(or_patch* p, or_module* m)
{
    static const double PI = 3.14159265358979323846264338327950288; /* std::f64::consts::PI */
    or_osc* o = &m->u.osc;
    const float* cv = or_resolve(p, m, 0);
    const float* sync_in = or_resolve(p, m, 1);
    float* sine = m->out[0];
    float* square = m->out[1];
    float* saw = m->out[2];
    for (uint32_t i = 0; i < p->buffer_size; i++) {
        float sync_val = sync_in ? sync_in[i] : 0.0f;
        if (or_is_transition(&o->sync_detector, sync_val)) o->pos = 0.0;
        /* get_freq_in_hz, oscillator.rs:43-48 */
        double hz = cv ? 440.0 * pow(2.0, (double)cv[i] + (double)o->val) : 440.0 * pow(2.0, (double)o->val);
        double delta = hz / (double)o->sample_rate;
        sine[i] = (float)sin(o->pos * PI * 2.0);
        square[i] = (o->pos < 0.5 ? -1.0f : 1.0f) -
                    (o->antialiasing ? (float)(or_poly_blep(o->pos, delta) - or_poly_blep(fmod(o->pos + 0.5, 1.0), delta)) : 0.0f);
        saw[i] = ((float)o->pos * 2.0f - 1.0f) - (o->antialiasing ? (float)or_poly_blep(o->pos, delta) : 0.0f);
        o->pos += delta;
        o->pos = fmod(o->pos, 1.0);
    }
}

/* filter.rs:58-92 */
static void or_vcf_state_calc(or_vcf* s, float input, float frequency, float res, float* lowpass, float* highpass, float* bandpass)
{
    if (frequency != s->sfreq || res != s->sres) {
        s->sfreq = frequency;
        s->sres = res;
        s->q = 1.0f - s->sfreq;
        s->p = s->sfreq + 0.8f * s->sfreq * s->q;
        s->f = s->p * 2.0f - 1.0f;
        s->q = s->sres * (1.0f + 0.5f * s->q * (1.0f - s->q + 5.6f * s->q * s->q));
    }
    input = input - (s->q * s->b[4]);
    float t1, t2;
    t1 = s->b[1];
    s->b[1] = (input + s->b[0]) * s->p - s->b[1] * s->f;
    t2 = s->b[2];
    s->b[2] = (s->b[1] + t1) * s->p - s->b[2] * s->f;
    t1 = s->b[3];
    s->b[3] = (s->b[2] + t2) * s->p - s->b[3] * s->f;
    s->b[4] = (s->b[3] + t1) * s->p - s->b[4] * s->f;
    s->b[4] = s->b[4] - (s->b[4] * s->b[4] * s->b[4]) * 0.166667f; /* powi(3) */
    s->b[0] = input;
    for (int k = 0; k < 5; k++) s->b[k] = fmaxf(fminf(s->b[k], 1.0f), -1.0f); /* clamp_buffers */
    *lowpass = s->b[4];
    *highpass = input - s->b[4];
    *bandpass = 3.0f * (s->b[3] - s->b[4]);
}

/* filter.rs:182-221.  Port 0 lowpass, 1 bandpass, 2 highpass (filter.rs:166-172); the tuple
 * (b4, in-b4, 3(b3-b4)) is assigned to (lowpass, highpass, bandpass) (filter.rs:211). */
static void or_calc_vcf(or_patch* p, or_module* m)
{
    or_vcf* v = &m->u.vcf;
    const float* audio_in = or_resolve(p, m, 0);
    const float* cv_in = or_resolve(p, m, 1);
    float* lowpass = m->out[0];
    float* bandpass = m->out[1];
    float* highpass = m->out[2];
    for (uint32_t idx = 0; idx < p->buffer_size; idx++) {
        float audio = audio_in ? audio_in[idx] : 0.0f;
        float cv = cv_in ? cv_in[idx] : 0.0f;
        or_vcf_state_calc(v, audio, fminf(fmaxf(v->freq + cv * v->exp_amt, 0.0f), 0.9f), fminf(fmaxf(v->res, 0.0f), 1.0f),
                          &lowpass[idx], &highpass[idx], &bandpass[idx]);
    }
}

/* adsr.rs:134-217 */
static void or_calc_adsr(or_patch* p, or_module* m)
{
    or_adsr* a = &m->u.adsr;
    const float* gate = or_resolve(p, m, 0);
    float* out = m->out[0];
    for (uint32_t idx = 0; idx < p->buffer_size; idx++) {
        int is_transition = or_is_transition(&a->td, gate ? gate[idx] : 0.0f);
        switch (a->mode) {
        case SRACK_ADSR_MODE_NONE:
            if (gate && gate[idx] > 0.0f) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_ATTACK;
            }
            break;
        case SRACK_ADSR_MODE_ATTACK:
            a->phase += 1.0f / (a->sample_rate * a->a_sec);
            if (a->phase >= 1.0f) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_DECAY;
            } else if (is_transition) {
                a->phase = 0.0f;
                a->r_val = a->from_a_val;
            }
            break;
        case SRACK_ADSR_MODE_DECAY:
            a->phase += 1.0f / (a->sample_rate * a->d_sec);
            if (a->phase >= 1.0f) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_SUSTAIN;
            }
            if (is_transition) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_ATTACK;
            }
            break;
        case SRACK_ADSR_MODE_SUSTAIN:
            if (!gate || gate[idx] <= 0.0f) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_RELEASE;
            }
            if (is_transition) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_ATTACK;
            }
            break;
        case SRACK_ADSR_MODE_RELEASE:
            if (gate && gate[idx] > 0.0f) {
                a->phase = 0.0f;
                a->mode = SRACK_ADSR_MODE_ATTACK;
            }
            a->phase += 1.0f / (a->sample_rate * a->r_sec);
            if (a->phase >= 1.0f) {
                a->phase = 0.0f;
                a->r_val = 0.0f;
                a->mode = SRACK_ADSR_MODE_NONE;
            }
            break;
        }
        switch (a->mode) {
        case SRACK_ADSR_MODE_NONE: out[idx] = 0.0f; break;
        case SRACK_ADSR_MODE_ATTACK: out[idx] = a->r_val + (1.0f - a->r_val) * a->phase; break;
        case SRACK_ADSR_MODE_DECAY: out[idx] = a->s_val + (1.0f - a->s_val) * (1.0f - a->phase); break;
        case SRACK_ADSR_MODE_SUSTAIN: out[idx] = a->s_val; break;
        case SRACK_ADSR_MODE_RELEASE: out[idx] = a->s_val * (1.0f - a->phase); break;
        }
        if (a->mode != SRACK_ADSR_MODE_ATTACK)
            a->r_val = out[idx];
        else
            a->from_a_val = out[idx];
    }
}

/* vca.rs:117-148 */
static void or_calc_vca(or_patch* p, or_module* m)
{
    const float* audio = or_resolve(p, m, 0);
    const float* cv = or_resolve(p, m, 1);
    float* out = m->out[0];
    if (audio && cv) {
        for (uint32_t i = 0; i < p->buffer_size; i++) out[i] = (m->u.vca.negative || cv[i] > 0.0f) ? audio[i] * cv[i] : 0.0f;
    } else {
        for (uint32_t i = 0; i < p->buffer_size; i++) out[i] = 0.0f;
    }
}

/* mixer.rs:101-122 */
static void or_calc_mixer(or_patch* p, or_module* m)
{
    float* out = m->out[0];
    for (uint32_t i = 0; i < p->buffer_size; i++) out[i] = 0.0f;
    for (int k = 0; k < 4; k++) {
        const float* buf = or_resolve(p, m, k);
        if (!buf) continue;
        float gain = m->u.mix.gain[k];
        for (uint32_t i = 0; i < p->buffer_size; i++) out[i] += buf[i] * gain;
    }
}

/* math.rs:46-52, 139-160 */
static float or_math_op(int op, float a, float b)
{
    switch (op) {
    case SRACK_MATH_ADD: return a + b;
    case SRACK_MATH_SUBTRACT: return a - b;
    default: return a * b;
    }
}

static void or_calc_math(or_patch* p, or_module* m)
{
    const float* i1 = or_resolve(p, m, 0);
    const float* i2 = or_resolve(p, m, 1);
    float* out = m->out[0];
    for (uint32_t i = 0; i < p->buffer_size; i++)
        out[i] = or_math_op(m->u.math.operation, i1 ? i1[i] : 0.0f, i2 ? i2[i] : m->u.math.constant);
}

/* NonLinearModule::operation, math.rs:203-205: sign-preserving power (f32::powf = libm powf) */
static float or_nonlin_op(float a, float b)
{
    if (a > 0.0f) return powf(a, b);
    return -powf(-a, b);
}

/* NonLinearModule::calc, math.rs:291-311 */
static void or_calc_nonlin(or_patch* p, or_module* m)
{
    const float* i1 = or_resolve(p, m, 0);
    const float* i2 = or_resolve(p, m, 1);
    float* out = m->out[0];
    for (uint32_t i = 0; i < p->buffer_size; i++) out[i] = or_nonlin_op(i1 ? i1[i] : 0.0f, i2 ? i2[i] : m->u.nonlin.constant);
}

/* Rust's `f32 as usize`: saturating, NaN -> 0 */
static size_t or_f32_as_usize(float x)
{
    if (!(x > 0.0f)) return 0; /* negative, -0, NaN */
    if (x >= 18446744073709551616.0f) return SIZE_MAX;
    return (size_t)x;
}

/* SampleModule::calc, sample.rs:192-240 (the try_lock failure branch needs a second thread holding
 * the WaveBox; there is none in an offline render) */
static void or_calc_sample(or_patch* p, or_module* m)
{
    or_sample* s = &m->u.smp;
    const float* gate_in = or_resolve(p, m, 0);
    const float* cv_in = or_resolve(p, m, 1);
    float* out = m->out[0];
    if (s->wave_new) {
        s->pos = 0.0f;
        s->playing = 0;
        s->wave_new = 0;
    }
    for (uint32_t idx = 0; idx < p->buffer_size; idx++) {
        int trigger = or_is_transition(&s->td, gate_in ? gate_in[idx] : 0.0f);
        if (trigger) {
            s->pos = 0.0f;
            s->playing = 1;
        }
        if (or_f32_as_usize(s->pos) >= (size_t)s->n_samples) {
            s->pos = 0.0f;
            s->playing = 0;
        }
        if (s->n_samples != 0)
            out[idx] = s->samples[or_f32_as_usize(s->pos)];
        else
            out[idx] = 0.0f;
        if (s->playing) s->pos += s->wave_sample_rate / s->sample_rate * powf(2.0f, cv_in ? cv_in[idx] : 0.0f);
    }
}

/* the stepping both sequencers share: sequencer.rs:219-231 and 504-516 */
static int or_seq_advance(or_seq* q, float step_in, float sync_in)
{
    if (or_is_transition(&q->td, step_in)) q->current_step += 1;
    if (or_is_transition(&q->sync_td, sync_in)) q->current_step = 0;
    int current_step = (int)q->current_step;
    if (current_step >= q->length) {
        q->current_step = 0;
        current_step = 0;
    }
    return current_step;
}

/* GridSequencerModule::calc, sequencer.rs:190-246 */
static void or_calc_gridseq(or_patch* p, or_module* m)
{
    or_seq* q = &m->u.seq;
    const float* step_buf = or_resolve(p, m, 0);
    const float* sync_buf = or_resolve(p, m, 1);
    float *cv_out = m->out[0], *gate_out = m->out[1], *sync_out = m->out[2];
    for (uint32_t idx = 0; idx < p->buffer_size; idx++) {
        float step_in = step_buf ? step_buf[idx] : 0.0f;
        float sync_in = sync_buf ? sync_buf[idx] : 0.0f;
        int cs = or_seq_advance(q, step_in, sync_in);
        if (q->present[0][cs]) {
            cv_out[idx] = (float)q->val[cs] * (1.0f / (float)(uint16_t)q->steps_per_octave);
            gate_out[idx] = q->hold[0][cs] ? 1.0f : step_in;
        } else {
            cv_out[idx] = q->last;
            gate_out[idx] = 0.0f;
        }
        sync_out[idx] = cs == 0 ? 1.0f : 0.0f;
        q->last = cv_out[idx];
    }
}

/* PatternSequencerModule::calc, sequencer.rs:482-533 */
static void or_calc_patseq(or_patch* p, or_module* m)
{
    or_seq* q = &m->u.seq;
    const float* step_buf = or_resolve(p, m, 0);
    const float* sync_buf = or_resolve(p, m, 1);
    for (uint32_t idx = 0; idx < p->buffer_size; idx++) {
        float step_in = step_buf ? step_buf[idx] : 0.0f;
        float sync_in = sync_buf ? sync_buf[idx] : 0.0f;
        int cs = or_seq_advance(q, step_in, sync_in);
        for (int c = 0; c < 8; c++) m->out[c][idx] = q->present[c][cs] ? (q->hold[c][cs] ? 1.0f : step_in) : 0.0f;
        m->out[8][idx] = cs == 0 ? 1.0f : 0.0f;
    }
}

/* output.rs:46-60 */
static void or_calc_output(or_patch* p, or_module* m)
{
    for (int c = 0; c < m->n_in; c++) {
        const float* buf = or_resolve(p, m, c);
        if (buf)
            memcpy(m->out[c], buf, sizeof(float) * p->buffer_size);
        else
            memset(m->out[c], 0, sizeof(float) * p->buffer_size);
    }
}

/* ---- the crate, restated (see the banner at or_freeverb) ---------------------------------------------------------------- */
static void or_fv_delay_new(or_fv_delay* d, uint32_t len)
{
    d->buffer = (double*)calloc(len ? len : 1, sizeof(double));
    d->len = len;
    d->index = 0;
}
static double or_fv_read(const or_fv_delay* d) { return d->buffer[d->index]; }
static void or_fv_write_and_advance(or_fv_delay* d, double v)
{
    d->buffer[d->index] = v;
    d->index = d->index == d->len - 1 ? 0 : d->index + 1;
}
static double or_fv_comb_tick(or_fv_comb* c, double input)
{
    const double output = or_fv_read(&c->delay);
    c->filter_state = output * c->dampening_inverse + c->filter_state * c->dampening;
    or_fv_write_and_advance(&c->delay, input + c->filter_state * c->feedback);
    return output;
}
static double or_fv_allpass_tick(or_fv_delay* d, double input)
{
    const double delayed = or_fv_read(d);
    const double output = -input + delayed;
    or_fv_write_and_advance(d, input + delayed * 0.5);
    return output;
}
static void or_fv_update_combs(or_freeverb* f)
{
    const double feedback = f->frozen ? 1.0 : f->room_size, dampening = f->frozen ? 0.0 : f->dampening;
    for (int k = 0; k < OR_FV_COMBS; k++)
        for (int c = 0; c < 2; c++) {
            f->comb[k][c].feedback = feedback;
            f->comb[k][c].dampening = dampening;
            f->comb[k][c].dampening_inverse = 1.0 - dampening;
        }
}
static void or_fv_update_wet_gains(or_freeverb* f)
{
    f->wet_gain0 = f->wet * (f->width / 2.0 + 0.5);
    f->wet_gain1 = f->wet * ((1.0 - f->width) / 2.0);
}
static void or_fv_set_dampening(or_freeverb* f, double v) { f->dampening = v * 0.4; or_fv_update_combs(f); }
static void or_fv_set_freeze(or_freeverb* f, int frozen) { f->frozen = frozen; or_fv_update_combs(f); } /* (pub set_freeze leaves input_gain alone) */
static void or_fv_set_wet(or_freeverb* f, double v) { f->wet = v * 3.0; or_fv_update_wet_gains(f); }
static void or_fv_set_width(or_freeverb* f, double v) { f->width = v; or_fv_update_wet_gains(f); }
static void or_fv_set_room_size(or_freeverb* f, double v) { f->room_size = v * 0.28 + 0.7; or_fv_update_combs(f); }
static void or_fv_set_dry(or_freeverb* f, double v) { f->dry = v; }
static uint32_t or_fv_adjust_length(uint32_t length, uint32_t sr) { return (uint32_t)((double)length * (double)sr / 44100.0); }
static const uint32_t OR_FV_COMB_TUNING[OR_FV_COMBS] = {1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617};
static const uint32_t OR_FV_ALLPASS_TUNING[OR_FV_ALLPASSES] = {556, 441, 341, 225};
static void or_fv_new(or_freeverb* f, uint32_t sr) /* Freeverb::new(sr) */
{
    for (int k = 0; k < OR_FV_COMBS; k++)
        for (int c = 0; c < 2; c++) {
            or_fv_delay_new(&f->comb[k][c].delay, or_fv_adjust_length(OR_FV_COMB_TUNING[k] + (c ? 23u : 0u), sr));
            f->comb[k][c].feedback = 0.5;
            f->comb[k][c].filter_state = 0.0;
            f->comb[k][c].dampening = 0.5;
            f->comb[k][c].dampening_inverse = 0.5;
        }
    for (int k = 0; k < OR_FV_ALLPASSES; k++)
        for (int c = 0; c < 2; c++) or_fv_delay_new(&f->allpass[k][c], or_fv_adjust_length(OR_FV_ALLPASS_TUNING[k] + (c ? 23u : 0u), sr));
    f->wet_gain0 = f->wet_gain1 = f->wet = f->width = f->dry = f->input_gain = f->dampening = f->room_size = 0.0;
    f->frozen = 0;
    or_fv_set_wet(f, 1.0);
    or_fv_set_width(f, 0.5);
    or_fv_set_dampening(f, 0.5);
    or_fv_set_room_size(f, 0.5);
    f->frozen = 0; /* set_frozen(false): */
    f->input_gain = 1.0;
    or_fv_update_combs(f);
    f->initialised = 1;
}
static void or_fv_tick(or_freeverb* f, double in0, double in1, double* o0, double* o1)
{
    const double input_mixed = (in0 + in1) * 0.015 * f->input_gain;
    double out0 = 0.0, out1 = 0.0;
    for (int k = 0; k < OR_FV_COMBS; k++) {
        out0 += or_fv_comb_tick(&f->comb[k][0], input_mixed);
        out1 += or_fv_comb_tick(&f->comb[k][1], input_mixed);
    }
    for (int k = 0; k < OR_FV_ALLPASSES; k++) {
        out0 = or_fv_allpass_tick(&f->allpass[k][0], out0);
        out1 = or_fv_allpass_tick(&f->allpass[k][1], out1);
    }
    *o0 = out0 * f->wet_gain0 + out1 * f->wet_gain1 + in0 * f->dry;
    *o1 = out1 * f->wet_gain0 + out0 * f->wet_gain1 + in1 * f->dry;
}

/* ---- the module: FreeverbModule::set_freeverb + calc, freeverb.rs:88-114, 208-270 (in the tree; followed literally) ------ */
static void or_fv_set_freeverb(or_freeverb* f, int all)
{
    if (f->p_dampening_ctl != f->p_dampening || all) { f->p_dampening = f->p_dampening_ctl; or_fv_set_dampening(f, f->p_dampening); }
    if (f->p_freeze_ctl != f->p_freeze || all) { f->p_freeze = f->p_freeze_ctl; or_fv_set_freeze(f, f->p_freeze); }
    if (f->p_wet_ctl != f->p_wet || all) { f->p_wet = f->p_wet_ctl; or_fv_set_wet(f, f->p_wet); }
    if (f->p_width_ctl != f->p_width || all) { f->p_width = f->p_width_ctl; or_fv_set_width(f, f->p_width); }
    if (f->p_room_size_ctl != f->p_room_size || all) { f->p_room_size = f->p_room_size_ctl; or_fv_set_room_size(f, f->p_room_size); }
    if (f->p_dry_ctl != f->p_dry || all) { f->p_dry = f->p_dry_ctl; or_fv_set_dry(f, f->p_dry); }
}
static void or_calc_freeverb(or_patch* p, or_module* m)
{
    or_freeverb* f = &m->u.fv;
    if (!f->initialised) {
        or_fv_new(f, f->sample_rate);
        or_fv_set_freeverb(f, 1);
    } else {
        or_fv_set_freeverb(f, 0);
    }
    const float* l = or_resolve(p, m, 0);
    const float* r = or_resolve(p, m, 1);
    for (uint32_t i = 0; i < p->buffer_size; i++) { /* the four (Some / None) arms of :231-265 feed 0.0 for a missing side */
        double o0, o1;
        or_fv_tick(f, l ? (double)l[i] : 0.0, r ? (double)r[i] : 0.0, &o0, &o1);
        m->out[0][i] = (float)o0;
        m->out[1][i] = (float)o1;
    }
}

/* NoiseModule::calc, oscillator.rs:381-387: `*sample = (rand::random::<f32>() - 0.5) * 2.0`.
 * rand 0.8.5 (Cargo.toml:29): random::<f32>() = thread_rng().gen() = Standard: (next_u32() >> 8) as f32 * 2^-24, i.e. one
 * of 2^24 equally likely multiples of 2^-24 in [0, 1); thread_rng is ChaCha12 seeded from the OS, so the reference's
 * sequence differs on every run and cannot be a parity target.  What is restated is the map from 24 random bits to
 * the sample; the bits come from the documented counter-based stream of include/srack_hip.h
 * (srack_patch_set_noise_seed): sample n of a voice = output n of splitmix64 seeded with the voice's key. */
static uint64_t or_splitmix64(uint64_t x);
static void or_calc_noise(or_patch* p, or_module* m)
{
    const int module = (int)(m - p->modules);
    const uint64_t base = or_splitmix64(p->noise_seed ^ or_splitmix64((uint64_t)module));
    const uint64_t key = or_splitmix64(base ^ p->noise_voice);
    float* out = m->out[0];
    for (uint32_t i = 0; i < p->buffer_size; i++) {
        const uint64_t z = or_splitmix64(key + (m->u.noise.n + i) * 0x9E3779B97F4A7C15ull);
        const float r = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);
        out[i] = (r - 0.5f) * 2.0f;
    }
    m->u.noise.n += p->buffer_size;
}

/* seed and GLOBAL index of the voice this patch object is (or_render_batch: proto's index + v) */
void or_set_noise_seed(or_patch* p, uint64_t seed, uint64_t voice)
{
    p->noise_seed = seed;
    p->noise_voice = voice;
}

static void or_calc(or_patch* p, or_module* m)
{
    if (m->hook) { /* a test's replacement (see or_calc_hook) */
        m->hook(p, m);
        return;
    }
    switch (m->type) {
    case SRACK_MOD_OUTPUT: or_calc_output(p, m); break;
    case SRACK_MOD_OSCILLATOR: or_calc_osc(p, m); break;
    case SRACK_MOD_MOOG_FILTER: or_calc_vcf(p, m); break;
    case SRACK_MOD_ADSR: or_calc_adsr(p, m); break;
    case SRACK_MOD_VCA: or_calc_vca(p, m); break;
    case SRACK_MOD_MONO_MIXER: or_calc_mixer(p, m); break;
    case SRACK_MOD_MATH: or_calc_math(p, m); break;
    case SRACK_MOD_NONLINEAR: or_calc_nonlin(p, m); break;
    case SRACK_MOD_SAMPLE: or_calc_sample(p, m); break;
    case SRACK_MOD_NOISE: or_calc_noise(p, m); break;
    case SRACK_MOD_FREEVERB: or_calc_freeverb(p, m); break;
    case SRACK_MOD_GRID_SEQUENCER: or_calc_gridseq(p, m); break;
    case SRACK_MOD_PATTERN_SEQUENCER: or_calc_patseq(p, m); break;
    }
}

int or_set_calc_hook(or_patch* p, int module, or_calc_hook hook, uint32_t word)
{
    if (module < 0 || module >= p->n_modules) return -1;
    p->modules[module].hook = hook;
    p->modules[module].hook_word = word;
    return 0;
}

/* One module's calc() on its own (the oscillator unit test drives calc() directly). */
int or_module_calc(or_patch* p, int module)
{
    if (module < 0 || module >= p->n_modules) return -1;
    or_calc(p, &p->modules[module]);
    return 0;
}

/* synth::execute, synth.rs:97-101 */
int or_execute(or_patch* p)
{
    if (!p->plan_valid) or_plan(p);
    for (int i = 0; i < p->n_plan; i++) or_calc(p, &p->modules[p->plan[i]]);
    return 0;
}

/* Copy a module's output buffer (for OutputModule: bufs[port]). */
int or_get_output(const or_patch* p, int module, int port, float* dst)
{
    if (module < 0 || module >= p->n_modules) return -1;
    const or_module* m = &p->modules[module];
    int n = m->type == SRACK_MOD_OUTPUT ? m->n_in : m->n_out;
    if (port < 0 || port >= n) return -2;
    memcpy(dst, m->out[port], sizeof(float) * p->buffer_size);
    return 0;
}

/* The offline counterpart of the audio callback (main.rs:59-90): execute once per buffer_size
 * frames, copy OutputModule.bufs[c], keep the first n_samples frames.
 * out: [channels][n_samples].  Optional tap: (tap_module, tap_port) copied to tap_out[n_samples]. */
int or_render(or_patch* p, uint32_t n_samples, float* out, int tap_module, int tap_port, float* tap_out)
{
    if (!p->plan_valid) or_plan(p);
    uint32_t B = p->buffer_size;
    for (uint32_t base = 0; base < n_samples; base += B) {
        or_execute(p);
        uint32_t n = n_samples - base < B ? n_samples - base : B;
        if (out) {
            for (uint32_t c = 0; c < p->channels; c++) {
                if (p->output >= 0)
                    memcpy(out + (size_t)c * n_samples + base, p->modules[p->output].out[c], sizeof(float) * n);
                else
                    memset(out + (size_t)c * n_samples + base, 0, sizeof(float) * n); /* main.rs:65: no output => src_buf stays 0 */
            }
        }
        if (tap_out) memcpy(tap_out + base, p->modules[tap_module].out[tap_port], sizeof(float) * n);
    }
    return 0;
}

/* Deep copy: one module-object graph per voice for the batch baseline. */
or_patch* or_patch_clone(const or_patch* src)
{
    or_patch* p = (or_patch*)calloc(1, sizeof(or_patch));
    *p = *src;
    p->modules = (or_module*)malloc(sizeof(or_module) * (size_t)(src->cap_modules ? src->cap_modules : 1));
    memcpy(p->modules, src->modules, sizeof(or_module) * (size_t)src->n_modules);
    for (int i = 0; i < p->n_modules; i++) {
        for (int k = 0; k < OR_MAX_OUT; k++)
            if (src->modules[i].out[k]) {
                p->modules[i].out[k] = (float*)malloc(sizeof(float) * src->buffer_size);
                memcpy(p->modules[i].out[k], src->modules[i].out[k], sizeof(float) * src->buffer_size);
            }
        if (src->modules[i].type == SRACK_MOD_FREEVERB) p->modules[i].u.fv.initialised = 0; /* Clone: `freeverb: None` (freeverb.rs:41) */
        if (src->modules[i].type == SRACK_MOD_SAMPLE && src->modules[i].u.smp.samples) {
            const or_sample* ss = &src->modules[i].u.smp;
            p->modules[i].u.smp.samples = (float*)malloc(sizeof(float) * ss->n_samples);
            memcpy(p->modules[i].u.smp.samples, ss->samples, sizeof(float) * ss->n_samples);
        }
    }
    p->plan = NULL;
    p->removed = NULL;
    if (src->plan) {
        p->plan = (int*)malloc(sizeof(int) * (size_t)(src->n_plan ? src->n_plan : 1));
        memcpy(p->plan, src->plan, sizeof(int) * (size_t)src->n_plan);
    }
    if (src->removed) {
        p->removed = (int*)malloc(sizeof(int) * 2 * (size_t)(src->n_removed ? src->n_removed : 1));
        memcpy(p->removed, src->removed, sizeof(int) * 2 * (size_t)src->n_removed);
    }
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* Batch render = the CPU baseline in the reference's structure: one object graph per voice,
 * block-major execute, all three oscillator outputs computed; voices split over threads.      */

typedef struct {
    int module, field;
    const double* values; /* [n_voices] */
} or_override;

typedef struct {
    const or_patch* proto;
    uint32_t v0, v1, n_voices, n_samples;
    const or_override* ov;
    int n_ov;
    float* frames;       /* [channels][n_samples][n_voices] or NULL */
    double* mix;         /* per-thread [channels][n_samples] f64 partial, or NULL */
} or_batch_job;

static void* or_batch_worker(void* arg)
{
    or_batch_job* j = (or_batch_job*)arg;
    uint32_t C = j->proto->channels, T = j->n_samples;
    float* tmp = (float*)malloc(sizeof(float) * (size_t)C * T);
    for (uint32_t v = j->v0; v < j->v1; v++) {
        or_patch* p = or_patch_clone(j->proto);
        p->noise_voice = j->proto->noise_voice + v;
        for (int k = 0; k < j->n_ov; k++) or_set_field(p, j->ov[k].module, j->ov[k].field, j->ov[k].values[v]);
        or_render(p, T, tmp, -1, 0, NULL);
        if (j->frames)
            for (uint32_t c = 0; c < C; c++)
                for (uint32_t i = 0; i < T; i++) j->frames[((size_t)c * T + i) * j->n_voices + v] = tmp[(size_t)c * T + i];
        if (j->mix)
            for (size_t i = 0; i < (size_t)C * T; i++) j->mix[i] += (double)tmp[i];
        or_patch_free(p);
    }
    free(tmp);
    return NULL;
}

/* frames: [channels][n_samples][n_voices] f32 (may be NULL); mix: [channels][n_samples] f64 sum
 * over voices (may be NULL).  ov_*: per-voice field overrides.  Returns 0. */
int or_render_batch(const or_patch* proto, uint32_t n_voices, uint32_t n_samples, int n_ov, const int* ov_module, const int* ov_field,
                    const double* const* ov_values, float* frames, double* mix, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    if ((uint32_t)n_threads > n_voices) n_threads = (int)n_voices;
    or_override* ov = (or_override*)calloc((size_t)(n_ov ? n_ov : 1), sizeof(or_override));
    for (int k = 0; k < n_ov; k++) {
        ov[k].module = ov_module[k];
        ov[k].field = ov_field[k];
        ov[k].values = ov_values[k];
    }
    or_patch* planned = or_patch_clone(proto);
    if (!planned->plan_valid) or_plan(planned);
    size_t mixn = (size_t)proto->channels * n_samples;
    or_batch_job* jobs = (or_batch_job*)calloc((size_t)n_threads, sizeof(or_batch_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; t++) {
        jobs[t].proto = planned;
        jobs[t].v0 = (uint32_t)((uint64_t)n_voices * (uint64_t)t / (uint64_t)n_threads);
        jobs[t].v1 = (uint32_t)((uint64_t)n_voices * (uint64_t)(t + 1) / (uint64_t)n_threads);
        jobs[t].n_voices = n_voices;
        jobs[t].n_samples = n_samples;
        jobs[t].ov = ov;
        jobs[t].n_ov = n_ov;
        jobs[t].frames = frames;
        jobs[t].mix = mix ? (double*)calloc(mixn, sizeof(double)) : NULL;
        pthread_create(&th[t], NULL, or_batch_worker, &jobs[t]);
    }
    if (mix) memset(mix, 0, sizeof(double) * mixn);
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (mix) {
            for (size_t i = 0; i < mixn; i++) mix[i] += jobs[t].mix[i];
            free(jobs[t].mix);
        }
    }
    free(jobs);
    free(th);
    free(ov);
    or_patch_free(planned);
    return 0;
}

/* splitmix64-based per-voice uniform in [0,1): top 24 bits -> f32.
 * u(seed, voice, k) = (splitmix64(seed ^ (voice*2+k)) >> 40) * 2^-24 */
static uint64_t or_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

float or_voice_uniform(uint64_t seed, uint64_t voice, uint32_t k)
{
    return (float)(or_splitmix64(seed ^ (voice * 2u + k)) >> 40) * (1.0f / 16777216.0f);
}
