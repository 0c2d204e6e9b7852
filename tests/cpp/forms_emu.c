/* forms_emu — the GPU default mode's cheaper FORMS on the CPU, module by module, inside the oracle's tick.  TEST INFRASTRUCTURE ONLY.
 *
 * csrc/approx.cpp decides, per module, which approximations of the reference's arithmetic a patch may take — the f32 PolyBLEP, the f32 sine,
 * the fma-contracted ladder, NonLinear's power through f32 log2 / exp2 — from a first-order error bound.  What that bound takes on trust
 * (how a cutoff moves, when a ladder is a contraction, that errors compose to first order) was found wanting five times in round 5, each time
 * by a soak on the GPU.  This file lets the same soak run on the CPU: it compiles oracle/srack_oracle.c into a library of its own and
 * replaces the calc() of single modules (or_set_calc_hook) by the oracle's calc() with the form in question, restated from
 * csrc/modules.hip.h operation for operation:
 *     oscillator  EMU_OSC_F32_BLEP    osc_step's default saw / square: poly_blep_sel in f32 on the f32 phase, square_sign_safe; with
 *                 EMU_OSC_ONE_PORT    (exactly one port is read: flatten.cpp's OSC_CONST_FAST) and a constant pitch below a quarter cycle per sample,
 *                                     no sync: the CARRIED-phase forms every kernel renders such an oscillator with — cosc_saw (the second
 *                                     window as (next phase / dt)^2), cosc_square (poly_blep_fast) — round 6: tools/emu_vs_gpu.py found the
 *                                     kernels at 1.0e-6 where this file, restating osc_step only, had 1.2e-7
 *                 EMU_OSC_FIXED       ... and that saw with its phase in 2^-64 fixed point (fosc_saw): f32(pos) and t = pos / dt from the phase's UPPER 32
 *                                     bits — which is where the 1.0e-6 came from (a 17 Hz saw: 2^-32 / dt = 6.4e-7 in t).  The fixed-point accumulator
 *                                     itself (2^-64 per step) is not restated: the upper word is taken from the oracle's f64 phase
 *                 EMU_OSC_SINE_LOOSE  sine_loose: the exact fold, a degree-9 polynomial in f32
 *                 EMU_OSC_SINE_FAST   sine_fast: the fold, a degree-13 polynomial in f64, one rounding (differs from the libm's sine by an
 *                                     f32 ulp about once in 3e5 samples)
 *     filter      EMU_VCF_CONTRACTED  vcf_polys<true> / vcf_step<true>: one product of every a * b - c * d folded into an fma
 *     NonLinear   EMU_NONLIN_LOOSE    powf_pos_loose: exp2f(b * log2f(|a|)) while |b log2 |a|| < 32
 * Everything else — phase, increment (the libm's pow), envelopes, mixers, the tick's order and its block delays — is the oracle's.  It is not
 * the GPU's result bit for bit (the kernels evaluate 2^cv by a polynomial good to 1e-12 and fuse differently), but the errors it injects are
 * the forms' own, at their own places, so whatever a patch does to them — integrates, thresholds, amplifies in a chaotic loop — it does here
 * too.  tools/cpu_soak.py renders fuzz patches twice, plain and with the forms approx.cpp chose, and holds the difference to the contract;
 * tests/test_forms_emu.py pins the emulation to the GPU's measured errors on the benchmarked patches and to round 5's five finds.
 *
 * Build: gcc -O2 -std=c11 -fPIC -ffp-contract=off -fno-fast-math -shared (tests/forms_emu.py). */
#include "../../oracle/srack_oracle.c"

enum { EMU_OSC_F32_BLEP = 1u, EMU_OSC_SINE_LOOSE = 2u, EMU_OSC_SINE_FAST = 4u, EMU_OSC_ONE_PORT = 8u, EMU_OSC_FIXED = 16u, EMU_VCF_CONTRACTED = 1u, EMU_NONLIN_LOOSE = 1u };

/* ---- oscillator (modules.hip.h: sine_fold, sine_fast, sine_loose, poly_blep_sel, square_sign_safe, osc_step) ------------------------- */
static double emu_sine_fold(double pos, uint32_t* sign)
{
    const double qn = 0.5 - pos;
    union { double d; uint64_t u; } b = {qn};
    *sign = (uint32_t)(b.u >> 32) & 0x80000000u;
    const double t = fabs(qn) - 0.25;
    return 0.25 - fabs(t);
}
static float emu_xor_sign(float x, uint32_t sign)
{
    union { float f; uint32_t u; } b = {x};
    b.u ^= sign;
    return b.f;
}
static float emu_sine_fast(double pos)
{
    uint32_t sign;
    const double x = emu_sine_fold(pos, &sign);
    const double z = x * x;
    const double a01 = fma(-41.34170223990684, z, 6.283185307179272);
    const double a23 = fma(-76.70584757807868, z, 81.60524914955879);
    const double a45 = fma(-15.081496425342264, z, 42.05813586028645);
    const double z2 = z * z;
    const double b0 = fma(a23, z2, a01);
    const double b1 = fma(3.6659216216293173, z2, a45);
    const double z4 = z2 * z2;
    const double p = fma(b1, z4, b0);
    return emu_xor_sign((float)(p * x), sign);
}
static float emu_sine_loose(double pos)
{
    uint32_t sign;
    const float x = (float)emu_sine_fold(pos, &sign);
    const float z = x * x;
    const float a01 = fmaf(-41.34168243408203f, z, 6.2831854820251465f);
    const float a23 = fmaf(-76.58116912841797f, z, 81.60247802734375f);
    const float z2 = z * z;
    const float p = fmaf(fmaf(39.75982666015625f, z2, a23), z2, a01);
    return emu_xor_sign(p * x, sign);
}
static float emu_poly_blep_sel(float t, float tm1, float inv_dt, int first, int second)
{
    const float ta = t * inv_dt;
    const float tb = tm1 * inv_dt;
    const float fa = fmaf(ta, 2.0f - ta, -1.0f);
    const float fb = fmaf(tb, tb + 2.0f, 1.0f);
    return first ? fa : (second ? fb : 0.0f);
}

static void emu_calc_osc(or_patch* p, or_module* m)
{
    static const double PI = 3.14159265358979323846264338327950288;
    const uint32_t forms = m->hook_word;
    or_osc* o = &m->u.osc;
    const float* cv = or_resolve(p, m, 0);
    const float* sync_in = or_resolve(p, m, 1);
    float* sine = m->out[0];
    float* square = m->out[1];
    float* saw = m->out[2];
    for (uint32_t i = 0; i < p->buffer_size; i++) {
        float sync_val = sync_in ? sync_in[i] : 0.0f;
        if (or_is_transition(&o->sync_detector, sync_val)) o->pos = 0.0;
        double hz = cv ? 440.0 * pow(2.0, (double)cv[i] + (double)o->val) : 440.0 * pow(2.0, (double)o->val);
        double delta = hz / (double)o->sample_rate;
        const double pos = o->pos;
        sine[i] = (forms & EMU_OSC_SINE_LOOSE) ? emu_sine_loose(pos) : (forms & EMU_OSC_SINE_FAST) ? emu_sine_fast(pos) : (float)sin(pos * PI * 2.0);
        if ((forms & EMU_OSC_F32_BLEP) && (forms & EMU_OSC_ONE_PORT) && (forms & EMU_OSC_FIXED) && !cv && !sync_in && o->antialiasing && delta < 0.25) {
            /* fosc_saw (modules.hip.h) */
            const float inv_s = (float)(1.0 / delta) * 0x1p-32f;
            const uint32_t hi = (uint32_t)floor(ldexp(pos, 32));
            double w = pos + delta;
            const int wrapped = w >= 1.0;
            w = w - floor(w);
            const uint32_t nhi = (uint32_t)floor(ldexp(w, 32));
            const float c32 = (float)hi, cn = (float)nhi;
            float s_ = fmaf(c32, 0x1p-31f, -1.0f);
            if (pos < delta) {
                float u = 1.0f - c32 * inv_s;
                u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);
                s_ = fmaf(u, u, s_);
            }
            if (wrapped) {
                const float tn = cn * inv_s;
                s_ = fmaf(-tn, tn, s_);
            }
            saw[i] = s_;
            square[i] = 0.0f;   /* (not read: one port) */
        } else if ((forms & EMU_OSC_F32_BLEP) && (forms & EMU_OSC_ONE_PORT) && !cv && !sync_in && o->antialiasing && delta < 0.25) {
            /* cosc_saw / cosc_square (modules.hip.h): functions of (pos, delta) alone — the carried terms are f32(pos) and f32(pos) * inv_dt */
            const float inv_dt = (float)(1.0 / delta);              /* inv_dt_f32: one rounding */
            const float p32 = (float)pos;
            const float ta = p32 * inv_dt;
            double w = pos + delta;
            const int wrapped = w >= 1.0;
            w = w - floor(w);
            const float tn = (float)w * inv_dt;
            const float base = fmaf(p32, 2.0f, -1.0f);
            float u = 1.0f - ta;
            u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u);          /* v_med3(1 - ta, 0, 1); ta is never a NaN here (delta > 0, finite) */
            const float s1 = fmaf(u, u, base);
            const float s2 = fmaf(-tn, tn, s1);
            saw[i] = wrapped ? s2 : s1;
            {   /* poly_blep_fast twice, square_sign_safe */
                const float tb = (float)(pos - 1.0) * inv_dt;
                const float fa = fmaf(ta, 2.0f - ta, -1.0f), fb = fmaf(tb, tb + 2.0f, 1.0f);
                const float blep0 = ta < 1.0f ? fa : (tb > -1.0f ? fb : 0.0f);
                double p2 = pos + 0.5;
                p2 = p2 >= 1.0 ? p2 - 1.0 : p2;
                const float ta2 = (float)p2 * inv_dt, tb2 = (float)(p2 - 1.0) * inv_dt;
                const float fa2 = fmaf(ta2, 2.0f - ta2, -1.0f), fb2 = fmaf(tb2, tb2 + 2.0f, 1.0f);
                const float blep1 = ta2 < 1.0f ? fa2 : (tb2 > -1.0f ? fb2 : 0.0f);
                float sq = (pos < 0.5 ? -1.0f : 1.0f) - (blep0 - blep1);
                if (fabsf(sq) < 2.0e-6f) sq = (pos < 0.5 ? -1.0f : 1.0f) - (float)(or_poly_blep(pos, delta) - or_poly_blep(fmod(pos + 0.5, 1.0), delta));
                square[i] = sq;
            }
        } else if (forms & EMU_OSC_F32_BLEP) {
            const float inv_dt = cv ? 1.0f / (float)delta : (float)(1.0 / delta);   /* osc_step: per sample where the pitch moves; inv_dt_f32 once where it does not */
            const float p32 = (float)pos;
            const double upper = 1.0 - delta;
            float blep0 = 0.0f, blep1 = 0.0f;
            if (o->antialiasing) blep0 = emu_poly_blep_sel(p32, (float)(pos - 1.0), inv_dt, pos < delta, pos > upper);
            saw[i] = fmaf(p32, 2.0f, -1.0f) - blep0;
            if (o->antialiasing) {
                double p2 = pos + 0.5;
                p2 = p2 >= 1.0 ? p2 - 1.0 : p2;
                blep1 = emu_poly_blep_sel((float)p2, (float)(p2 - 1.0), inv_dt, p2 < delta, p2 > upper);
            }
            float sq = (pos < 0.5 ? -1.0f : 1.0f) - (blep0 - blep1);
            if (o->antialiasing && fabsf(sq) < 2.0e-6f) /* square_sign_safe: the reference's own operations this close to zero */
                sq = (pos < 0.5 ? -1.0f : 1.0f) - (float)(or_poly_blep(pos, delta) - or_poly_blep(fmod(pos + 0.5, 1.0), delta));
            square[i] = sq;
        } else {
            square[i] = (pos < 0.5 ? -1.0f : 1.0f) - (o->antialiasing ? (float)(or_poly_blep(pos, delta) - or_poly_blep(fmod(pos + 0.5, 1.0), delta)) : 0.0f);
            saw[i] = ((float)pos * 2.0f - 1.0f) - (o->antialiasing ? (float)or_poly_blep(pos, delta) : 0.0f);
        }
        o->pos += delta;
        o->pos = fmod(o->pos, 1.0);
    }
}

/* ---- ladder (modules.hip.h: vcf_polys<true>, vcf_step<true>) ---------------------------------------------------------------------------- */
static void emu_vcf_state_calc(or_vcf* s, float input, float frequency, float res, float* lowpass, float* highpass, float* bandpass)
{
    if (frequency != s->sfreq || res != s->sres) {
        s->sfreq = frequency;
        s->sres = res;
        const float q = 1.0f - frequency;
        s->p = fmaf(0.8f * frequency, q, frequency);
        s->f = fmaf(s->p, 2.0f, -1.0f);
        s->q = res * fmaf(0.5f * q, fmaf(5.6f * q, q, 1.0f - q), 1.0f);
    }
    input = fmaf(-s->q, s->b[4], input);
    float t1 = s->b[1];
    s->b[1] = fmaf(input + s->b[0], s->p, -(s->b[1] * s->f));
    float t2 = s->b[2];
    s->b[2] = fmaf(s->b[1] + t1, s->p, -(s->b[2] * s->f));
    t1 = s->b[3];
    s->b[3] = fmaf(s->b[2] + t2, s->p, -(s->b[3] * s->f));
    s->b[4] = fmaf(s->b[3] + t1, s->p, -(s->b[4] * s->f));
    s->b[4] = fmaf(-(s->b[4] * s->b[4] * s->b[4]), 0.166667f, s->b[4]);
    s->b[0] = input;
    for (int k = 0; k < 5; k++) s->b[k] = fmaxf(fminf(s->b[k], 1.0f), -1.0f);
    *lowpass = s->b[4];
    *highpass = input - s->b[4];
    *bandpass = 3.0f * (s->b[3] - s->b[4]);
}
static void emu_calc_vcf(or_patch* p, or_module* m)
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
        emu_vcf_state_calc(v, audio, fminf(fmaxf(v->freq + cv * v->exp_amt, 0.0f), 0.9f), fminf(fmaxf(v->res, 0.0f), 1.0f), &lowpass[idx], &highpass[idx],
                           &bandpass[idx]);
    }
}

/* ---- NonLinear (modules.hip.h: powf_pos_loose) ------------------------------------------------------------------------------------------ */
static float emu_powf_pos_loose(float x, float b)
{
    const int normal = x >= 0x1p-126f;
    const float y = b * log2f(normal ? x : 1.0f);
    const int fast = normal && x < INFINITY && fabsf(y) < 32.0f;
    return fast ? exp2f(y) : powf(x, b);
}
static void emu_calc_nonlin(or_patch* p, or_module* m)
{
    const float* i1 = or_resolve(p, m, 0);
    const float* i2 = or_resolve(p, m, 1);
    float* out = m->out[0];
    for (uint32_t i = 0; i < p->buffer_size; i++) {
        const float a = i1 ? i1[i] : 0.0f, b = i2 ? i2[i] : m->u.nonlin.constant;
        out[i] = a > 0.0f ? emu_powf_pos_loose(a, b) : -emu_powf_pos_loose(-a, b);
    }
}

/* forms == 0: the reference's calc() again */
int emu_set_forms(or_patch* p, int module, uint32_t forms)
{
    if (module < 0 || module >= p->n_modules) return -1;
    or_calc_hook hook = NULL;
    if (forms) switch (p->modules[module].type) {
        case SRACK_MOD_OSCILLATOR: hook = emu_calc_osc; break;
        case SRACK_MOD_MOOG_FILTER: hook = emu_calc_vcf; break;
        case SRACK_MOD_NONLINEAR: hook = emu_calc_nonlin; break;
        default: return -2;
        }
    return or_set_calc_hook(p, module, hook, forms);
}
