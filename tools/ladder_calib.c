/* ladder_calib — where csrc/approx.cpp's ladder numbers come from (CPU only: gcc -O2 -ffp-contract=off tools/ladder_calib.c -lm).
 * Emulates, operation for operation, the reference's ladder (filter.rs:58-92: what the oracle and the exact render mode compute) and the
 * default mode's contracted form (csrc/modules.hip.h, vcf_step<true> / vcf_polys<true>: one product of each a*b - c*d folded into an fma),
 * both fed the reference's own saw (oscillator.rs:132-152), and measures
 *   own   <samples> <trials>   max |contracted - literal| / max(|literal|, 1) per port, by how the cutoff CV moves (kEpsLadder, kLadderRareJumps,
 *                              and why a cutoff that jumps at audio rate has no contracted form)
 *   gain  <samples> <trials>   the literal ladder's response to a +-2.4e-7 disturbance of its input, same cutoff motions, by resonance bucket
 *                              for the two kinds that jump (the static L1 norms no longer bound it: res + 0.1 and x 4 for edges, unbounded for noise)
 *   noisein <samples> <trials> `own` with white noise on the audio input, by resonance (a saw excites a resonance now and then, noise all the time)
 *   amp   <samples> <trials>   both by the input's AMPLITUDE (the reference clamps the stages, not the input): fine up to 1.75; from 1.9 up the ladder
 *                              is chaotic at low resonance — 1e4 .. 1e6 x — where the cutoff is high (kLadderDriveMax)
 *   tame  <samples> <trials>   ... and only there: by cutoff, for inputs far above the clamps (kLadderTameCutoff)
 *   l1    <samples>            L1 norms of the small-signal impulse responses against the measured gain and the sensitivity to the cutoff
 *                              (1.4 / cutoff x L1: approx.cpp uses 1.5)
 * The test suite runs `own` and `gain` at a small size against the constants (tests/test_approx.py). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { float f, p, q, b0, b1, b2, b3, b4, freq, res; } St;
static float clamp1(float x) { return fmaxf(fminf(x, 1.0f), -1.0f); }
static void coeffs(St* s, float fr, float res, int fast)
{
    if (fr == s->freq && res == s->res) return; /* filter.rs:61: only when (frequency, res) changed */
    s->freq = fr; s->res = res;
    float q = 1.0f - fr;
    if (fast) { s->p = fmaf(0.8f * fr, q, fr); s->f = fmaf(s->p, 2.0f, -1.0f); s->q = res * fmaf(0.5f * q, fmaf(5.6f * q, q, 1.0f - q), 1.0f); }
    else { s->p = fr + 0.8f * fr * q; s->f = s->p * 2.0f - 1.0f; s->q = res * (1.0f + 0.5f * q * (1.0f - q + 5.6f * q * q)); }
}
static void step(St* s, float in, int fast, float* lp, float* bp, float* hp)
{
    if (fast) {
        in = fmaf(-s->q, s->b4, in);
        float t1 = s->b1; s->b1 = fmaf(in + s->b0, s->p, -(s->b1 * s->f));
        float t2 = s->b2; s->b2 = fmaf(s->b1 + t1, s->p, -(s->b2 * s->f));
        t1 = s->b3; s->b3 = fmaf(s->b2 + t2, s->p, -(s->b3 * s->f));
        s->b4 = fmaf(s->b3 + t1, s->p, -(s->b4 * s->f));
        s->b4 = fmaf(-(s->b4 * s->b4 * s->b4), 0.166667f, s->b4);
    } else {
        in = in - (s->q * s->b4);
        float t1 = s->b1; s->b1 = (in + s->b0) * s->p - s->b1 * s->f;
        float t2 = s->b2; s->b2 = (s->b1 + t1) * s->p - s->b2 * s->f;
        t1 = s->b3; s->b3 = (s->b2 + t2) * s->p - s->b3 * s->f;
        s->b4 = (s->b3 + t1) * s->p - s->b4 * s->f;
        s->b4 = s->b4 - (s->b4 * s->b4 * s->b4) * 0.166667f;
    }
    s->b0 = clamp1(in); s->b1 = clamp1(s->b1); s->b2 = clamp1(s->b2); s->b3 = clamp1(s->b3); s->b4 = clamp1(s->b4);
    *lp = s->b4; *hp = in - s->b4; *bp = 3.0f * (s->b3 - s->b4);
}
static int l1(double fr, double res, int n, double out[3])
{
    double q0 = 1.0 - fr, p = fr + 0.8 * fr * q0, f = 2 * p - 1, q = res * (1 + 0.5 * q0 * (1 - q0 + 5.6 * q0 * q0));
    double b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, last = 0; out[0] = out[1] = out[2] = 0;
    for (int i = 0; i < n; i++) {
        double in = (i == 0 ? 1.0 : 0.0) - q * b4; double t1 = b1; b1 = (in + b0) * p - b1 * f; double t2 = b2; b2 = (b1 + t1) * p - b2 * f;
        t1 = b3; b3 = (b2 + t2) * p - b3 * f; b4 = (b3 + t1) * p - b4 * f; b0 = in;
        out[0] += fabs(b4); out[1] += fabs(3 * (b3 - b4)); out[2] += fabs(in - b4);
        if (i >= n - 64) last += fabs(b4);
        if (!(fabs(b4) < 1e6)) return 0;
    }
    return last < 1e-9;
}
static double rnd(unsigned long long* s) { *s = *s * 6364136223846793005ULL + 1442695040888963407ULL; return (double)(*s >> 11) / 9007199254740992.0; }
static double blep(double t, double dt) { if (dt == 0) return 0; if (t < dt) { t /= dt; return 2 * t - t * t - 1; } if (t > 1 - dt) { t = (t - 1) / dt; return t * t + 2 * t + 1; } return 0; }
static const char* kNames[] = {"none", "ramp", "sineLFO", "sine700", "saw", "square", "noise", "squareLFO"};
/* one trial: returns max relative difference per port between ladder A and ladder B.  mode 0: literal vs contracted, same input;
 * mode 1: literal vs literal with the input disturbed by +-2.4e-7 */
static int g_noise_in = 0;
static float g_amp = 1.0f; /* `amp`: the input's amplitude (the reference has no clamp in front of the ladder's first stage) */
static void trial(int mode, int kind, int N, float res, float fr, float ex, unsigned long long* seed, double m3[3])
{
    double delta = 440.0 * pow(2.0, rnd(seed) * 6 - 4) / 48000.0, pos = rnd(seed);
    double cd = (kind == 2 ? 3.0 : kind == 7 ? 5.0 : 440.0 * pow(2.0, rnd(seed) * 4 - 2)) / 48000.0, cp = rnd(seed);
    St a, b; memset(&a, 0, sizeof a); a.freq = -1; b = a; m3[0] = m3[1] = m3[2] = 0;
    for (int i = 0; i < N; i++) {
        float saw = g_amp * (((float)pos * 2.0f - 1.0f) - (float)blep(pos, delta)); pos = fmod(pos + delta, 1.0);
        if (g_noise_in) saw = g_amp * (float)(rnd(seed) * 2 - 1); /* `noisein`: white noise on the audio input instead of the saw */
        float cv = 0;
        switch (kind) {
        case 1: cv = (float)fabs(fmod(i / 20000.0, 2.0) - 1.0); break;
        case 2: case 3: cv = (float)sin(cp * M_PI * 2.0); break;
        case 4: cv = ((float)cp * 2.0f - 1.0f) - (float)blep(cp, cd); break;
        case 5: case 7: cv = (cp < 0.5 ? -1.0f : 1.0f) - (float)(blep(cp, cd) - blep(fmod(cp + 0.5, 1.0), cd)); break;
        case 6: cv = (float)(rnd(seed) * 2 - 1); break;
        }
        cp = fmod(cp + cd, 1.0);
        float f = fminf(fmaxf(fr + cv * ex, 0.0f), 0.9f); /* filter.rs:213 */
        coeffs(&a, f, res, 0); coeffs(&b, f, res, mode == 0);
        float l0, b0, h0, l1_, b1, h1;
        step(&a, saw, 0, &l0, &b0, &h0);
        step(&b, mode == 0 ? saw : saw + (rnd(seed) < 0.5 ? -2.4e-7f : 2.4e-7f), mode == 0, &l1_, &b1, &h1);
        double d[3] = {fabs((double)l1_ - l0) / fmax(fabs(l0), 1), fabs((double)b1 - b0) / fmax(fabs(b0), 1), fabs((double)h1 - h0) / fmax(fabs(h0), 1)};
        for (int k = 0; k < 3; k++) if (d[k] > m3[k]) m3[k] = d[k];
    }
}
int main(int argc, char** argv)
{
    const char* mode = argc > 1 ? argv[1] : "own";
    int N = argc > 2 ? atoi(argv[2]) : 100000, trials = argc > 3 ? atoi(argv[3]) : 300;
    unsigned long long seed = 777;
    if (!strcmp(mode, "noisein")) {
        /* `own` with white noise on the audio input, by resonance bucket, for the cutoff motions that have a contracted form */
        g_noise_in = 1;
        for (int kind = 0; kind < 8; kind++) {
            if (kind != 0 && kind != 1 && kind != 2 && kind != 7) continue;
            for (int rb = 0; rb < 9; rb++) {
                double worst[3] = {0, 0, 0};
                for (int t = 0; t < trials; t++) {
                    float res = (float)(rb * 0.1 + rnd(&seed) * 0.1 * (rb == 8 ? 0.9 : 1.0)), fr = (float)(0.02 + rnd(&seed) * 0.78), ex = (float)rnd(&seed); double m3[3];
                    trial(0, kind, N, res, fr, ex, &seed, m3);
                    for (int k = 0; k < 3; k++) if (m3[k] > worst[k]) worst[k] = m3[k];
                }
                printf("noisein %-9s res %.1f lp %.3e bp %.3e hp %.3e\n", kNames[kind], rb * 0.1, worst[0], worst[1], worst[2]);
            }
        }
        g_noise_in = 0;
    } else if (!strcmp(mode, "own")) {
        for (int kind = 0; kind < 8; kind++) {
            double worst[3] = {0, 0, 0};
            for (int t = 0; t < trials; t++) {
                float res = (float)(rnd(&seed) * 0.89), fr = (float)(0.02 + rnd(&seed) * 0.78), ex = (float)rnd(&seed); double m3[3];
                trial(0, kind, N, res, fr, ex, &seed, m3);
                for (int k = 0; k < 3; k++) if (m3[k] > worst[k]) worst[k] = m3[k];
            }
            printf("own %-9s lp %.3e bp %.3e hp %.3e\n", kNames[kind], worst[0], worst[1], worst[2]);
        }
    } else if (!strcmp(mode, "amp")) {
        /* both measurements by the input's amplitude, a still cutoff (up to 0.8) and a ramping one (up to the clamp at 0.9), resonance 0 - 0.89 */
        const float amps[] = {1, 1.5, 1.75, 1.9, 2.1, 3, 8};
        for (int ai = 0; ai < 7; ai++)
            for (int kind = 0; kind < 2; kind++) {
                double own = 0, gain = 0;
                g_amp = amps[ai];
                for (int t = 0; t < trials; t++) {
                    float res = (float)(rnd(&seed) * 0.89), fr = (float)(0.02 + rnd(&seed) * 0.78), ex = (float)rnd(&seed); double m3[3];
                    trial(0, kind, N, res, fr, ex, &seed, m3);
                    for (int k = 0; k < 3; k++) if (m3[k] > own) own = m3[k];
                    trial(1, kind, N, res, fr, ex, &seed, m3);
                    for (int k = 0; k < 3; k++) if (m3[k] > gain) gain = m3[k];
                }
                printf("amp %5g %-9s own %.3e gain %.3e\n", amps[ai], kNames[kind], own, gain / 2.4e-7);
            }
        g_amp = 1.0f;
    } else if (!strcmp(mode, "tame")) {
        /* the literal ladder's response to the disturbance by CUTOFF (still), for inputs far above the clamps */
        const float amps[] = {2.5, 8, 1000}, frs[] = {0.2, 0.3, 0.4, 0.5, 0.6};
        for (int ai = 0; ai < 3; ai++)
            for (int fi = 0; fi < 5; fi++) {
                double gain = 0;
                g_amp = amps[ai];
                for (int t = 0; t < trials; t++) {
                    float res = (float)(rnd(&seed) * 0.89), fr = frs[fi] - 0.05f * (float)rnd(&seed); double m3[3];
                    trial(1, 0, N, res, fr, 0.0f, &seed, m3);
                    for (int k = 0; k < 3; k++) if (m3[k] > gain) gain = m3[k];
                }
                printf("tame %5g cutoff %.1f gain %.3e\n", amps[ai], frs[fi], gain / 2.4e-7);
            }
        g_amp = 1.0f;
    } else if (!strcmp(mode, "gain")) {
        for (int kind = 0; kind < 8; kind++)
            for (int rb = 0; rb < 9; rb++) {
                double worst[3] = {0, 0, 0};
                for (int t = 0; t < trials; t++) {
                    float res = (float)(rb * 0.1 + rnd(&seed) * 0.1 * (rb == 8 ? 0.9 : 1.0)), fr = (float)(0.02 + rnd(&seed) * 0.78), ex = (float)rnd(&seed); double m3[3];
                    trial(1, kind, N, res, fr, ex, &seed, m3);
                    for (int k = 0; k < 3; k++) if (m3[k] > worst[k]) worst[k] = m3[k];
                }
                printf("gain %-9s res %.1f lp %.3e bp %.3e hp %.3e\n", kNames[kind], rb * 0.1, worst[0] / 2.4e-7, worst[1] / 2.4e-7, worst[2] / 2.4e-7);
            }
    } else {
        double ress[] = {0.0, 0.3, 0.5, 0.7, 0.8, 0.85, 0.89, 0.92, 0.95}, frs[] = {0.02, 0.05, 0.1, 0.2, 0.4, 0.6, 0.8, 0.9};
        for (int ri = 0; ri < 9; ri++) for (int fi = 0; fi < 8; fi++) {
            double L[3]; int ok = l1(frs[fi], ress[ri], 200000, L);
            /* sensitivity to the cutoff: the literal ladder at fr and fr + 1e-4, same saw */
            double cg[3] = {0, 0, 0};
            for (int rep = 0; rep < 3 && ok; rep++) {
                double delta = 440.0 * pow(2.0, rnd(&seed) * 6 - 4) / 48000.0, pos = rnd(&seed);
                St a, d; memset(&a, 0, sizeof a); a.freq = -1; d = a; coeffs(&a, (float)frs[fi], (float)ress[ri], 0); float fr2 = (float)frs[fi] + 1e-4f; coeffs(&d, fr2, (float)ress[ri], 0);
                double dfr = (double)fr2 - (double)(float)frs[fi];
                for (int i = 0; i < N; i++) {
                    float saw = ((float)pos * 2.0f - 1.0f) - (float)blep(pos, delta); pos = fmod(pos + delta, 1.0);
                    float l0, b0, h0, l3, b3, h3; step(&a, saw, 0, &l0, &b0, &h0); step(&d, saw, 0, &l3, &b3, &h3);
                    double c[3] = {fabs((double)l3 - l0) / dfr, fabs((double)b3 - b0) / dfr, fabs((double)h3 - h0) / dfr};
                    for (int k = 0; k < 3; k++) if (c[k] > cg[k]) cg[k] = c[k];
                }
            }
            if (ok) printf("l1 res %.2f fr %.2f lp %8.3g bp %8.3g hp %8.3g | cutoff gain x fr / L1: %6.2f %6.2f %6.2f\n", ress[ri], frs[fi], L[0], L[1], L[2],
                           cg[0] * frs[fi] / L[0], cg[1] * frs[fi] / L[1], cg[2] * frs[fi] / L[2]);
            else printf("l1 res %.2f fr %.2f does not decay\n", ress[ri], frs[fi]);
        }
    }
    return 0;
}
