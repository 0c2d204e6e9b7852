/* tools/sine_check.c — the exact oscillator's sine (modules.hip.h, sine_exact_plain) against the host libm's, on the CPU.
 * The kernels evaluate sin(2 pi pos) by a degree-13 polynomial after an exact fold and take its f32 rounding wherever that is DECIDED: where
 * (float)(y - d) == (float)(y + d), d = 1e-13 y + 2e-15 covering the polynomial's 8e-14 y and the reference's own 1.6e-15 + 1.2e-16 y; elsewhere
 * the reference's expression itself is evaluated (osc_exact_cold).  This program restates polynomial and decision operation for operation and
 * counts, over N phases per family, the decided phases whose value is NOT `(pos * PI * 2.0).sin() as f32` (oscillator.rs:133) as glibc computes
 * it — must be 0 — and how many phases stay undecided.  Families: uniform in [0, 1); within 1e-3 ... 1e-12 of the quarter points (zeros and
 * peaks); phases that are multiples of 2^-k (what an oscillator started at 0 with a dyadic increment visits).
 * build: gcc -O2 -ffp-contract=off -o /tmp/sine_check tools/sine_check.c -lm -lpthread     usage: sine_check [millions per family per thread] [threads] */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double fold(double pos, uint32_t* sign)
{
    const double qn = 0.5 - pos;
    uint64_t b;
    memcpy(&b, &qn, 8);
    *sign = (uint32_t)(b >> 32) & 0x80000000u;
    const double t = fabs(qn) - 0.25;
    return 0.25 - fabs(t);
}
static int plain(double pos, float* out)   /* -> decided? */
{
    uint32_t sign;
    const double x = fold(pos, &sign);
    const double z = x * x;
    const double a01 = fma(-41.34170223990684, z, 6.283185307179272);
    const double a23 = fma(-76.70584757807868, z, 81.60524914955879);
    const double a45 = fma(-15.081496425342264, z, 42.05813586028645);
    const double z2 = z * z;
    const double b0 = fma(a23, z2, a01);
    const double b1 = fma(3.6659216216293173, z2, a45);
    const double z4 = z2 * z2;
    const double y = fma(b1, z4, b0) * x;
    const double d = fma(1.0e-13, y, 2.0e-15);
    const float r = (float)(y - d), r2 = (float)(y + d);
    uint32_t u, u2;
    memcpy(&u, &r, 4);
    memcpy(&u2, &r2, 4);
    u ^= sign;
    memcpy(out, &u, 4);
    return u2 == (u ^ sign) && pos >= 0.0 && pos < 1.0;
}
static uint64_t sm(uint64_t* s)
{
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
struct job { uint64_t seed, n; uint64_t wrong[3], undecided[3]; };
static void* run(void* arg)
{
    struct job* j = arg;
    static const double PI = 3.14159265358979323846264338327950288;
    uint64_t s = j->seed;
    for (int fam = 0; fam < 3; fam++)
        for (uint64_t i = 0; i < j->n; i++) {
            const uint64_t r = sm(&s);
            double pos;
            if (fam == 0) pos = (double)(r >> 11) * 0x1p-53;
            else if (fam == 1) {
                const double q = 0.25 * (double)(r & 3u);
                const double mag = pow(10.0, -3.0 - 9.0 * (double)((r >> 8) & 0xffff) / 65536.0);
                pos = q + ((r & 4u) ? mag : -mag) * ((double)(r >> 40) * 0x1p-24);
                pos = pos - floor(pos);
            } else {
                const int k = 8 + (int)((r >> 3) % 40);
                pos = ldexp((double)((r >> 12) & ((1ull << (k < 52 ? k : 52)) - 1)), -k);
                pos = pos - floor(pos);
            }
            float got;
            const int decided = plain(pos, &got);
            const float want = (float)sin(pos * PI * 2.0);
            if (!decided) j->undecided[fam]++;
            else if (memcmp(&got, &want, 4) != 0 && !(got == 0.0f && want == 0.0f)) {
                if (j->wrong[fam]++ < 3) fprintf(stderr, "family %d pos %.17g: %a, the libm's %a\n", fam, pos, got, want);
            }
        }
    return NULL;
}
int main(int argc, char** argv)
{
    const uint64_t n = (argc > 1 ? strtoull(argv[1], 0, 10) : 10) * 1000000ull;
    const int T = argc > 2 ? atoi(argv[2]) : 8;
    pthread_t th[64];
    struct job jobs[64];
    memset(jobs, 0, sizeof jobs);
    for (int t = 0; t < T; t++) {
        jobs[t].seed = 0x5EED0000ull + (uint64_t)t * 7919;
        jobs[t].n = n;
        pthread_create(&th[t], 0, run, &jobs[t]);
    }
    uint64_t wrong[3] = {0}, und[3] = {0};
    for (int t = 0; t < T; t++) {
        pthread_join(th[t], 0);
        for (int f = 0; f < 3; f++) wrong[f] += jobs[t].wrong[f], und[f] += jobs[t].undecided[f];
    }
    static const char* name[3] = {"uniform", "near the quarter points", "dyadic"};
    int bad = 0;
    for (int f = 0; f < 3; f++) {
        printf("%-24s %llu phases: %llu decided differently from the libm, %llu undecided (%.2e)\n", name[f], (unsigned long long)(n * T), (unsigned long long)wrong[f],
               (unsigned long long)und[f], (double)und[f] / (double)(n * T));
        bad |= wrong[f] != 0;
    }
    return bad;
}
