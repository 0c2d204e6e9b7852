// energybench.hip — what does ONE wave-instruction of each class cost in ENERGY on this part?  (tools/energy_probe.sh samples rocm-smi
// while each variant runs for a few seconds.)  The flagship and cfg3_poly run at the package power cap, where time is energy / 1400 W:
// which instructions carry the energy decides what is worth removing.  Every variant: 4096 one-wave workgroups (four waves per SIMD, as
// config 3), a loop of 64 inline-asm instructions of one class on independent registers.
//   hipcc --offload-arch=gfx950 -O2 -o tools/energybench tools/energybench.hip ;  ./tools/energybench <variant> <seconds>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

template <int kVariant>
__global__ __launch_bounds__(64) void k(float* out, int iters, float seed)
{
    float a = seed + threadIdx.x, b = a * 1.0001f, c = a * 0.9999f, d = 0.5f, e = 0.25f, f = 0.125f, g = 1.5f, h = 2.5f;
    double p = a, q = b, r = c, s = d;
    unsigned u = threadIdx.x, v = 12345u, w = 777u, x = 99u;
    float* mine = out + (size_t)blockIdx.x * 64 + threadIdx.x;
    __shared__ float lds[64 * 4];
    for (int i = 0; i < iters; i++) {
        if (kVariant == 0) {  // v_fma_f32, three VGPR sources, four independent chains
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (kVariant == 1) {  // v_add_f32 (two sources)
            REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));)
        } else if (kVariant == 2) {  // v_mul_f32
            REP16(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(g));)
        } else if (kVariant == 3) {  // v_med3_f32 (clamp to [-1, 1])
            REP16(asm volatile("v_med3_f32 %0, %0, -1.0, 1.0\n v_med3_f32 %1, %1, -1.0, 1.0\n v_med3_f32 %2, %2, -1.0, 1.0\n v_med3_f32 %3, %3, -1.0, 1.0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (kVariant == 4) {  // v_cmp_lt_f32 + v_cndmask (a select)
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %2, vcc\n v_cmp_lt_f32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e) : "vcc");)
        } else if (kVariant == 5) {  // v_add_f64
            REP16(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"((double)e));)
        } else if (kVariant == 6) {  // v_fma_f64
            REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(p), "+v"(q), "+v"(r), "+v"(s) : "v"((double)e), "v"((double)f));)
        } else if (kVariant == 7) {  // integer add with carry + v_cvt_f32_u32 (the fixed-point phase)
            REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_cvt_f32_u32 %2, %1\n v_add_u32 %3, %3, %4" : "+v"(u), "+v"(v), "+v"(a), "+v"(x) : "v"(w) : "vcc");)
        } else if (kVariant == 8) {  // v_mov_b32
            REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (kVariant == 9) {  // one ds_write_b32 per four v_fma (LDS traffic of the mix tile, denser than the kernels': x8)
            REP16(asm volatile("ds_write_b32 %4, %0\n v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %1\n v_fma_f32 %3, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"((unsigned)(threadIdx.x * 4u)) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (kVariant == 10) {  // the same without the ds_write (the reference for 9)
            REP16(asm volatile("v_fma_f32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %3, %1\n v_fma_f32 %3, %3, %1, %2" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (kVariant == 11) {  // scalar ALU
            unsigned s0 = i, s1 = 3;
            REP16(asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0\n s_add_u32 %0, %0, %1\n s_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1));)
            u += s0;
        } else if (kVariant == 12) {  // v_fma_f32 with the constant operands in SGPRs (one VGPR source instead of three)
            float se = 0.25f;  // (one SGPR per instruction: the constant bus)
            REP16(asm volatile("v_fma_f32 %0, %0, %4, 0.5\n v_fma_f32 %1, %1, %4, 0.5\n v_fma_f32 %2, %2, %4, 0.5\n v_fma_f32 %3, %3, %4, 0.5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(se));)
        } else if (kVariant == 13) {  // nothing but the loop (s_add, s_cmp, branch): the floor
            asm volatile("s_nop 0");
        } else if (kVariant == 14) {  // v_fract_f64
            REP16(asm volatile("v_fract_f64 %0, %0\n v_fract_f64 %1, %1\n v_fract_f64 %2, %2\n v_fract_f64 %3, %3" : "+v"(p), "+v"(q), "+v"(r), "+v"(s));)
        } else if (kVariant == 16 || kVariant == 17) {  // v_fma_f32 with ONE lane (16) / SIXTEEN lanes (17) enabled: what does a predicated-off lane cost?
            asm volatile("s_mov_b64 exec, %0" ::"s"(kVariant == 16 ? 1ull : 0xffffull));
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
            asm volatile("s_mov_b64 exec, -1");
        } else if (kVariant == 15) {  // v_pk_fma_f32 (two values per lane and instruction)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 pa = {a, b}, pb = {c, d}, pc = {e, f}, pd = {g, h};
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3" : "+v"(pa), "+v"(pb) : "v"(pc), "v"(pd));)
            a = pa.x + pa.y; c = pb.x + pb.y;
        }
    }
    *mine = a + b + c + d + (float)(p + q + r + s) + (float)(u + v + x) + lds[threadIdx.x];
}

template <int V>
static void run(float* out, double seconds)
{
    int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(4096), dim3(64), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<V>, dim3(4096), dim3(64), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const int n = (int)(seconds * 1e3 / ms) + 1;
    const auto t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < n; j++) hipLaunchKernelGGL(k<V>, dim3(4096), dim3(64), 0, 0, out, iters, 1.0f);
    hipDeviceSynchronize();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // wave-instructions of the class per launch: 4096 waves x iters x 64 (variants 9: 16 ds_write + 48 fma; 10: 48; 13: 1 nop; 15: 32 pk)
    printf("{\"variant\": %d, \"launches\": %d, \"seconds\": %.3f, \"ms_per_launch\": %.4f, \"waves\": 4096, \"iters\": %d}\n", V, n, dt, dt / n * 1e3, iters);
}

int main(int argc, char** argv)
{
    const int v = argc > 1 ? atoi(argv[1]) : 0;
    const double sec = argc > 2 ? atof(argv[2]) : 4.0;
    float* out; hipMalloc(&out, 4096 * 64 * 4);
    switch (v) {
#define C(n) case n: run<n>(out, sec); break;
        C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17)
    }
    return 0;
}
