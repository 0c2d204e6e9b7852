// ubench.hip — VALU instruction throughput / latency probes for gfx950 (measure, don't guess).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip ; run on the GPU box.
// Each kernel runs ITER iterations of 8 independent chains (throughput) or 1 chain (latency) per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

template <int OP, int CHAINS>
__global__ __launch_bounds__(256) void probe(float* out, float seed)
{
    float f[8];
    double d[8];
    int n[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f[k] = seed + threadIdx.x * 1e-3f + k;
        d[k] = (double)f[k] * 0.001;
        n[k] = (int)f[k];
    }
    const float a = seed * 0.99f, b = seed * 0.5f;
    const double da = (double)a, db = 1e-3;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int k = 0; k < CHAINS; k++) {
            if (OP == 0) f[k] = __builtin_fmaf(f[k], a, b);
            if (OP == 1) d[k] = d[k] + db;
            if (OP == 2) d[k] = __builtin_fma(d[k], da, db);
            if (OP == 3) d[k] = __builtin_amdgcn_fract(d[k] + 0.0) ;
            if (OP == 4) { f[k] = (float)d[k]; asm volatile("" : "+v"(f[k])); d[k] = __hiloint2double(__double2hiint(d[k]), __float_as_int(f[k])); }
            if (OP == 5) d[k] = __builtin_floor(d[k]);
            if (OP == 6) f[k] = __builtin_amdgcn_fmed3f(f[k], -1.0f, a);
            if (OP == 7) f[k] = fmaxf(fminf(f[k], a), -1.0f);
            if (OP == 8) f[k] = f[k] > b ? a : f[k];
            if (OP == 9) n[k] = n[k] + (n[k] >> 3);
            if (OP == 10) f[k] = f[k] * a;
            if (OP == 11) f[k] = f[k] + a;
            if (OP == 12) d[k] = d[k] * da;
            if (OP == 13) f[k] = 1.0f / f[k];
            if (OP == 14) { f[k] = __builtin_amdgcn_rcpf(f[k]); }
            if (OP == 15) { d[k] = d[k] >= 1.0 ? d[k] - 1.0 : d[k]; }
            if (OP == 16) {  // packed f32: two independent values per lane in one instruction
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {f[k], f[(k + 4) & 7]};
                f2 av = {a, a}, bv = {b, b};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(av), "v"(bv));
                f[k] = v.x;
                f[(k + 4) & 7] = v.y;
            }
            if (OP == 17) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {f[k], f[(k + 4) & 7]};
                f2 av = {a, a};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(av));
                f[k] = v.x;
                f[(k + 4) & 7] = v.y;
            }
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc += f[k] + (float)d[k] + (float)n[k];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int OP, int CHAINS>
static void run_occ(const char* name, float* d_out, int waves_per_simd)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;  // one 256-thread block = one wave per SIMD of a CU
    hipLaunchKernelGGL((probe<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per wave: ITER iterations of CHAINS ops; time per iteration per wave in ns
    printf("%-22s chains=%d waves/SIMD=%d  %7.3f ms  %6.2f ns per dependent step  (%5.2f ns per instr per SIMD)\n", name, CHAINS, waves_per_simd, ms,
           ms * 1e6 / ITER, ms * 1e6 / ITER / CHAINS / waves_per_simd);
}

template <int OP, int CHAINS>
static void run(const char* name, float* d_out, int ops_per_iter_per_chain)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU: 8 waves / SIMD
    hipLaunchKernelGGL((probe<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<OP, CHAINS>), dim3(blocks), dim3(256), 0, 0, d_out, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_insts = (double)blocks * 4 * ITER * CHAINS * ops_per_iter_per_chain;
    // cycles per wave-instruction per SIMD at an assumed 2.4 GHz (1024 SIMDs)
    const double cyc = ms * 1e-3 * 2.4e9 * 1024 / wave_insts;
    printf("%-28s chains=%d  %8.3f ms  %6.2f cyc/wave-inst/SIMD @2.4GHz   %7.2f T lane-ops/s\n", name, CHAINS, ms, cyc, wave_insts * 64 / (ms * 1e-3) / 1e12);
}

int main()
{
    float* d_out;
    hipMalloc(&d_out, 4096);
    run<0, 8>("v_fma_f32", d_out, 1);
    run<0, 1>("v_fma_f32 dependent", d_out, 1);
    run<10, 8>("v_mul_f32", d_out, 1);
    run<11, 8>("v_add_f32", d_out, 1);
    run<11, 1>("v_add_f32 dependent", d_out, 1);
    run<1, 8>("v_add_f64", d_out, 1);
    run<1, 1>("v_add_f64 dependent", d_out, 1);
    run<2, 8>("v_fma_f64", d_out, 1);
    run<12, 8>("v_mul_f64", d_out, 1);
    run<3, 8>("v_add_f64+v_fract_f64", d_out, 2);
    run<4, 8>("v_cvt_f32_f64 (+mov)", d_out, 1);
    run<5, 8>("v_floor_f64", d_out, 1);
    run<15, 8>("f64 wrap cmp+add+2cndmask", d_out, 4);
    run<6, 8>("v_med3_f32", d_out, 1);
    run<7, 8>("v_min+v_max f32", d_out, 2);
    run<8, 8>("v_cmp+v_cndmask f32", d_out, 2);
    run<9, 8>("v_ashr+v_add i32", d_out, 2);
    run<16, 4>("v_pk_fma_f32 (2 values)", d_out, 1);
    run<17, 4>("v_pk_mul_f32 (2 values)", d_out, 1);
    run<13, 8>("f32 divide (IEEE)", d_out, 1);
    run<14, 8>("v_rcp_f32", d_out, 1);
    for (int w : {1, 2, 4, 8}) run_occ<0, 1>("v_fma_f32 dep", d_out, w);
    for (int w : {1, 2, 4, 8}) run_occ<0, 2>("v_fma_f32 2 chains", d_out, w);
    for (int w : {1, 2, 4, 8}) run_occ<0, 4>("v_fma_f32 4 chains", d_out, w);
    for (int w : {1, 4}) run_occ<1, 1>("v_add_f64 dep", d_out, w);
    for (int w : {1, 4}) run_occ<6, 1>("v_med3_f32 dep", d_out, w);
    return 0;
}
