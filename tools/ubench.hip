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
            if (OP == 18) {  // v_permlane32_swap: two registers exchanged in place
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(f[k]), "+v"(f[(k + 4) & 7]));
            }
            if (OP == 19) {
                asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(f[k]), "+v"(f[(k + 4) & 7]));
            }
            if (OP == 20) {
                asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(f[k]));
            }
            if (OP == 21) {
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(f[k]));
            }
            if (OP == 22) asm volatile("v_mul_f32_e64 %0, %0, -%1" : "+v"(f[k]) : "v"(a));
            if (OP == 23) asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(f[k]));
            if (OP == 24) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %2\n\tv_addc_co_u32_e32 %1, vcc, %1, %2, vcc" : "+v"(n[k]), "+v"(f[k]) : "v"(a) : "vcc");
            if (OP == 25) asm volatile("v_sub_f32_e64 %0, 1.0, %0 clamp" : "+v"(f[k]));
            if (OP == 26) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(a) : "vcc");
            if (OP == 27) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(f[k]), "v"(a) : "vcc");
            if (OP == 28) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(f[k]) : "v"(a), "v"(b));
            if (OP == 29) asm volatile("v_fma_f32 %0, %0, %1, -1.0" : "+v"(f[k]) : "s"(a));
            if (OP == 30) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(f[k]) : "v"(a), "s"(0x5555555555555555ull));
            if (OP == 31) asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(f[k]) : "v"(a));
            if (OP == 32) asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(f[k]) : "s"(a));
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
    run<18, 4>("v_permlane32_swap", d_out, 1);
    run<19, 4>("v_permlane16_swap", d_out, 1);
    run<20, 8>("v_add_f32_dpp row_mirror", d_out, 1);
    run<21, 8>("v_add_f32_dpp quad_perm", d_out, 1);
    run<22, 8>("v_mul_f32_e64 neg", d_out, 1);
    run<23, 8>("v_cvt_f32_u32", d_out, 1);
    run<24, 8>("v_add_co+v_addc_co", d_out, 2);
    run<25, 8>("v_sub_f32_e64 clamp", d_out, 1);
    run<26, 8>("v_cndmask_e32 (vcc)", d_out, 1);
    run<27, 8>("v_cmp_lt_f32 (vcc)", d_out, 1);
    run<28, 8>("v_fmac_f32_e32", d_out, 1);
    run<29, 8>("v_fma_f32 sgpr operand", d_out, 1);
    run<30, 8>("v_cndmask_e64 (sgpr mask)", d_out, 1);
    run<31, 8>("v_max_f32", d_out, 1);
    run<32, 8>("v_mul_f32 sgpr operand", d_out, 1);
    run<13, 8>("f32 divide (IEEE)", d_out, 1);
    run<14, 8>("v_rcp_f32", d_out, 1);
    for (int w : {1, 2, 4, 8}) run_occ<0, 1>("v_fma_f32 dep", d_out, w);
    for (int w : {1, 2, 4, 8}) run_occ<0, 2>("v_fma_f32 2 chains", d_out, w);
    for (int w : {1, 2, 4, 8}) run_occ<0, 4>("v_fma_f32 4 chains", d_out, w);
    for (int w : {1, 4}) run_occ<1, 1>("v_add_f64 dep", d_out, w);
    for (int w : {1, 4}) run_occ<6, 1>("v_med3_f32 dep", d_out, w);
    // f64 at the occupancy of the FM kernels (65 536 voices = one wave per SIMD): how many independent chains fill the pipe?
    for (int w : {1, 2, 4}) run_occ<2, 1>("v_fma_f64 dep", d_out, w);
    for (int w : {1, 2, 4}) run_occ<2, 2>("v_fma_f64 2 chains", d_out, w);
    for (int w : {1, 2, 4}) run_occ<2, 4>("v_fma_f64 4 chains", d_out, w);
    for (int w : {1, 2}) run_occ<2, 8>("v_fma_f64 8 chains", d_out, w);
    for (int w : {1, 4}) run_occ<12, 4>("v_mul_f64 4 chains", d_out, w);
    for (int w : {1, 4}) run_occ<5, 4>("v_floor_f64 4 chains", d_out, w);
    for (int w : {1, 4}) run_occ<4, 4>("v_cvt_f32_f64 4 chains", d_out, w);
    return 0;
}
