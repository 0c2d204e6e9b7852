// storebench.hip — how fast can 4096 waves stream frames [T][V] to HBM with different store shapes?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: the render kernels' pattern: per sample one dword per lane, row stride V (256 B per wave per row)
__global__ __launch_bounds__(64) void store_rows(float* out, uint32_t V, uint32_t T, int work)
{
    float* p = out + (size_t)blockIdx.x * 64 + threadIdx.x;
    float x = threadIdx.x * 1e-3f;
    for (uint32_t t = 0; t < T; t++) {
        for (int k = 0; k < work; k++) x = __builtin_fmaf(x, 0.999f, 0.001f);  // dependent VALU filler
        *p = x;
        p += V;
    }
}
// B: 4 rows staged, then one dwordx4 per lane covering 4 consecutive voices of one row (lane -> row = lane/16)
__global__ __launch_bounds__(64) void store_x4(float* out, uint32_t V, uint32_t T, int work)
{
    __shared__ float tile[4 * 64];
    const int lane = threadIdx.x;
    float x = lane * 1e-3f;
    float* base = out + (size_t)blockIdx.x * 64;
    for (uint32_t t = 0; t < T; t += 4) {
        for (int r = 0; r < 4; r++) {
            for (int k = 0; k < work; k++) x = __builtin_fmaf(x, 0.999f, 0.001f);
            tile[r * 64 + lane] = x;
        }
        __syncthreads();
        const int r = lane >> 4, c = (lane & 15) * 4;
        float4 v = *reinterpret_cast<float4*>(&tile[r * 64 + c]);
        *reinterpret_cast<float4*>(base + (size_t)(t + r) * V + c) = v;
        __syncthreads();
    }
}
// C: wave-tiled layout [V/64][T][64]: each wave streams one contiguous 12 MB region
__global__ __launch_bounds__(64) void store_tiled(float* out, uint32_t V, uint32_t T, int work)
{
    float* p = out + (size_t)blockIdx.x * T * 64 + threadIdx.x;
    float x = threadIdx.x * 1e-3f;
    for (uint32_t t = 0; t < T; t++) {
        for (int k = 0; k < work; k++) x = __builtin_fmaf(x, 0.999f, 0.001f);
        *p = x;
        p += 64;
    }
}
// D: grid-stride float4 fill (the streaming-store ceiling)
__global__ __launch_bounds__(256) void fill4(float4* out, size_t n4)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}

int main()
{
    const uint32_t V = 262144, T = 12000;  // 12.6 GB
    float* d;
    CK(hipMalloc(&d, (size_t)V * T * 4));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %8.3f ms  %8.1f GB/s\n", name, ms, (double)V * T * 4 / (ms * 1e-3) / 1e9);
    };
    for (int work : {0, 16, 32, 48, 64}) {
        char nm[64];
        snprintf(nm, 64, "rows dword (work=%d)", work);
        time(nm, [&] { hipLaunchKernelGGL(store_rows, dim3(V / 64), dim3(64), 0, 0, d, V, T, work); });
    }
    for (int work : {0, 32, 48}) {
        char nm[64];
        snprintf(nm, 64, "rows dwordx4 via LDS (work=%d)", work);
        time(nm, [&] { hipLaunchKernelGGL(store_x4, dim3(V / 64), dim3(64), 0, 0, d, V, T, work); });
    }
    for (int work : {0, 32, 48}) {
        char nm[64];
        snprintf(nm, 64, "wave-tiled dword (work=%d)", work);
        time(nm, [&] { hipLaunchKernelGGL(store_tiled, dim3(V / 64), dim3(64), 0, 0, d, V, T, work); });
    }
    time("grid-stride float4 fill", [&] { hipLaunchKernelGGL(fill4, dim3(2048), dim3(256), 0, 0, (float4*)d, (size_t)V * T / 4); });
    return 0;
}
