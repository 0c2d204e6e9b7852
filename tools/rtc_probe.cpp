// Does hiprtc work on the GPU box (no network, same image)?  Compiles a kernel at run time, loads it, runs it, checks the result.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <chrono>
#include <cstdio>
#include <string>
#include <vector>
int main()
{
    const char* src = R"(
extern "C" __global__ void k(float* p, float a) { int i = blockIdx.x * blockDim.x + threadIdx.x; p[i] = __builtin_fmaf(p[i], a, 1.0f); }
)";
    auto t0 = std::chrono::steady_clock::now();
    hiprtcProgram prog;
    if (hiprtcCreateProgram(&prog, src, "k.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) return 1;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, 0);
    hiprtcGetProgramLog(prog, &log[0]);
    printf("compile: %s\n%s\n", hiprtcGetErrorString(r), log.c_str());
    if (r != HIPRTC_SUCCESS) return 2;
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    std::vector<char> code(cs);
    hiprtcGetCode(prog, code.data());
    auto t1 = std::chrono::steady_clock::now();
    hipModule_t mod;
    hipFunction_t fn;
    if (hipModuleLoadData(&mod, code.data()) != hipSuccess) return 3;
    if (hipModuleGetFunction(&fn, mod, "k") != hipSuccess) return 4;
    float* d;
    hipMalloc(&d, 256 * 4);
    std::vector<float> h(256, 2.0f);
    hipMemcpy(d, h.data(), 256 * 4, hipMemcpyHostToDevice);
    float a = 3.0f;
    void* args[] = {&d, &a};
    if (hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, nullptr, args, nullptr) != hipSuccess) return 5;
    hipMemcpy(h.data(), d, 256 * 4, hipMemcpyDeviceToHost);
    printf("result %f (expect 7), compile %.1f ms\n", h[0], std::chrono::duration<double, std::milli>(t1 - t0).count());
    return h[0] == 7.0f ? 0 : 6;
}
