// jit.hpp — voice kernels specialised for one flattened program at run time (jit.cpp).
#pragma once
#include <cstdint>
#include <memory>
#include <string>

#include "flatten.hpp"
#include "kernel_args.hip.h"

namespace srack {

constexpr int kMixRowsHost = 32;  // = kMixRows of wave.hip.h: samples per tile of the specialised and the fused kernels

struct JitKernel {  // a loaded module: shared between the process-wide cache and every patch that renders with it
    void* module = nullptr;    // hipModule_t
    void* function = nullptr;  // hipFunction_t
    JitKernel() = default;
    JitKernel(const JitKernel&) = delete;
    JitKernel& operator=(const JitKernel&) = delete;
    ~JitKernel();              // hipModuleUnload, once the last owner lets go
};

struct JitFetchInfo {  // how one kernel was come by
    int how = 0;              // 0: this process had it (memory), 1: the disk cache, 2: compiled now
    double compile_ms = 0.0;  // hiprtc time when how == 2
    int waves = 0;            // the register budget the kernel was compiled under (waves per SIMD; 0: none asked for)
    int vgprs = 0;            // its VGPRs
};

struct JitCacheStats {  // process-wide, since start (srack_kernel_cache_stats)
    uint64_t compiled = 0, disk_hits = 0, memory_hits = 0, modules_loaded = 0, code_evictions = 0, module_evictions = 0;
    uint64_t resident_code_objects = 0, resident_modules = 0;
    double compile_ms = 0.0;
    char directory[512] = {0};  // the disk cache in use ("" = none)
};

// Can the generator express this program?  (`why` names the first obstacle.)
bool jit_supported(const FlatProgram& P, std::string* why = nullptr);
// Can every unit of the control program be generated too (co-scheduled with the voice blocks)?
bool jit_ctl_supported(const FlatPair& pair, std::string* why = nullptr);
// The HIP source of the kernel for the pair's voice program with the given output mode (1 frames, 2 mix, 3 both, 4 neither);
// with_ctl: the control program's units ride along as blocks [0, block0) of every launch (KernelArgs::ctl_slots).
// waves: > 0 asks the compiler for a register budget that lets so many waves share a SIMD (amdgpu_waves_per_eu): see jit_get.
int jit_source(const FlatPair& pair, int out_mode, bool with_ctl, std::string& src, int waves = 0, int want_waves = 0);  // want_waves: waves per SIMD the render has for the kernel (what LDS it may spend)
// Generate + compile for the current device's architecture (gfx950 when the process has no device); nothing is loaded.
int jit_compile_only(const FlatPair& pair, int out_mode, bool with_ctl);
// Generate, fetch the code object (memory -> disk -> hiprtc; jit.cpp "the kernel cache") and load it on the current device.
// want_waves: how many waves per SIMD the render has for this kernel (voices / 64 / 1024 SIMDs, at most 4).  A kernel whose registers
// allow fewer is compiled again under the budget for that many (then for half as many) and the tighter kernel is taken IF it does not
// spill (`how->waves`: the budget taken, 0: the compiler's own choice): more waves hide a recurrence's latency, scratch traffic does not.
int jit_get(const FlatPair& pair, int out_mode, bool with_ctl, std::shared_ptr<const JitKernel>* out, JitFetchInfo* how = nullptr, int want_waves = 1);
// The disk cache's directory: a path, "off", or nullptr for the default resolution (SRACK_KERNEL_CACHE_DIR, next to the library, ~/.cache).
int jit_cache_set_dir(const char* dir);
JitCacheStats jit_cache_stats();
int jit_launch(const JitKernel& k, const KernelArgs& ka, uint32_t n_blocks, void* stream);

}  // namespace srack
