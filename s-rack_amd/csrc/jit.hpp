// jit.hpp — voice kernels specialised for one flattened program at run time (jit.cpp).
#pragma once
#include <string>

#include "flatten.hpp"
#include "kernel_args.hip.h"

namespace srack {

constexpr int kMixRowsHost = 32;  // = kMixRows of wave.hip.h: samples per tile of the specialised and the fused kernels

struct JitKernel {
    void* module = nullptr;    // hipModule_t
    void* function = nullptr;  // hipFunction_t
};

// Can the generator express this program?  (`why` names the first obstacle.)
bool jit_supported(const FlatProgram& P, std::string* why = nullptr);
// Can every unit of the control program be generated too (co-scheduled with the voice blocks)?
bool jit_ctl_supported(const FlatPair& pair, std::string* why = nullptr);
// The HIP source of the kernel for the pair's voice program with the given output mode (1 frames, 2 mix, 3 both, 4 neither);
// with_ctl: the control program's units ride along as blocks [0, block0) of every launch (KernelArgs::ctl_slots).
int jit_source(const FlatPair& pair, int out_mode, bool with_ctl, std::string& src);
// Generate + compile for the current device's architecture (gfx950 when the process has no device); nothing is loaded.
int jit_compile_only(const FlatPair& pair, int out_mode, bool with_ctl);
// Generate, compile (cached per source) and load on the current device (cached per device).
int jit_get(const FlatPair& pair, int out_mode, bool with_ctl, const JitKernel** out);
int jit_launch(const JitKernel& k, const KernelArgs& ka, uint32_t n_blocks, void* stream);

}  // namespace srack
