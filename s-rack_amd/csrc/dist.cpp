// dist.cpp — the one collective of the path: sum of the per-GPU partial mixes over RCCL/xGMI.
// librccl is resolved at call time (dlopen) so the library also loads on hosts without it and
// binds to the RCCL already in the process (torch ships its own copy under the same soname).
#include <dlfcn.h>

#include <string>

#include "graph.hpp"

namespace {
using nccl_reduce_fn = int (*)(const void*, void*, size_t, int, int, int, void*, void*);
using nccl_err_fn = const char* (*)(int);
constexpr int kNcclFloat32 = 7;  // ncclFloat32
constexpr int kNcclSum = 0;      // ncclSum
}  // namespace

extern "C" int srack_dist_reduce_mix(void* comm, float* d_mix, size_t count, int root, void* stream)
{
    static nccl_reduce_fn reduce = nullptr;
    static nccl_err_fn errstr = nullptr;
    if (!reduce) {
        void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib) {
            srack::set_error(std::string("dist_reduce_mix: cannot load librccl: ") + dlerror());
            return SRACK_ERR_DEVICE;
        }
        reduce = (nccl_reduce_fn)dlsym(lib, "ncclReduce");
        errstr = (nccl_err_fn)dlsym(lib, "ncclGetErrorString");
        if (!reduce) {
            srack::set_error("dist_reduce_mix: ncclReduce not found in librccl");
            return SRACK_ERR_DEVICE;
        }
    }
    if (!comm || !d_mix) {
        srack::set_error("dist_reduce_mix: null communicator or buffer");
        return SRACK_ERR_INVALID;
    }
    int rc = reduce(d_mix, d_mix, count, kNcclFloat32, kNcclSum, root, comm, stream);
    if (rc != 0) {
        srack::set_error(std::string("ncclReduce: ") + (errstr ? errstr(rc) : "error"));
        return SRACK_ERR_DEVICE;
    }
    return SRACK_OK;
}
