// dist.cpp — the multi-GPU half of the path: one process per GPU, voices sharded by global voice index, and ONE collective,
// the sum of the per-rank partial mixes over RCCL/xGMI.  The reference has no counterpart (it is
// single-threaded, src/main.rs:59-63).
//
//   srack_dist_unique_id   rank 0: ncclGetUniqueId -> 128 bytes the host hands to every rank (any side channel: a file,
//                          a socket, torch.distributed's store — the library does not care)
//   srack_dist_init        every rank: ncclCommInitRank on the calling thread's current device
//   srack_dist_comm_count  ncclCommCount: how many ranks the communicator really spans
//   srack_dist_reduce_mix  ncclReduce(sum, f32, root) of the [channels][T] partial mix, in place, on the caller's stream
//   srack_dist_destroy     ncclCommDestroy
//
// librccl is resolved at call time (dlopen) so the library also loads on hosts without it and binds to the RCCL already in
// the process (torch ships its own copy under the same soname).
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>

#include "graph.hpp"

namespace {

struct UniqueId {  // ncclUniqueId: 128 opaque bytes, passed BY VALUE to ncclCommInitRank
    char internal[128];
};
static_assert(sizeof(UniqueId) == SRACK_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");

using get_id_fn = int (*)(UniqueId*);
using init_rank_fn = int (*)(void**, int, UniqueId, int);
using destroy_fn = int (*)(void*);
using count_fn = int (*)(void*, int*);
using reduce_fn = int (*)(const void*, void*, size_t, int, int, int, void*, void*);
using err_fn = const char* (*)(int);
constexpr int kNcclFloat32 = 7;  // ncclFloat32
constexpr int kNcclSum = 0;      // ncclSum

struct Rccl {
    void* lib = nullptr;
    get_id_fn get_id = nullptr;
    init_rank_fn init_rank = nullptr;
    destroy_fn destroy = nullptr;
    count_fn count = nullptr;
    reduce_fn reduce = nullptr;
    err_fn errstr = nullptr;
};

// nullptr + srack error on failure.  Resolved once per process (call_once: two threads entering srack_dist_* together must not race
// on dlopen / dlsym and must never see a half-filled table); a failure keeps its reason for every later call.
const Rccl* rccl()
{
    static Rccl r;
    static std::string why;
    static bool ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!r.lib) {
            const char* e = dlerror();
            why = std::string("dist: cannot load librccl: ") + (e ? e : "unknown error");
            return;
        }
        r.get_id = (get_id_fn)dlsym(r.lib, "ncclGetUniqueId");
        r.init_rank = (init_rank_fn)dlsym(r.lib, "ncclCommInitRank");
        r.destroy = (destroy_fn)dlsym(r.lib, "ncclCommDestroy");
        r.count = (count_fn)dlsym(r.lib, "ncclCommCount");
        r.reduce = (reduce_fn)dlsym(r.lib, "ncclReduce");
        r.errstr = (err_fn)dlsym(r.lib, "ncclGetErrorString");
        if (!r.get_id || !r.init_rank || !r.destroy || !r.count || !r.reduce) {
            why = "dist: librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclCommCount / ncclReduce";
            return;
        }
        ok = true;
    });
    if (!ok) srack::set_error(why);
    return ok ? &r : nullptr;
}

// No exception crosses the C boundary (the entry points build std::strings): as capi.cpp's guarded().
template <class F>
int guarded(F&& f) noexcept
{
    try {
        return f();
    } catch (const std::bad_alloc&) {
        try { srack::set_error("out of memory"); } catch (...) {}
        return SRACK_ERR_NOMEM;
    } catch (const std::exception& e) {
        try { srack::set_error(std::string("internal error: ") + e.what()); } catch (...) {}
        return SRACK_ERR_INVALID;
    } catch (...) {
        try { srack::set_error("internal error"); } catch (...) {}
        return SRACK_ERR_INVALID;
    }
}

int fail(const Rccl* r, const char* what, int rc)
{
    srack::set_error(std::string(what) + ": " + (r->errstr ? r->errstr(rc) : "error") + " (ncclResult " + std::to_string(rc) + ")");
    return SRACK_ERR_DEVICE;
}

}  // namespace

extern "C" int srack_dist_unique_id(void* id_out)
{
    return guarded([&]() -> int {
        if (!id_out) {
            srack::set_error("dist_unique_id: null buffer");
            return SRACK_ERR_INVALID;
        }
        const Rccl* r = rccl();
        if (!r) return SRACK_ERR_DEVICE;
        UniqueId id;
        std::memset(&id, 0, sizeof id);
        const int rc = r->get_id(&id);
        if (rc != 0) return fail(r, "ncclGetUniqueId", rc);
        std::memcpy(id_out, &id, sizeof id);
        return SRACK_OK;
    });
}

extern "C" int srack_dist_init(const void* id, int n_ranks, int rank, void** comm_out)
{
    return guarded([&]() -> int {
        if (!id || !comm_out || n_ranks < 1 || rank < 0 || rank >= n_ranks) {
            srack::set_error("dist_init: null id / communicator pointer, or rank outside [0, n_ranks)");
            return SRACK_ERR_INVALID;
        }
        *comm_out = nullptr;
        const Rccl* r = rccl();
        if (!r) return SRACK_ERR_DEVICE;
        UniqueId u;
        std::memcpy(&u, id, sizeof u);
        void* comm = nullptr;
        const int rc = r->init_rank(&comm, n_ranks, u, rank);
        if (rc != 0) return fail(r, "ncclCommInitRank", rc);
        *comm_out = comm;
        return SRACK_OK;
    });
}

extern "C" int srack_dist_comm_count(void* comm, int* n_ranks)
{
    return guarded([&]() -> int {
        if (!comm || !n_ranks) {
            srack::set_error("dist_comm_count: null communicator or result pointer");
            return SRACK_ERR_INVALID;
        }
        const Rccl* r = rccl();
        if (!r) return SRACK_ERR_DEVICE;
        const int rc = r->count(comm, n_ranks);
        if (rc != 0) return fail(r, "ncclCommCount", rc);
        return SRACK_OK;
    });
}

extern "C" int srack_dist_destroy(void* comm)
{
    return guarded([&]() -> int {
        if (!comm) return SRACK_OK;
        const Rccl* r = rccl();
        if (!r) return SRACK_ERR_DEVICE;
        const int rc = r->destroy(comm);
        if (rc != 0) return fail(r, "ncclCommDestroy", rc);
        return SRACK_OK;
    });
}

extern "C" int srack_dist_reduce_mix(void* comm, float* d_mix, size_t count, int root, void* stream)
{
    return guarded([&]() -> int {
        if (!comm || !d_mix) {
            srack::set_error("dist_reduce_mix: null communicator or buffer");
            return SRACK_ERR_INVALID;
        }
        const Rccl* r = rccl();
        if (!r) return SRACK_ERR_DEVICE;
        const int rc = r->reduce(d_mix, d_mix, count, kNcclFloat32, kNcclSum, root, comm, stream);
        if (rc != 0) return fail(r, "ncclReduce", rc);
        return SRACK_OK;
    });
}
