// comm_rccl.hpp -- the collectives of the sharded path behind the C ABI (included by capi.hip only).
//
// The path shards by contiguous camera ranges; what the shards exchange is tiny: the 8-byte sum
// behind BAProblem::total_reprojection_error (src/baproblem.rs:265-279) and the 20- / 3-double shares of the
// statistics (src/baproblem.rs:282-337, src/noise.rs:75-87).  They travel through RCCL over xGMI.  RCCL is bound at
// first use with dlopen (SONAME librccl.so.1): a single-GPU host never loads the 570 MB library, and a process that
// already holds a copy (PyTorch-ROCm bundles one under the same SONAME) shares it instead of loading a second one.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and enums only; every function is resolved with dlsym

#include <cstdlib>
#include <mutex>
#include <string>

namespace c2b {

struct RcclApi {
    void *handle = nullptr;
    std::string error, path;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return handle != nullptr && error.empty(); }
};

inline RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *env = std::getenv("C2B_RCCL_LIB");
        const char *names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        std::string tried;
        for (const char *n : names) {
            if (!n || !*n) continue;
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) { api.path = n; break; }
            const char *e = dlerror();
            tried += std::string(n) + ": " + (e ? e : "?") + "; ";
        }
        if (!api.handle) { api.error = "could not load RCCL (" + tried + ")"; return; }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + name;
            return p;
        };
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return api;
}

}  // namespace c2b

// one rank's membership of a communicator (opaque at the ABI)
struct c2b_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};
