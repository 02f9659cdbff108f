// Thin RCCL wrappers of the C ABI (SURVEY 8b's export list: lc_allgather_f32 / lc_allreduce_sum_f32 and friends): what the
// voxel-sharded fit exchanges over xGMI -- the V-independent f32 operators (all-gather), the per-alpha score sums and the
// alpha histogram (all-reduce), the packed per-fold results (all-gather of f64) -- as plain calls on device pointers and a
// HIP stream, with an opaque communicator created from a unique id the host side passes around by whatever means it has
// (litcoder_core_amd/dist.py: one torch.distributed object broadcast, once).  north_star wants PyTorch for containers
// only; through round 4 the collectives themselves went through torch.distributed (since round 6 that is the transport
// under gloo only: with the "nccl" backend ShardContext takes these wrappers by default).
//
// RCCL is NOT a link-time dependency of this library: the entry points are looked up at the first use -- in the RCCL the
// process has loaded already (PyTorch-ROCm ships one), else in librccl.so.1 of the ROCm installation -- so a single-GPU
// process never loads it.  Nor is its header a hard BUILD dependency: on a ROCm installation without
// <rccl/rccl.h> the library still builds and every entry point below returns LC_E_HIP with a message saying so.
#include "lc_common.h"

#include <cstring>
#include <dlfcn.h>
#include <mutex>

#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define LC_HAVE_RCCL_HEADER 1
#else
#define LC_HAVE_RCCL_HEADER 0
#endif

#if !LC_HAVE_RCCL_HEADER
struct lc_comm {
    int unused;
};
#define LC_NO_RCCL(who) return lc::fail(LC_E_HIP, who ": this library was built without <rccl/rccl.h>: no RCCL transport")
extern "C" int lc_comm_unique_id_bytes(void) { return 0; }
extern "C" int lc_comm_unique_id(void*, int) { LC_NO_RCCL("lc_comm_unique_id"); }
extern "C" int lc_comm_create(const void*, int, int, int, int, lc_comm_t**) { LC_NO_RCCL("lc_comm_create"); }
extern "C" int lc_comm_destroy(lc_comm_t*) { LC_NO_RCCL("lc_comm_destroy"); }
extern "C" int lc_allgather_bytes(lc_comm_t*, const void*, void*, int64_t, lc_stream_t) { LC_NO_RCCL("lc_allgather_bytes"); }
extern "C" int lc_allgather_f32(lc_comm_t*, const float*, float*, int64_t, lc_stream_t) { LC_NO_RCCL("lc_allgather_f32"); }
extern "C" int lc_allreduce(lc_comm_t*, void*, int64_t, int, int, lc_stream_t) { LC_NO_RCCL("lc_allreduce"); }
extern "C" int lc_allreduce_sum_f32(lc_comm_t*, float*, int64_t, lc_stream_t) { LC_NO_RCCL("lc_allreduce_sum_f32"); }
#else

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

RcclApi& api() {
    static RcclApi a;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);              // the copy the process has already (PyTorch's)
            if (a.lib) break;
        }
        if (!a.lib)
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (a.lib) break;
            }
        if (!a.lib) return;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.lib, "ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce && a.GetErrorString;
    });
    return a;
}

#define LC_RCCL(call)                                                                                     \
    do {                                                                                                  \
        ncclResult_t r_ = (call);                                                                         \
        if (r_ != ncclSuccess) return lc::fail(LC_E_HIP, "%s: %s", #call, api().GetErrorString(r_));      \
    } while (0)

int need_api(const char* who) {
    if (!api().ok) return lc::fail(LC_E_HIP, "%s: RCCL (librccl.so.1) could not be loaded", who);
    return LC_OK;
}

}  // namespace

struct lc_comm {
    ncclComm_t comm;
    int rank, world;
};

extern "C" int lc_comm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }

extern "C" int lc_comm_unique_id(void* h_id, int bytes) {
    LC_REQUIRE(h_id && bytes == (int)sizeof(ncclUniqueId), LC_E_BADARG, "lc_comm_unique_id: need a %d-byte buffer",
               (int)sizeof(ncclUniqueId));
    if (int rc = need_api("lc_comm_unique_id")) return rc;
    LC_RCCL(api().GetUniqueId(static_cast<ncclUniqueId*>(h_id)));
    return LC_OK;
}

extern "C" int lc_comm_create(const void* h_id, int bytes, int world, int rank, int device, lc_comm_t** out) {
    LC_REQUIRE(h_id && out && bytes == (int)sizeof(ncclUniqueId) && world >= 1 && rank >= 0 && rank < world, LC_E_BADARG,
               "lc_comm_create: bad argument");
    if (int rc = need_api("lc_comm_create")) return rc;
    // the communicator belongs to `device`; the calling thread's current device is put back afterwards (the
    // call used to leave it changed)
    int before = -1;
    LC_HIP(hipGetDevice(&before));
    LC_HIP(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t r = api().CommInitRank(&c, world, id, rank);
    if (before >= 0 && before != device) (void)hipSetDevice(before);
    if (r != ncclSuccess) return lc::fail(LC_E_HIP, "ncclCommInitRank: %s", api().GetErrorString(r));
    *out = new lc_comm{c, rank, world};
    return LC_OK;
}

extern "C" int lc_comm_destroy(lc_comm_t* comm) {
    LC_REQUIRE(comm, LC_E_BADARG, "lc_comm_destroy: null communicator");
    if (api().ok) (void)api().CommDestroy(comm->comm);
    delete comm;
    return LC_OK;
}

// recv (world x bytes) = every rank's send (bytes), rank-major; any element type (the wire moves bytes)
extern "C" int lc_allgather_bytes(lc_comm_t* comm, const void* d_send, void* d_recv, int64_t bytes, lc_stream_t stream) {
    LC_REQUIRE(comm && d_send && d_recv && bytes >= 0, LC_E_BADARG, "lc_allgather: bad argument");
    if (bytes == 0) return LC_OK;
    LC_RCCL(api().AllGather(d_send, d_recv, (size_t)bytes, ncclUint8, comm->comm, lc::as_stream(stream)));
    return LC_OK;
}

extern "C" int lc_allgather_f32(lc_comm_t* comm, const float* d_send, float* d_recv, int64_t count, lc_stream_t stream) {
    LC_REQUIRE(comm && d_send && d_recv && count >= 0, LC_E_BADARG, "lc_allgather_f32: bad argument");
    if (count == 0) return LC_OK;
    LC_RCCL(api().AllGather(d_send, d_recv, (size_t)count, ncclFloat32, comm->comm, lc::as_stream(stream)));
    return LC_OK;
}

// in place: d_buf = sum (op 0) or max (op 1) over the ranks; dtype LC_F32 / LC_F64 / LC_I32
extern "C" int lc_allreduce(lc_comm_t* comm, void* d_buf, int64_t count, int dtype, int op, lc_stream_t stream) {
    LC_REQUIRE(comm && d_buf && count >= 0 && (op == 0 || op == 1), LC_E_BADARG, "lc_allreduce: bad argument");
    ncclDataType_t t;
    if (dtype == LC_F32) t = ncclFloat32;
    else if (dtype == LC_F64) t = ncclFloat64;
    else if (dtype == LC_I32) t = ncclInt32;
    else return lc::fail(LC_E_BADARG, "lc_allreduce: dtype %d unsupported", dtype);
    if (count == 0) return LC_OK;
    LC_RCCL(api().AllReduce(d_buf, d_buf, (size_t)count, t, op == 0 ? ncclSum : ncclMax, comm->comm, lc::as_stream(stream)));
    return LC_OK;
}

extern "C" int lc_allreduce_sum_f32(lc_comm_t* comm, float* d_buf, int64_t count, lc_stream_t stream) {
    return lc_allreduce(comm, d_buf, count, LC_F32, 0, stream);
}
#endif  // LC_HAVE_RCCL_HEADER
