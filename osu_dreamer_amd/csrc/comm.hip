// Data-parallel gradient exchange through the C ABI: RCCL all-reduce / broadcast over xGMI on a
// caller-supplied communicator and stream (SURVEY.md section 8(b) `od_allreduce_grads`, 8(e)).
//
// The reference trains on one device (models/diffusion/model.yml:11); a data-parallel run of
// DiffusionTrainer.training_step (train.py:120-123) needs exactly one exchange per step: the mean of
// the 46.9 M fp32 parameter gradients.  RCCL is bound at run time (dlopen of the path the host names,
// default librccl.so.1) so the library has no link-time dependency on it and a process that already
// holds an RCCL instance (PyTorch's) shares it instead of loading a second copy.
#include "od_common.h"
#include "od_api_internal.h"

#if defined(OD_EMU)
// the CPU test build has no RCCL: the entry points exist (the ABI is complete) and refuse
extern "C" int od_comm_load(const char*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_comm_version(void) { return 0; }
extern "C" int od_comm_unique_id(void*, int) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_comm_init(void**, int, int, const void*, int) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_comm_destroy(void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_comm_abort(void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_comm_count(void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_allreduce_grads(void*, float*, long, int, void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_broadcast_f32(void*, float*, long, int, void*) { return OD_ERR_UNSUPPORTED; }
extern "C" int od_comm_ring_standin(float*, long, int, int, int, void*) { return OD_ERR_UNSUPPORTED; }
#else
#include <dlfcn.h>
#include <string.h>

namespace {

// the subset of rccl.h this file uses, restated so the build does not depend on a particular RCCL header
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;                        // ncclSuccess == 0
enum { kNcclFloat32 = 7 };                       // ncclDataType_t::ncclFloat32
enum { kNcclSum = 0, kNcclAvg = 4 };             // ncclRedOp_t

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
};
Rccl g_rccl;

template <class F> bool bind(F& f, const char* name) {
    f = (F)dlsym(g_rccl.handle, name);
    return f != nullptr;
}

int load(const char* path) {
    if (g_rccl.handle) return 0;
    const char* p = (path && path[0]) ? path : "librccl.so.1";
    g_rccl.handle = dlopen(p, RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.handle) return OD_ERR_COMM;
    const bool ok = bind(g_rccl.GetVersion, "ncclGetVersion") && bind(g_rccl.GetUniqueId, "ncclGetUniqueId") &&
                    bind(g_rccl.CommInitRank, "ncclCommInitRank") && bind(g_rccl.CommDestroy, "ncclCommDestroy") &&
                    bind(g_rccl.CommAbort, "ncclCommAbort") && bind(g_rccl.CommCount, "ncclCommCount") &&
                    bind(g_rccl.AllReduce, "ncclAllReduce") && bind(g_rccl.Broadcast, "ncclBroadcast");
    if (!ok) { dlclose(g_rccl.handle); g_rccl = Rccl(); return OD_ERR_COMM; }
    return 0;
}

}  // namespace

extern "C" int od_comm_load(const char* path) { return load(path); }

extern "C" int od_comm_version(void) {
    if (load(nullptr)) return 0;
    int v = 0;
    return g_rccl.GetVersion(&v) == 0 ? v : 0;
}

extern "C" int od_comm_unique_id(void* out, int nbytes) {
    if (nbytes != (int)sizeof(ncclUniqueId)) return OD_ERR_ARG;
    if (int rc = load(nullptr)) return rc;
    ncclUniqueId id;
    if (g_rccl.GetUniqueId(&id) != 0) return OD_ERR_COMM;
    memcpy(out, &id, sizeof(id));
    return 0;
}

extern "C" int od_comm_init(void** comm_out, int nranks, int rank, const void* unique_id, int nbytes) {
    if (!comm_out || nranks < 1 || rank < 0 || rank >= nranks || nbytes != (int)sizeof(ncclUniqueId)) return OD_ERR_ARG;
    if (int rc = load(nullptr)) return rc;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t c = nullptr;
    if (g_rccl.CommInitRank(&c, nranks, id, rank) != 0) return OD_ERR_COMM;
    *comm_out = (void*)c;
    return 0;
}

extern "C" int od_comm_destroy(void* comm) {
    if (!comm || !g_rccl.handle) return OD_ERR_ARG;
    return g_rccl.CommDestroy((ncclComm_t)comm) == 0 ? 0 : OD_ERR_COMM;
}

// failure path: frees the communicator without waiting for outstanding collectives (a peer that died never completes them)
extern "C" int od_comm_abort(void* comm) {
    if (!comm || !g_rccl.handle) return OD_ERR_ARG;
    return g_rccl.CommAbort((ncclComm_t)comm) == 0 ? 0 : OD_ERR_COMM;
}

// ranks RCCL itself counts in the communicator (>= 1), or a negative error
extern "C" int od_comm_count(void* comm) {
    if (!comm || !g_rccl.handle) return OD_ERR_ARG;
    int n = 0;
    return g_rccl.CommCount((ncclComm_t)comm, &n) == 0 ? n : OD_ERR_COMM;
}

extern "C" int od_allreduce_grads(void* comm, float* grads, long count, int average, void* stream) {
    if (!comm || !g_rccl.handle || count < 0) return OD_ERR_ARG;
    if (count == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (g_rccl.AllReduce(grads, grads, (size_t)count, kNcclFloat32, average ? kNcclAvg : kNcclSum, (ncclComm_t)comm, st) != 0)
        return OD_ERR_COMM;
    return 0;
}

extern "C" int od_broadcast_f32(void* comm, float* buf, long count, int root, void* stream) {
    if (!comm || !g_rccl.handle || count < 0) return OD_ERR_ARG;
    if (count == 0) return 0;
    if (g_rccl.Broadcast(buf, buf, (size_t)count, kNcclFloat32, root, (ncclComm_t)comm, (hipStream_t)stream) != 0) return OD_ERR_COMM;
    return 0;
}

// What a ring all-reduce looks like to the kernels it runs beside, for boxes with ONE GPU (a single-rank ncclAllReduce in place launches
// nothing): `channels` long-lived workgroups of `threads` threads, each streaming its slice of `buf` (read + write back unchanged,
// 16 bytes per lane) `rounds` times — resident on `channels` CUs for the whole exchange and drawing HBM bandwidth, like RCCL's channel
// kernels.  The soak of the fused attention backward (tools/soak_fused.py, tests/test_ddp_rccl.py) runs it on the exchange's side stream.
namespace {
__global__ void ring_standin_kernel(float* buf, long n4, int rounds) {
    const long per = (n4 + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < n4 ? lo + per : n4;
    f32x4* b4 = (f32x4*)buf;
    for (int r = 0; r < rounds; r++)
        for (long i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            f32x4 v = __builtin_nontemporal_load(b4 + i);
            asm volatile("" : "+v"(v));
            __builtin_nontemporal_store(v, b4 + i);
        }
}
}  // namespace
extern "C" int od_comm_ring_standin(float* buf, long count, int channels, int threads, int rounds, void* stream) {
    if (!buf || count < 4 || channels < 1 || channels > 1024 || rounds < 1 || (threads != 256 && threads != 512 && threads != 1024)) return OD_ERR_ARG;
    hipLaunchKernelGGL(ring_standin_kernel, dim3(channels), dim3(threads), 0, (hipStream_t)stream, buf, count / 4, rounds);
    OD_CHECK_LAUNCH();
    return 0;
}
#endif
