// The collective of the multi-GPU exchange, issued by the library itself (BASELINE north_star: "RCCL all-gather over xGMI of per-GPU
// top-K matches"; the caller shape is the reference's own C++ driver, linemodLevelup/test.cpp:111-130, which has no Python around it).
// librccl.so is loaded at run time (dlopen), so the library neither links RCCL nor needs it for single-GPU use:
//   lm_comm_unique_id       rank 0 makes the 128-byte id (ncclGetUniqueId); the caller carries it to the other ranks (MPI, a file, a socket,
//                           torch.distributed's store — the library has no transport of its own)
//   lm_comm_create          ncclCommInitRank on this rank's device
//   lm_exchange_allgather   ncclAllGather of equal byte blocks on the detector's exchange stream, in stream order with the pack / merge kernels
//   lm_detector_exchange_group   pack + all-gather + merge of a group of frames: what sharded.DeviceExchange._exchange_group does, in one call
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "detector_internal.h"

namespace {

// the part of rccl.h this file uses (ABI of NCCL 2.x / RCCL: an opaque communicator pointer, a 128-byte id, int enums)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[128]; };
constexpr int kNcclUint8 = 1;      // ncclDataType_t: ncclInt8 = 0, ncclUint8 = 1
constexpr int kNcclSuccess = 0;

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    char why[256] = {0};
};

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("LM_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !n[0]) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            snprintf(r.why, sizeof(r.why), "%s", dlerror());
        }
        if (!r.lib) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather) {
            snprintf(r.why, sizeof(r.why), "librccl.so lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
            dlclose(r.lib);
            r.lib = nullptr;
        }
    });
    return &r;
}

int rccl_error(const char* what, int rc) {
    Rccl* r = rccl();
    return lm_set_error(LM_ERR_HIP, "%s: RCCL error %d (%s)", what, rc, r->GetErrorString ? r->GetErrorString(rc) : "?");
}

}  // namespace

struct lm_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
};

extern "C" int lm_comm_available(void) { return rccl()->lib ? 1 : 0; }

extern "C" int lm_comm_unique_id(void* id128) {
    if (!id128) return lm_set_error(LM_ERR_INVALID, "null argument");
    Rccl* r = rccl();
    if (!r->lib) return lm_set_error(LM_ERR_NO_DEVICE, "RCCL is not available: %s", r->why);
    ncclUniqueId id;
    const int rc = r->GetUniqueId(&id);
    if (rc != kNcclSuccess) return rccl_error("ncclGetUniqueId", rc);
    memcpy(id128, &id, sizeof(id));
    return LM_OK;
}

extern "C" int lm_comm_create(const void* id128, int rank, int world, int device, lm_comm** out) {
    if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return lm_set_error(LM_ERR_INVALID, "bad argument (rank %d of %d)", rank, world);
    *out = nullptr;
    Rccl* r = rccl();
    if (!r->lib) return lm_set_error(LM_ERR_NO_DEVICE, "RCCL is not available: %s", r->why);
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    lm_comm* c = new lm_comm();
    c->rank = rank; c->world = world; c->device = device;
    const int rc = r->CommInitRank(&c->comm, world, id, rank);
    if (rc != kNcclSuccess) { delete c; return rccl_error("ncclCommInitRank", rc); }
    *out = c;
    return LM_OK;
}

extern "C" void lm_comm_destroy(lm_comm* c) {
    if (!c) return;
    if (c->comm && rccl()->lib) (void)rccl()->CommDestroy(c->comm);
    delete c;
}

extern "C" int lm_comm_rank(const lm_comm* c) { return c ? c->rank : -1; }
extern "C" int lm_comm_world(const lm_comm* c) { return c ? c->world : 0; }

extern "C" int lm_exchange_allgather(lm_detector* d, lm_comm* c, const void* send, void* recv, size_t bytes_per_rank) {
    if (!d || !c || !send || !recv) return lm_set_error(LM_ERR_INVALID, "null argument");
    if (c->device != d->device) return lm_set_error(LM_ERR_INVALID, "the communicator lives on device %d, the detector on %d", c->device, d->device);
    hipStream_t s = (hipStream_t)lm_detector_exchange_stream(d);
    if (!s) return LM_ERR_HIP;
    const int rc = rccl()->AllGather(send, recv, bytes_per_rank, kNcclUint8, c->comm, s);
    if (rc != kNcclSuccess) return rccl_error("ncclAllGather", rc);
    return LM_OK;
}

extern "C" int lm_detector_exchange_group(lm_detector* d, lm_comm* c, uint64_t first, int n, void* send_blocks, void* recv_blocks, int capacity) {
    if (!d || !c) return lm_set_error(LM_ERR_INVALID, "null argument");
    int rc = lm_detector_exchange_pack_group(d, first, n, send_blocks, capacity);
    if (rc) return rc;
    const size_t bytes = lm_exchange_block_bytes(capacity) * (size_t)n;
    if ((rc = lm_exchange_allgather(d, c, send_blocks, recv_blocks, bytes))) return rc;
    return lm_detector_exchange_merge_group(d, first, n, recv_blocks, c->world, capacity);
}
