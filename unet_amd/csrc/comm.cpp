// RCCL entry points of the C ABI (include/unet_hip.h): what a host WITHOUT torch.distributed needs for the data-parallel
// path of the reference (train.py:33 init_process_group("nccl"), :143 DDP wrap = rank-0 broadcast of parameters and
// buffers + per-bucket gradient all-reduce).  The Python mirror (unet_amd/parallel.py) goes through torch.distributed,
// whose "nccl" backend is the same RCCL.
//
// RCCL is resolved at run time (dlopen of librccl.so.1): libunet_hip.so keeps no link-time dependency on it, and inside a
// PyTorch process the RCCL instance torch has already loaded is the one that gets used (same SONAME), never a second copy.
// The library holds no communicator of its own: the handle unet_comm_init returns belongs to the caller.
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>          // types and enums only; every function is called through the table below

void unet_set_error(const char* fmt, ...);

namespace {
struct Api {
    void* handle = nullptr;
    ncclResult_t (*get_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*err_string)(ncclResult_t) = nullptr;
    bool ok = false;
};

template <typename F> bool sym(void* h, const char* name, F& f) {
    f = (F)dlsym(h, name);
    return f != nullptr;
}

Api load_api() {
    Api a;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (a.handle) break;
    }
    if (!a.handle) return a;
    a.ok = sym(a.handle, "ncclGetUniqueId", a.get_id) && sym(a.handle, "ncclCommInitRank", a.init_rank) &&
           sym(a.handle, "ncclCommDestroy", a.destroy) && sym(a.handle, "ncclAllReduce", a.all_reduce) &&
           sym(a.handle, "ncclBroadcast", a.broadcast) && sym(a.handle, "ncclGetErrorString", a.err_string);
    return a;
}

Api* api() {
    static Api a = load_api();          // C++11: initialised once, thread-safe
    if (!a.ok) {
        unet_set_error("RCCL is not available: %s", a.handle ? "librccl lacks an expected symbol" : "dlopen(librccl.so.1) failed");
        return nullptr;
    }
    return &a;
}

int fail(Api* a, const char* what, ncclResult_t r) {
    unet_set_error("%s: %s", what, a->err_string(r));
    return -3;
}

bool to_nccl(int dtype, ncclDataType_t* t) {
    if (dtype == 0) { *t = ncclFloat32; return true; }       // UNET_F32
    if (dtype == 1) { *t = ncclBfloat16; return true; }      // UNET_BF16
    return false;
}
}  // namespace

extern "C" int unet_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "the ABI documents 128 bytes");
    Api* a = api();
    if (!a) return -1;
    if (!id128) { unet_set_error("comm_unique_id: null argument"); return -1; }
    ncclUniqueId id;
    const ncclResult_t r = a->get_id(&id);
    if (r != ncclSuccess) return fail(a, "ncclGetUniqueId", r);
    memcpy(id128, &id, sizeof(id));
    return 0;
}

extern "C" int unet_comm_init(void** comm, int world, int rank, const void* id128) {
    Api* a = api();
    if (!a) return -1;
    if (!comm || !id128 || world < 1 || rank < 0 || rank >= world) { unet_set_error("comm_init: bad arguments (world %d, rank %d)", world, rank); return -1; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t r = a->init_rank(&c, world, id, rank);
    if (r != ncclSuccess) return fail(a, "ncclCommInitRank", r);
    *comm = (void*)c;
    return 0;
}

extern "C" int unet_comm_destroy(void* comm) {
    Api* a = api();
    if (!a) return -1;
    if (!comm) return 0;
    const ncclResult_t r = a->destroy((ncclComm_t)comm);
    return r == ncclSuccess ? 0 : fail(a, "ncclCommDestroy", r);
}

extern "C" int unet_allreduce_bucket(void* comm, void* buf, long long count, int dtype, void* stream) {
    Api* a = api();
    if (!a) return -1;
    ncclDataType_t t;
    if (!comm || (!buf && count > 0) || count < 0 || !to_nccl(dtype, &t)) { unet_set_error("allreduce_bucket: bad arguments"); return -1; }
    if (count == 0) return 0;
    const ncclResult_t r = a->all_reduce(buf, buf, (size_t)count, t, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : fail(a, "ncclAllReduce", r);
}

extern "C" int unet_broadcast(void* comm, void* buf, long long count, int dtype, int root, void* stream) {
    Api* a = api();
    if (!a) return -1;
    ncclDataType_t t;
    if (!comm || (!buf && count > 0) || count < 0 || root < 0 || !to_nccl(dtype, &t)) { unet_set_error("broadcast: bad arguments"); return -1; }
    if (count == 0) return 0;
    const ncclResult_t r = a->broadcast(buf, buf, (size_t)count, t, root, (ncclComm_t)comm, (hipStream_t)stream);
    return r == ncclSuccess ? 0 : fail(a, "ncclBroadcast", r);
}
