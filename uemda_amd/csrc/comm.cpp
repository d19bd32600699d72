// Data-parallel collective of the path behind the C ABI: all-reduce(sum) of the flat fp32 gradient arena over RCCL
// (xGMI inside a node) on the CALLER's stream.  New relative to the reference (single GPU); SURVEY 8(b) lists
// `allreduce_flat` in the boundary's op set.  RCCL is resolved at run time from the process (a PyTorch-ROCm host already
// carries librccl.so: a second copy linked in here would be a second, unrelated communicator library) and only falls
// back to loading /opt/rocm's librccl when the host process has none.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

namespace {
struct Rccl {
    void* h = nullptr;
    int (*get_unique_id)(void*) = nullptr;
    void* init_rank = nullptr;          // ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int): called through a typed thunk
    int (*all_reduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*destroy)(void*) = nullptr;
    const char* (*err)(int) = nullptr;
};
struct id128 { char b[128]; };
Rccl g;

int load_rccl() {
    if (g.h) return UEM_OK;
    void* h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);             // the copy the host process already uses
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return uem_fail(UEM_ERR_UNSUPPORTED, "comm: librccl.so not found (%s)", dlerror());
    g.get_unique_id = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g.init_rank = dlsym(h, "ncclCommInitRank");
    g.all_reduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g.destroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g.err = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g.get_unique_id || !g.init_rank || !g.all_reduce || !g.destroy)
        return uem_fail(UEM_ERR_UNSUPPORTED, "comm: librccl.so lacks the collective entry points");
    g.h = h;
    return UEM_OK;
}
int rccl_fail(const char* what, int rc) {
    return uem_fail(UEM_ERR_LAUNCH, "%s: RCCL error %d (%s)", what, rc, g.err ? g.err(rc) : "?");
}
}  // namespace

extern "C" int uem_comm_unique_id(void* id_out_128_bytes) {
    UEM_REQUIRE(id_out_128_bytes, "comm_unique_id: null pointer");
    int rc = load_rccl();
    if (rc) return rc;
    rc = g.get_unique_id(id_out_128_bytes);
    return rc ? rccl_fail("ncclGetUniqueId", rc) : UEM_OK;
}
extern "C" int uem_comm_init(void** comm_out, const void* id_128_bytes, int rank, int world) {
    UEM_REQUIRE(comm_out && id_128_bytes && world >= 1 && rank >= 0 && rank < world, "comm_init: bad arguments");
    int rc = load_rccl();
    if (rc) return rc;
    id128 id;
    memcpy(&id, id_128_bytes, sizeof(id));
    typedef int (*init_fn)(void**, int, id128, int);
    rc = ((init_fn)g.init_rank)(comm_out, world, id, rank);
    return rc ? rccl_fail("ncclCommInitRank", rc) : UEM_OK;
}
extern "C" int uem_allreduce_flat(void* comm, float* buf, int64_t count, void* stream) {
    UEM_REQUIRE(comm && buf && count > 0, "allreduce_flat: bad arguments");
    UEM_REQUIRE(g.h, "allreduce_flat: no communicator library loaded (uem_comm_init first)");
    const int rc = g.all_reduce(buf, buf, (size_t)count, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t)stream);
    return rc ? rccl_fail("ncclAllReduce", rc) : UEM_OK;
}
extern "C" int uem_comm_destroy(void* comm) {
    if (!comm || !g.h) return UEM_OK;
    const int rc = g.destroy(comm);
    return rc ? rccl_fail("ncclCommDestroy", rc) : UEM_OK;
}
