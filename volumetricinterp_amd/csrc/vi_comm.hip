// Multi-GPU plumbing of libvinterp.so: one RCCL broadcast of the shared parameters (beam geometry,
// regularisation matrices, hull facets) from rank 0 over xGMI.  The fit/evaluate path itself has NO collective:
// records are independent (volumetricinterp/interpolate.py:511 carries no state between iterations) and are
// sharded by the caller.  RCCL is loaded lazily (dlopen) so that single-GPU use has no dependency on it; the
// unique id travels over the caller's own control channel (volumetricinterp_amd/parallel.py).
#include "vi_common.h"

#include <dlfcn.h>

namespace {

struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};

struct UniqueId {
    char internal[128];
};

typedef int (*init_rank_fn)(void**, int, UniqueId, int);

RcclApi g_api;
init_rank_fn g_init = nullptr;

int load_rccl()
{
    if (g_api.lib) return VI_OK;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        g_api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_api.lib) break;
    }
    if (!g_api.lib) {
        vi_set_error("RCCL not found: %s", dlerror());
        return VI_ERR_RCCL;
    }
    g_api.GetUniqueId = (int (*)(void*))dlsym(g_api.lib, "ncclGetUniqueId");
    g_init = (init_rank_fn)dlsym(g_api.lib, "ncclCommInitRank");
    g_api.Broadcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(g_api.lib, "ncclBroadcast");
    g_api.CommDestroy = (int (*)(void*))dlsym(g_api.lib, "ncclCommDestroy");
    g_api.GetErrorString = (const char* (*)(int))dlsym(g_api.lib, "ncclGetErrorString");
    if (!g_api.GetUniqueId || !g_init || !g_api.Broadcast || !g_api.CommDestroy) {
        vi_set_error("RCCL symbols missing");
        return VI_ERR_RCCL;
    }
    return VI_OK;
}

int rccl_check(int rc, const char* what)
{
    if (rc == 0) return VI_OK;
    vi_set_error("%s failed: %s", what, g_api.GetErrorString ? g_api.GetErrorString(rc) : "RCCL error");
    return VI_ERR_RCCL;
}

}  // namespace

extern "C" int vi_rccl_unique_id(char* out128)
{
    VI_REQUIRE(out128, "null argument");
    int rc = load_rccl();
    if (rc != VI_OK) return rc;
    UniqueId id;
    memset(&id, 0, sizeof(id));
    rc = rccl_check(g_api.GetUniqueId(&id), "ncclGetUniqueId");
    if (rc != VI_OK) return rc;
    memcpy(out128, id.internal, 128);
    return VI_OK;
}

extern "C" int vi_rccl_init(vi_ctx* c, int nranks, int rank, const char* id128)
{
    VI_REQUIRE(c && id128, "null argument");
    VI_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "bad rank");
    int rc = load_rccl();
    if (rc != VI_OK) return rc;
    VI_HIP(hipSetDevice(c->device));
    UniqueId id;
    memcpy(id.internal, id128, 128);
    void* comm = nullptr;
    rc = rccl_check(g_init(&comm, nranks, id, rank), "ncclCommInitRank");
    if (rc != VI_OK) return rc;
    c->rccl_comm = comm;
    return VI_OK;
}

// ncclFloat64 = 8 in nccl.h's ncclDataType_t
extern "C" int vi_rccl_bcast_f64(vi_ctx* c, double* d_buf, int64_t count, int root)
{
    VI_REQUIRE(c && c->rccl_comm, "RCCL communicator not initialised");
    VI_REQUIRE(count >= 0 && (count == 0 || d_buf), "bad buffer");
    if (count == 0) return VI_OK;
    VI_HIP(hipSetDevice(c->device));
    int rc = rccl_check(g_api.Broadcast(d_buf, d_buf, (size_t)count, 8, root, c->rccl_comm, c->stream), "ncclBroadcast");
    if (rc != VI_OK) return rc;
    VI_HIP(hipStreamSynchronize(c->stream));
    return VI_OK;
}

extern "C" int vi_rccl_destroy(vi_ctx* c)
{
    VI_REQUIRE(c, "null context");
    if (c->rccl_comm && g_api.CommDestroy) {
        (void)hipSetDevice(c->device);
        g_api.CommDestroy(c->rccl_comm);
    }
    c->rccl_comm = nullptr;
    return VI_OK;
}
