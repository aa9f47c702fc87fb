// gs_api_comm.hip -- RCCL below the boundary, for hosts without torch.distributed (ONE all-reduce of the flat gradient
// buffer per step; the reference has no collective call site).  librccl is loaded with dlopen on first use.
#include "gs_ctx.h"

#include <dlfcn.h>

namespace {

// RCCL entry points, resolved from librccl.so.1 on first use (the same copy torch loaded, if any)
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (h) return true;
        h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { err = std::string("cannot load librccl.so.1: ") + dlerror(); return false; }
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(h, "ncclAllReduce"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllReduce || !CommDestroy) { err = "librccl: missing symbols"; h = nullptr; return false; }
        return true;
    }
} g_rccl;

}  // namespace

void comm_release(gs_ctx *c) {
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    c->comm = nullptr; c->comm_ranks = 0;
}

extern "C" {

int gs_comm_unique_id(void *id128) {
    if (!id128) return GS_ERR_INVALID;
    std::string err;
    if (!g_rccl.load(err)) return fail(nullptr, GS_ERR_UNSUPPORTED, err);
    static_assert(sizeof(ncclUniqueId) == GS_COMM_ID_BYTES, "ncclUniqueId must be 128 bytes");
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, GS_ERR_HIP, std::string("ncclGetUniqueId: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
    std::memcpy(id128, &id, sizeof(id));
    return GS_OK;
}

int gs_comm_init(gs_ctx *c, int rank, int nranks, const void *id128) {
    if (!c || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return GS_ERR_INVALID;
    std::string err;
    if (!g_rccl.load(err)) return fail(c, GS_ERR_UNSUPPORTED, err);
    if (bind_device(c)) return GS_ERR_HIP;
    if (c->comm) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) { c->comm = nullptr; return fail(c, GS_ERR_HIP, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error")); }
    c->comm_ranks = nranks;
    return GS_OK;
}

int gs_allreduce_grads(gs_ctx *c, const gs_grads *g) {
    if (!c || !g) return GS_ERR_INVALID;
    if (!c->comm) return fail(c, GS_ERR_INVALID, "gs_allreduce_grads: gs_comm_init first");
    if (bind_device(c)) return GS_ERR_HIP;
    const size_t n = (size_t)c->n;
    float *p[5] = {g->d_means, g->d_scales, g->d_quats, g->d_opacities, g->d_shs};
    const size_t w[5] = {c->width[0] * n, c->width[1] * n, c->width[2] * n, c->width[3] * n, c->width[4] * n};
    bool flat = p[0] != nullptr;
    for (int i = 0; i + 1 < 5 && flat; ++i) flat = p[i + 1] == p[i] + w[i];
    auto reduce = [&](float *buf, size_t count) -> int {
        if (!buf || !count) return GS_OK;
        const ncclResult_t r = g_rccl.AllReduce(buf, buf, count, ncclFloat, ncclSum, c->comm, c->stream);
        if (r != ncclSuccess) return fail(c, GS_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        return GS_OK;
    };
    if (flat) return reduce(p[0], w[0] + w[1] + w[2] + w[3] + w[4]);            // ONE collective (59 N floats at SH3)
    for (int i = 0; i < 5; ++i) if (int rc = reduce(p[i], w[i])) return rc;
    return GS_OK;
}

int gs_comm_destroy(gs_ctx *c) {
    if (!c) return GS_ERR_INVALID;
    comm_release(c);
    return GS_OK;
}

}  // extern "C"
