// libnepmi355: the one collective of the hot path -- Beyn's quadrature nodes sharded over the GPUs of a node
// (src/method_contour_common.jl:46,61-94 `integrate_interval` seam, src/method_beyncontour.jl:89-104): every rank
// accumulates the moments of ITS nodes, then ONE all-gather of the 2 n k partial block over xGMI (RCCL) and a sum in
// fixed rank order, so that every rank holds bit-identical A0, A1 independent of arrival order.
//
// RCCL is loaded on first use (dlopen; the copy the host process already has -- torch's -- is preferred), so the rest of
// the library has no link-time dependency on it.  One process per GPU; the 128-byte unique id travels out of band (the
// host's own launcher: MPI.jl bcast, torch.distributed, a shared file).
#include "common.h"
#include <dlfcn.h>
#include <mutex>

namespace {
struct nccl_uid { char internal[128]; };
typedef void* nccl_comm_t;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(nccl_comm_t*, int, nccl_uid, int);
typedef int (*fn_destroy)(nccl_comm_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, nccl_comm_t, hipStream_t);
typedef const char* (*fn_errstr)(int);
struct RcclApi {
    void* lib = nullptr; bool tried = false;
    fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr; fn_errstr errstr = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;
const int NCCL_DOUBLE = 8;      // ncclFloat64 (rccl.h)

int rccl_ready() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.allgather) return NEP_OK;
    if (g_rccl.tried) { nep_set_error("RCCL is not available (librccl.so could not be loaded)"); return NEP_ERR_HIP; }
    g_rccl.tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) { g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); if (g_rccl.lib) break; }   // already mapped (torch)?
    if (!g_rccl.lib)
        for (const char* nm : names) { g_rccl.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL); if (g_rccl.lib) break; }
    if (!g_rccl.lib) { nep_set_error("dlopen(librccl.so): %s", dlerror()); return NEP_ERR_HIP; }
    g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.allgather = (fn_allgather)dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.errstr = (fn_errstr)dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allgather) {
        g_rccl.allgather = nullptr;
        nep_set_error("RCCL symbols missing");
        return NEP_ERR_HIP;
    }
    return NEP_OK;
}
int rccl_fail(const char* what, int st) {
    nep_set_error("%s failed: %s (status %d)", what, g_rccl.errstr ? g_rccl.errstr(st) : "?", st);
    return NEP_ERR_HIP;
}
}  // namespace

struct nep_comm {
    nccl_comm_t comm = nullptr;
    int rank = 0, world = 1;
    NepScratch gather;      // world x len
};

// total[i] = sum_r parts[r*len + i], r = 0..world-1 in this order on every rank
__global__ void k_sum_ranks(int64_t len, int world, const cplx* parts, cplx* total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
        cplx s = parts[i];
        for (int r = 1; r < world; ++r) { const cplx v = parts[(int64_t)r * len + i]; s.x += v.x; s.y += v.y; }
        total[i] = s;
    }
}

extern "C" {

int32_t nep_comm_unique_id(void* h_out128) {
    ARGCHK(h_out128 != nullptr);
    int rc = rccl_ready();
    if (rc) return rc;
    nccl_uid id;
    const int st = g_rccl.get_uid(&id);
    if (st != 0) return rccl_fail("ncclGetUniqueId", st);
    memcpy(h_out128, id.internal, 128);
    return NEP_OK;
}

int32_t nep_comm_create(int32_t rank, int32_t world, const void* h_unique_id128, nep_comm** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(world >= 1 && rank >= 0 && rank < world && h_unique_id128 != nullptr);
    int rc = rccl_ready();
    if (rc) return rc;
    nccl_uid id;
    memcpy(id.internal, h_unique_id128, 128);
    nep_comm* c = new nep_comm();
    c->rank = rank; c->world = world;
    const int st = g_rccl.init_rank(&c->comm, world, id, rank);     // binds to the calling thread's current device
    if (st != 0) { delete c; return rccl_fail("ncclCommInitRank", st); }
    *out = c;
    return NEP_OK;
}

int32_t nep_comm_destroy(nep_comm* c) {
    if (!c) return NEP_OK;
    if (c->comm) (void)g_rccl.destroy(c->comm);
    c->gather.release();
    delete c;
    return NEP_OK;
}

int32_t nep_comm_info(const nep_comm* c, int32_t out[2]) {
    ARGCHK(c && out);
    out[0] = c->rank; out[1] = c->world;
    return NEP_OK;
}

// the second half of nep_allgather_sum on a gather buffer the caller provides (world x len, rank r's block at r * len):
// total[i] = parts[0][i] + parts[1][i] + ... in THIS order -- what every rank executes after the all-gather, so that all of
// them hold the same bits.  d_total may alias block 0 of d_parts.  Exposed so that the reduction can be tested (and used by
// hosts that move the blocks themselves, e.g. ranks sharing one GPU) without a multi-rank RCCL communicator.
int32_t nep_sum_ranks(const nep_cdouble* d_parts, int64_t len, int32_t world, nep_cdouble* d_total, nep_stream stream) {
    ARGCHK(d_parts && d_total && len > 0 && world >= 1);
    hipStream_t st = as_stream(stream);
    const int g = (int)std::min<int64_t>((len + 255) / 256, 4096);
    hipLaunchKernelGGL(k_sum_ranks, dim3(g), dim3(256), 0, st, len, (int)world, (const cplx*)d_parts, (cplx*)d_total);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_allgather_sum(nep_comm* c, const nep_cdouble* d_partial, int64_t len, nep_cdouble* d_total, nep_stream stream) {
    ARGCHK(c && d_partial && d_total && len > 0);
    hipStream_t st = as_stream(stream);
    int rc = c->gather.ensure((size_t)c->world * len * sizeof(cplx));
    if (rc) return rc;
    const int s = g_rccl.allgather(d_partial, c->gather.dptr, (size_t)2 * len, NCCL_DOUBLE, c->comm, st);
    if (s != 0) return rccl_fail("ncclAllGather", s);
    return nep_sum_ranks((const nep_cdouble*)c->gather.dptr, len, c->world, d_total, stream);
}

}  // extern "C"
