// libnepmi355: error handling, device memory, BLAS-1 style helpers (gfx950).
#include "common.h"
#include <vector>
#include <algorithm>
#include <math.h>
#include <time.h>

static thread_local char g_err[1024] = "";

void nep_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int NepScratch::ensure(size_t bytes) {
    if (bytes <= cap) return NEP_OK;
    // Growing hands the old block back to the shared pool.  Kernels that were ENQUEUED with it may not have run yet, and the
    // pool serves other streams and host threads (iar's convergence checks run on their own stream next to a recurrence that
    // is enqueued tens of milliseconds ahead of the device): wait for the device before the block can change hands.  Rare
    // by construction: capacities start at 4 MiB and double.
    if (dptr) { (void)hipDeviceSynchronize(); nep_pool_free(dptr); dptr = nullptr; cap = 0; }
    size_t want = std::max<size_t>(2 * bytes, (size_t)4 << 20);
    int rc = nep_pool_alloc(&dptr, want);
    if (rc) return rc;
    cap = want;
    return NEP_OK;
}
void NepScratch::release() {
    if (dptr) nep_pool_free(dptr);
    dptr = nullptr; cap = 0;
}

// ---- caching device allocator ---------------------------------------------------------------------
#include <map>
#include <mutex>
#include <unordered_map>
namespace {
struct PoolBlock { void* p; hipEvent_t ev; };      // ev != nullptr: work that may still touch the block (nep_pool_free_on)
std::mutex g_pool_mu;
std::multimap<size_t, PoolBlock> g_pool_free;    // size -> block
std::unordered_map<void*, size_t> g_pool_live;   // block -> size
size_t g_pool_cached = 0;
const size_t POOL_CAP = (size_t)4 << 30;         // at most 4 GiB of idle blocks
const int POOL_GROW_BUSY = 3;                    // blocks of one size class that may be in flight before a request waits for one
size_t pool_round(size_t b) {
    if (b < 4096) return 4096;
    size_t p = 4096;
    while (p < b) p <<= 1;                        // next power of two ...
    const size_t q = p >> 3;                      // ... in steps of an eighth
    return ((b + q - 1) / q) * q;
}
void pool_put(void* p, hipEvent_t ev, std::vector<PoolBlock>& to_free) {   // g_pool_mu held
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) { to_free.push_back(PoolBlock{p, ev}); return; }
    g_pool_free.emplace(it->second, PoolBlock{p, ev});
    g_pool_cached += it->second;
    g_pool_live.erase(it);
    while (g_pool_cached > POOL_CAP && !g_pool_free.empty()) {
        auto big = std::prev(g_pool_free.end());
        to_free.push_back(big->second);
        g_pool_cached -= big->first;
        g_pool_free.erase(big);
    }
}
void pool_drop(std::vector<PoolBlock>& v) {
    for (PoolBlock& b : v) {
        if (b.ev) { (void)hipEventSynchronize(b.ev); (void)hipEventDestroy(b.ev); }
        (void)hipFree(b.p);
    }
}
}  // namespace

int nep_pool_alloc(void** p, size_t bytes) {
    const size_t want = pool_round(bytes);
    // A cached block whose last user has finished is handed out at once.  When every candidate is still in flight (freed behind
    // a stream by nep_pool_free_on) the pool GROWS by a fresh block instead of making the host wait for one of them: a host that
    // waits here serialises streams that were meant to overlap (round 6: the two solve streams of contour_beyn ran one after the
    // other -- 21 instead of 13 ms for 64 node solves -- whenever the process had few work blocks of that size cached; with one block
    // per stream in flight the population settles after a call or two).  Only when the device refuses the allocation does the
    // caller wait for a busy block.
    void* busy_p = nullptr; hipEvent_t busy_ev = nullptr; size_t busy_sz = 0;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto lo = g_pool_free.lower_bound(want);
        auto pick = g_pool_free.end();
        int nbusy = 0;
        auto first_busy = g_pool_free.end();
        for (auto it = lo; it != g_pool_free.end() && it->first <= want + want / 4; ++it) {
            if (!it->second.ev || hipEventQuery(it->second.ev) == hipSuccess) { pick = it; break; }
            if (first_busy == g_pool_free.end()) first_busy = it;
            ++nbusy;
        }
        // ... but only up to POOL_GROW_BUSY blocks of a size in flight: beyond that the request waits for the oldest of them as before
        // (a population that grows with every in-flight block ran into the idle cap, and the hipFree / hipMalloc pairs that
        // followed cost more than the wait: C4 45 -> 75 ms per call)
        if (pick == g_pool_free.end() && nbusy >= POOL_GROW_BUSY) {
            busy_p = first_busy->second.p; busy_ev = first_busy->second.ev; busy_sz = first_busy->first;
            g_pool_live[busy_p] = busy_sz;
            g_pool_cached -= busy_sz;
            g_pool_free.erase(first_busy);
        }
        if (pick != g_pool_free.end()) {
            *p = pick->second.p;
            hipEvent_t done_ev = pick->second.ev;
            g_pool_live[*p] = pick->first;
            g_pool_cached -= pick->first;
            g_pool_free.erase(pick);
            if (done_ev) (void)hipEventDestroy(done_ev);
            return NEP_OK;
        }
        (void)hipGetLastError();          // (a NotReady answer of hipEventQuery is not an error of ours)
    }
    if (busy_p) {
        if (busy_ev) { (void)hipEventSynchronize(busy_ev); (void)hipEventDestroy(busy_ev); }
        *p = busy_p;
        return NEP_OK;
    }
    if (hipMalloc(p, want) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool_live[*p] = want;
        return NEP_OK;
    }
    (void)hipGetLastError();
    {   // out of device memory: a busy block of the right size, if there is one, after its users have finished
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto lo = g_pool_free.lower_bound(want);
        if (lo != g_pool_free.end() && lo->first <= want + want / 4) {
            busy_p = lo->second.p; busy_ev = lo->second.ev; busy_sz = lo->first;
            g_pool_live[busy_p] = busy_sz;
            g_pool_cached -= busy_sz;
            g_pool_free.erase(lo);
        }
    }
    if (busy_p) {
        if (busy_ev) { (void)hipEventSynchronize(busy_ev); (void)hipEventDestroy(busy_ev); }
        *p = busy_p;
        return NEP_OK;
    }
    // last resort: give the idle cache back to the device and try once more
    {
        std::vector<PoolBlock> to_free;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            for (auto& kv : g_pool_free) to_free.push_back(kv.second);
            g_pool_free.clear(); g_pool_cached = 0;
        }
        pool_drop(to_free);
    }
    HIPCHK(hipMalloc(p, want));
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[*p] = want;
    return NEP_OK;
}

void nep_pool_free(void* p) {
    if (!p) return;
    std::vector<PoolBlock> to_free;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_put(p, nullptr, to_free);
    }
    pool_drop(to_free);
}

void nep_pool_free_on(void* p, hipStream_t st, bool in_flight) {
    if (!p) return;
    hipEvent_t ev = nullptr;
    if (in_flight) {
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, st) != hipSuccess) {
            (void)hipGetLastError();
            if (ev) { (void)hipEventDestroy(ev); ev = nullptr; }
            (void)hipStreamSynchronize(st);               // no event available: be safe
        }
    }
    std::vector<PoolBlock> to_free;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_put(p, ev, to_free);
    }
    pool_drop(to_free);
}

// ---- cache of pinned host blocks -----------------------------------------------------------------------------------------------
// hipHostMalloc / hipHostFree cost 0.3-1 ms each.  The staging rings are thread_local and some host threads live for one
// driver call only (iar's checker thread: 8 slots allocated and freed per call): released slots are kept here (power-of-two
// sizes from 1 MiB, at most 256 MiB idle) and handed to the next ring that asks.
namespace {
std::mutex g_pin_mu;
std::multimap<size_t, void*> g_pin_free;
std::unordered_map<void*, size_t> g_pin_live;
size_t g_pin_cached = 0;
const size_t PIN_CAP = (size_t)256 << 20;
}  // namespace
static int pinned_alloc(void** p, size_t bytes, size_t* got) {
    size_t want = (size_t)1 << 20;
    while (want < bytes) want <<= 1;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_free.find(want);
        if (it != g_pin_free.end()) {
            *p = it->second; *got = want;
            g_pin_live[*p] = want; g_pin_cached -= want;
            g_pin_free.erase(it);
            return NEP_OK;
        }
    }
    HIPCHK(hipHostMalloc(p, want, hipHostMallocDefault));
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pin_live[*p] = want; *got = want;
    return NEP_OK;
}
static void pinned_free(void* p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        auto it = g_pin_live.find(p);
        if (it != g_pin_live.end() && g_pin_cached + it->second <= PIN_CAP) {
            g_pin_free.emplace(it->second, p); g_pin_cached += it->second;
            g_pin_live.erase(it);
            return;
        }
        if (it != g_pin_live.end()) g_pin_live.erase(it);
    }
    (void)hipHostFree(p);
}

int PinnedRing::upload(void* ddst, const void* hsrc, size_t bytes, hipStream_t st) {
    const int i = next;
    next = (next + 1) % NSLOT;
    if (!ev[i]) HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    if (used[i]) HIPCHK(hipEventSynchronize(ev[i]));     // slot's previous copy has been consumed
    if (cap[i] < bytes) {
        if (slot[i]) pinned_free(slot[i]);
        slot[i] = nullptr; cap[i] = 0;
        // pinned allocations cost ~1 ms each: sizes start at 1 MiB and double, so a growing parameter block (the B
        // fragments of iar's Ritz GEMM grow every step) does not reallocate every few calls; blocks come from / go to the cache
        size_t got = 0;
        int rcp = pinned_alloc(&slot[i], bytes, &got);
        if (rcp) return rcp;
        cap[i] = got;
    }
    memcpy(slot[i], hsrc, bytes);
    HIPCHK(hipMemcpyAsync(ddst, slot[i], bytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipEventRecord(ev[i], st));
    used[i] = true;
    return NEP_OK;
}
void PinnedRing::release() {
    for (int i = 0; i < NSLOT; ++i) {
        if (slot[i] && ev[i] && used[i]) (void)hipEventSynchronize(ev[i]);      // the copy out of the slot may still be queued
        if (slot[i]) pinned_free(slot[i]);
        if (ev[i]) (void)hipEventDestroy(ev[i]);
        slot[i] = nullptr; ev[i] = nullptr; cap[i] = 0; used[i] = false;
    }
}

extern "C" {

int32_t nep_version(void) { return 101; }
#ifndef NEP_SRC_DIGEST
#define NEP_SRC_DIGEST "unknown"
#endif
// digest of the sources this binary was built from (build.py passes it): the ctypes binding compares it with the sources it
// sees, so a stale library with changed argument lists is refused instead of corrupting memory
const char* nep_src_digest(void) { return NEP_SRC_DIGEST; }
const char* nep_last_error(void) { return g_err; }

int32_t nep_device_count(int32_t* n) {
    ARGCHK(n != nullptr);
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { c = 0; (void)hipGetLastError(); }
    *n = c;
    return NEP_OK;
}
int32_t nep_set_device(int32_t dev) { HIPCHK(hipSetDevice(dev)); return NEP_OK; }
int32_t nep_device_name(char* buf, int32_t buflen) {
    ARGCHK(buf && buflen > 0);
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, dev));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return NEP_OK;
}

int32_t nep_dev_alloc(void** dptr, size_t bytes) {
    ARGCHK(dptr != nullptr);
    HIPCHK(hipMalloc(dptr, bytes ? bytes : 16));
    return NEP_OK;
}
int32_t nep_dev_free(void* dptr) {
    if (dptr) HIPCHK(hipFree(dptr));
    return NEP_OK;
}
int32_t nep_dev_memset(void* dptr, int32_t value, size_t bytes, nep_stream stream) {
    HIPCHK(hipMemsetAsync(dptr, value, bytes, as_stream(stream)));
    return NEP_OK;
}
int32_t nep_upload(void* ddst, const void* hsrc, size_t bytes, nep_stream stream) {
    HIPCHK(hipMemcpyAsync(ddst, hsrc, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    HIPCHK(hipStreamSynchronize(as_stream(stream)));
    return NEP_OK;
}
int32_t nep_download(void* hdst, const void* dsrc, size_t bytes, nep_stream stream) {
    HIPCHK(hipMemcpyAsync(hdst, dsrc, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    HIPCHK(hipStreamSynchronize(as_stream(stream)));
    return NEP_OK;
}
int32_t nep_dev_copy(void* ddst, const void* dsrc, size_t bytes, nep_stream stream);
int32_t nep_stream_sync(nep_stream stream) {
    HIPCHK(hipStreamSynchronize(as_stream(stream)));
    return NEP_OK;
}

int nep_raise_lds(const void* kernel, int bytes) {
    static thread_local std::vector<std::pair<int, const void*>> done;
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    for (const auto& d : done) if (d.first == dev && d.second == kernel) return NEP_OK;
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.emplace_back(dev, kernel);
    return NEP_OK;
}

// Do two streams execute their kernels one after the other?  The runtime maps streams onto a small pool of hardware queues (4
// by default) and two streams that land on the same queue serialise -- a 3 ms one-wavefront kernel (csrc/hesseig.hip) then
// holds up whatever shares its queue (measured: iar's convergence checks behind the eigen-decompositions, 4 ms per batch; a
// stream created with a compute-unit mask does get a queue of its own, but the recurrence's queue then stalls while such a
// kernel runs).  There is no query for the mapping, so it is measured: a one-wavefront kernel that idles for ~0.4 ms on
// stream a, an empty kernel on stream b right behind it, and the host's clock around b's synchronisation.
__global__ void k_idle_cycles(long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_empty() {}

int32_t nep_stream_pair_serializes(nep_stream a, nep_stream b, int32_t* out) {
    ARGCHK(out != nullptr);
    *out = 0;
    hipStream_t sa = as_stream(a), sb = as_stream(b);
    if (sa == sb) { *out = 1; return NEP_OK; }
    auto now_us = []() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
    // first-use costs (code object load, clock ramp of an idle device) out of the way: a cold first launch takes milliseconds and
    // used to count as a vote for "serialises" -- with an unlucky start every candidate stream of a host was refused that way and the
    // decompositions of iar ended up on the recurrence's own hardware queue (73 instead of 32 ms per call, round 6)
    hipLaunchKernelGGL(k_idle_cycles, dim3(1), dim3(64), 0, sa, (long long)200000);
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, sa);
    hipLaunchKernelGGL(k_idle_cycles, dim3(1), dim3(64), 0, sb, (long long)200000);
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, sb);
    HIPCHK(hipStreamSynchronize(sa)); HIPCHK(hipStreamSynchronize(sb));
    int votes = 0;
    const int REPS = 5;
    for (int rep = 0; rep < REPS; ++rep) {
        // baseline: an empty kernel on b with a idle
        HIPCHK(hipStreamSynchronize(sa)); HIPCHK(hipStreamSynchronize(sb));
        double t0 = now_us();
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, sb);
        HIPCHK(hipStreamSynchronize(sb));
        const double base = now_us() - t0;
        // the same behind a ~0.4 ms kernel on a: on one hardware queue it has to wait for that kernel
        t0 = now_us();
        hipLaunchKernelGGL(k_idle_cycles, dim3(1), dim3(64), 0, sa, (long long)1000000);       // ~0.4 ms at 2.4 GHz
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, sb);
        HIPCHK(hipStreamSynchronize(sb));
        const double busy = now_us() - t0;
        HIPCHK(hipStreamSynchronize(sa));
        if (busy > base + 200.0) ++votes;
    }
    LAUNCHCHK();
    *out = votes * 2 > REPS ? 1 : 0;
    return NEP_OK;
}

int32_t nep_csc_to_csr(int64_t n, const int64_t* colptr, const int64_t* rowval, const void* nzval,
                       int32_t val_is_complex, int32_t one_based, int32_t* rowptr, int32_t* colind,
                       void* vals) {
    ARGCHK(n >= 0 && colptr && rowptr);
    const int64_t off = one_based ? 1 : 0;
    const int64_t nnz = colptr[n] - off;
    ARGCHK(nnz < (int64_t)1 << 31);
    std::vector<int32_t> cnt(n + 1, 0);
    for (int64_t e = 0; e < nnz; ++e) cnt[rowval[e] - off + 1]++;
    rowptr[0] = 0;
    for (int64_t i = 0; i < n; ++i) rowptr[i + 1] = rowptr[i] + cnt[i + 1];
    std::vector<int32_t> pos(rowptr, rowptr + n);
    for (int64_t c = 0; c < n; ++c)
        for (int64_t e = colptr[c] - off; e < colptr[c + 1] - off; ++e) {
            int64_t r = rowval[e] - off;
            int32_t q = pos[r]++;
            colind[q] = (int32_t)c;
            if (val_is_complex) ((nep_cdouble*)vals)[q] = ((const nep_cdouble*)nzval)[e];
            else ((double*)vals)[q] = ((const double*)nzval)[e];
        }
    return NEP_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// kernels
__global__ void k_iar_shift_scale(int64_t n, int k, const cplx* __restrict__ src, cplx* __restrict__ dst) {
    // dst[(j+1)*n + r] = src[j*n + r]/(j+1)   -- one pass over n*k contiguous complex entries
    const int64_t total = n * (int64_t)k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(i / n);
        const double s = 1.0 / (double)(j + 1);
        cplx v = src[i];
        dst[i + n] = cmake(v.x * s, v.y * s);
    }
}

__global__ void k_axpy(int64_t len, cplx alpha, const cplx* __restrict__ x, cplx* __restrict__ y) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len;
         i += (int64_t)gridDim.x * blockDim.x) {
        cplx acc = y[i];
        cfma(acc, alpha, x[i]);
        y[i] = acc;
    }
}

__global__ void k_copy(int64_t len, const cplx* __restrict__ x, cplx* __restrict__ y) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len;
         i += (int64_t)gridDim.x * blockDim.x)
        y[i] = x[i];
}

__global__ void k_scal(int64_t len, cplx alpha, cplx* __restrict__ x) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len;
         i += (int64_t)gridDim.x * blockDim.x)
        x[i] = cmul(alpha, x[i]);
}

// per-block partial of sum conj(x_j) y_j for column j = blockIdx.y; partial[(j*gridDim.x + b)]
template <bool CONJ>
__global__ __launch_bounds__(256) void k_coldots_partial(int64_t rows, const cplx* __restrict__ X,
                                                         int64_t ldx, const cplx* __restrict__ Y,
                                                         int64_t ldy, cplx* __restrict__ partial) {
    const int j = blockIdx.y;
    const cplx* x = X + (int64_t)j * ldx;
    const cplx* y = Y + (int64_t)j * ldy;
    cplx acc = cmake(0.0, 0.0);
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < rows;
         i += (int64_t)gridDim.x * blockDim.x)
        if (CONJ) cfma_conj(acc, x[i], y[i]); else cfma(acc, x[i], y[i]);
    acc = group_reduce_sum<64>(acc);
    __shared__ cplx sm[4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) sm[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        cplx t = sm[0];
        for (int q = 1; q < 4; ++q) t = cadd(t, sm[q]);
        partial[(int64_t)j * gridDim.x + blockIdx.x] = t;
    }
}
// out[j] = sum_b partial[j*nb + b]  (fixed order -> deterministic)
__global__ void k_sum_partials(int nb, const cplx* __restrict__ partial, cplx* __restrict__ out, int k) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    cplx t = cmake(0.0, 0.0);
    for (int b = 0; b < nb; ++b) t = cadd(t, partial[(int64_t)j * nb + b]);
    out[j] = t;
}

// row-major (rows x k, ld lds) -> column-major selected columns
__global__ __launch_bounds__(256) void k_rm2cm(int64_t rows, const cplx* __restrict__ src, int64_t lds,
                                               const int* __restrict__ cols, int ncols,
                                               cplx* __restrict__ dst, int64_t ldd) {
    // tile 64 rows x 16 cols through LDS so that both sides are reasonably coalesced
    __shared__ cplx tile[16][65];
    const int64_t r0 = blockIdx.x * 64LL;
    const int c0 = blockIdx.y * 16;
    const int t = threadIdx.x;
    // load: threads over (row = t/16 + 16*i, col = t%16)
    for (int i = 0; i < 4; ++i) {
        const int rr = (t >> 4) + 16 * i;
        const int cc = t & 15;
        const int64_t r = r0 + rr;
        if (r < rows && c0 + cc < ncols) {
            const int sc = cols ? cols[c0 + cc] : (c0 + cc);
            tile[cc][rr] = src[r * lds + sc];
        }
    }
    __syncthreads();
    for (int i = 0; i < 4; ++i) {
        const int cc = (t >> 6) + 4 * i;
        const int rr = t & 63;
        const int64_t r = r0 + rr;
        if (r < rows && c0 + cc < ncols) dst[(int64_t)(c0 + cc) * ldd + r] = tile[cc][rr];
    }
}

// out[r] = sum_j A[r + j*lda] * B[r + j*ldb]   (no conjugation)
__global__ void k_rowdot(int64_t rows, int k, const cplx* __restrict__ A, int64_t lda, const cplx* __restrict__ B,
                         int64_t ldb, cplx* __restrict__ out) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x) {
        cplx acc = cmake(0.0, 0.0);
        for (int j = 0; j < k; ++j) cfma(acc, A[r + (int64_t)j * lda], B[r + (int64_t)j * ldb]);
        out[r] = acc;
    }
}
// A[r + j*lda] *= B[r + j*ldb]
__global__ void k_hadamard(int64_t rows, int k, cplx* __restrict__ A, int64_t lda, const cplx* __restrict__ B,
                           int64_t ldb) {
    const int64_t total = rows * (int64_t)k;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i % rows, j = i / rows;
        A[r + j * lda] = cmul(A[r + j * lda], B[r + j * ldb]);
    }
}
// out[i] = (|x[i]|, 0)
__global__ void k_absvec(int64_t len, const cplx* __restrict__ x, cplx* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = cmake(hypot(x[i].x, x[i].y), 0.0);
}
// per-block partial column sums of |X[r*ld + s]|^2 of a row-major block; partial[b*k + s]
__global__ __launch_bounds__(256) void k_rm_colnorm_partial(int64_t rows, int k, const cplx* __restrict__ XT, int64_t ld,
                                                            double* __restrict__ partial) {
    // thread t handles column s = t % kk-strided, rows strided by (256/ kpad)
    extern __shared__ double red[];
    const int s = threadIdx.x % 64;
    const int sub = threadIdx.x / 64;       // 4 row phases
    for (int s0 = 0; s0 < k; s0 += 64) {
        const int col = s0 + s;
        double acc = 0.0;
        if (col < k)
            for (int64_t r = blockIdx.x * 4LL + sub; r < rows; r += gridDim.x * 4LL) {
                const cplx v = XT[r * ld + col];
                acc = fma(v.x, v.x, fma(v.y, v.y, acc));
            }
        red[sub * 64 + s] = acc;
        __syncthreads();
        if (sub == 0 && col < k)
            partial[(int64_t)blockIdx.x * k + col] = (red[s] + red[64 + s]) + (red[128 + s] + red[192 + s]);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_sum_partials_rows(int nb, int len, const double* __restrict__ partial,
                                                           double* __restrict__ out) {
    __shared__ double sm[4];
    const int j = blockIdx.x;
    double t = 0.0;
    for (int b = threadIdx.x; b < nb; b += 256) t += partial[(int64_t)b * len + j];
    t = wave_reduce_sum(t);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) out[j] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

// NLEIGS continuation vector, block form (src/method_nleigs.jl:418-435):
//   Bw[0:n] = 0 ;  Bw[i n + r] = wc[(i-1) n + r] + c[i-1] * wc[i n + r],  i = 1..N
__global__ void k_rk_bw(int64_t n, int N, const cplx* __restrict__ wc, const cplx* __restrict__ c, cplx* __restrict__ Bw) {
    const int64_t total = n * (int64_t)(N + 1);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / n;
        if (i == 0) { Bw[t] = cmake(0.0, 0.0); continue; }
        cplx v = wc[t - n];
        cfma(v, c[i - 1], wc[t]);
        Bw[t] = v;
    }
}
// block recurrence x_i = a[i-1]*y_i + b[i-1]*x_{i-1}, i = 1..N (blocks of n entries, x_0 given; y may alias x)
// (src/method_nleigs.jl:445-487 for z, :496-515 for w); one thread per row, sequential over the blocks
__global__ void k_block_recur(int64_t n, int N, const cplx* __restrict__ a, const cplx* __restrict__ b, const cplx* y,
                              cplx* x) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        cplx prev = x[r];
        for (int i = 1; i <= N; ++i) {
            cplx v = cmul(a[i - 1], y[(int64_t)i * n + r]);
            cfma(v, b[i - 1], prev);
            x[(int64_t)i * n + r] = v;
            prev = v;
        }
    }
}

// y[j] = d[j] * sum_i conj(A[i + j*lda]) x[i]   (d optional): small dense A^H x with the result on the device -- the
// scaled-DFT products of the waveguide boundary operator P(lam)^{-1} = R diag(1/s) R^H / nz (Waveguide.jl:159-170) and the
// SMW coefficient solve alpha = M^{-1} f.  One wave per column, 16-byte coalesced reads down the column.
__global__ __launch_bounds__(256) void k_gemv_hd(const cplx* __restrict__ A, int64_t lda, int64_t rows, int k,
                                                 const cplx* __restrict__ x, const cplx* __restrict__ d, cplx* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= k) return;
    const cplx* a = A + (int64_t)j * lda;
    // eight 1 KB trips of the column in flight per wave (one at a time was latency-bound: 37 MB of the 1517^2 SMW inverse in
    // 10-14 us); two accumulators, fixed order
    cplx acc = cmake(0.0, 0.0), acc2 = cmake(0.0, 0.0);
    int64_t i = lane;
    for (; i + 7 * 64 < rows; i += 8 * 64) {
        cplx av[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { av[u] = a[i + u * 64]; xv[u] = x[i + u * 64]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { cfma_conj(acc, av[u], xv[u]); cfma_conj(acc2, av[u + 1], xv[u + 1]); }
    }
    for (; i < rows; i += 64) cfma_conj(acc, a[i], x[i]);
    acc = cadd(acc, acc2);
    acc = group_reduce_sum<64>(acc);
    if (lane == 0) y[j] = d ? cmul(d[j], acc) : acc;
}

static inline int grid_for(int64_t work, int block, int cap = 4096) {
    int64_t g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

static thread_local NepScratch g_util_scratch;  // partial sums for nrm2/coldots (single host thread per process use)

extern "C" {

int32_t nep_iar_shift_scale(int64_t n, int32_t k, const nep_cdouble* dsrc, nep_cdouble* ddst,
                            nep_stream stream) {
    ARGCHK(n > 0 && k >= 0);
    if (k == 0) return NEP_OK;
    hipLaunchKernelGGL(k_iar_shift_scale, dim3(grid_for(n * k, 256)), dim3(256), 0, as_stream(stream), n,
                       (int)k, (const cplx*)dsrc, (cplx*)ddst);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_dev_copy(void* ddst, const void* dsrc, size_t bytes, nep_stream stream) {
    // complex128-aligned copies go through a kernel (a D2D hipMemcpyAsync costs ~20-30 us of host time per call)
    if (bytes % 16 == 0 && ((uintptr_t)ddst % 16) == 0 && ((uintptr_t)dsrc % 16) == 0 && bytes > 0) {
        const int64_t len = (int64_t)(bytes / 16);
        hipLaunchKernelGGL(k_copy, dim3(grid_for(len, 256)), dim3(256), 0, as_stream(stream), len, (const cplx*)dsrc,
                           (cplx*)ddst);
        LAUNCHCHK();
        return NEP_OK;
    }
    HIPCHK(hipMemcpyAsync(ddst, dsrc, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return NEP_OK;
}

int32_t nep_axpy(int64_t len, nep_cdouble alpha, const nep_cdouble* dx, nep_cdouble* dy, nep_stream stream) {
    ARGCHK(len >= 0);
    if (len == 0) return NEP_OK;
    cplx a; a.x = alpha.re; a.y = alpha.im;
    hipLaunchKernelGGL(k_axpy, dim3(grid_for(len, 256)), dim3(256), 0, as_stream(stream), len, a,
                       (const cplx*)dx, (cplx*)dy);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_scal(int64_t len, nep_cdouble alpha, nep_cdouble* dx, nep_stream stream) {
    ARGCHK(len >= 0);
    if (len == 0) return NEP_OK;
    cplx a; a.x = alpha.re; a.y = alpha.im;
    hipLaunchKernelGGL(k_scal, dim3(grid_for(len, 256)), dim3(256), 0, as_stream(stream), len, a, (cplx*)dx);
    LAUNCHCHK();
    return NEP_OK;
}

static int coldots_impl(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx, const nep_cdouble* dY,
                        int64_t ldy, nep_cdouble* h_out, nep_stream stream, bool conj);

int32_t nep_coldots(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx, const nep_cdouble* dY,
                    int64_t ldy, nep_cdouble* h_out, nep_stream stream) {
    return coldots_impl(rows, k, dX, ldx, dY, ldy, h_out, stream, true);
}

int32_t nep_coldotsu(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx, const nep_cdouble* dY,
                     int64_t ldy, nep_cdouble* h_out, nep_stream stream) {
    return coldots_impl(rows, k, dX, ldx, dY, ldy, h_out, stream, false);
}

static int coldots_impl(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx, const nep_cdouble* dY,
                        int64_t ldy, nep_cdouble* h_out, nep_stream stream, bool conj) {
    ARGCHK(rows > 0 && k > 0 && h_out);
    const int nb = grid_for(rows, 256 * 8, 512);
    int rc = g_util_scratch.ensure(((size_t)k * nb + k) * sizeof(cplx));
    if (rc) return rc;
    cplx* partial = (cplx*)g_util_scratch.dptr;
    cplx* out = partial + (size_t)k * nb;
    if (conj)
        hipLaunchKernelGGL((k_coldots_partial<true>), dim3(nb, k), dim3(256), 0, as_stream(stream), rows, (const cplx*)dX,
                           ldx, (const cplx*)dY, ldy, partial);
    else
        hipLaunchKernelGGL((k_coldots_partial<false>), dim3(nb, k), dim3(256), 0, as_stream(stream), rows, (const cplx*)dX,
                           ldx, (const cplx*)dY, ldy, partial);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_sum_partials, dim3((k + 63) / 64), dim3(64), 0, as_stream(stream), nb, partial, out, (int)k);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(h_out, out, (size_t)k * sizeof(cplx), hipMemcpyDeviceToHost, as_stream(stream)));
    HIPCHK(hipStreamSynchronize(as_stream(stream)));
    return NEP_OK;
}

int32_t nep_colnorms(int64_t rows, int32_t k, const nep_cdouble* dX, int64_t ldx, double* h_out,
                     nep_stream stream) {
    ARGCHK(h_out != nullptr);
    std::vector<nep_cdouble> tmp(k);
    int rc = nep_coldots(rows, k, dX, ldx, dX, ldx, tmp.data(), stream);
    if (rc) return rc;
    for (int j = 0; j < k; ++j) h_out[j] = sqrt(tmp[j].re);
    return NEP_OK;
}

int32_t nep_nrm2(int64_t len, const nep_cdouble* dx, double* h_out, nep_stream stream) {
    return nep_colnorms(len, 1, dx, len, h_out, stream);
}

// host only: src/rk_helper/discretizepolygon.jl, the boundary walk (see include/nepmi355.h).  Every operation separately rounded, in
// the order of the interpreted walk (NumPy scalars): |z1 - z0| through hypot, a real times a complex as the full complex product
// with a zero imaginary part.
int32_t nep_discretize_polygon(int32_t nz, const nep_cdouble* h_z, int32_t npts, nep_cdouble* h_out) {
#pragma clang fp contract(off)
    ARGCHK(nz >= 3 && h_z && npts >= 1 && h_out);
    std::vector<double> zx((size_t)nz + 1), zy((size_t)nz + 1);
    for (int i = 0; i < nz; ++i) { zx[i] = h_z[i].re; zy[i] = h_z[i].im; }
    zx[nz] = zx[0]; zy[nz] = zy[0];
    // L = np.sum(abs(np.diff(z))): NumPy's pairwise summation for n >= 8 differs from a running sum -- the caller passes L's
    // terms through the same np.sum, so L comes in h_out[0].re
    const double L = h_out[0].re;
    int ind = 0; double alph = 0.0;
    const double step = L / (double)npts;
    double remL = step;
    int have = 1;
    h_out[0].re = zx[0]; h_out[0].im = zy[0];
    while (have < npts) {
        if (ind >= nz) { nep_set_error("discretize_polygon: walked past the last edge"); return NEP_ERR_ARG; }
        const double dx = zx[ind + 1] - zx[ind], dy = zy[ind + 1] - zy[ind];
        const double d = hypot(dx, dy);
        const double t = (1.0 - alph) * d;
        if (t < remL) {
            ind += 1;
            remL = remL - t;
            alph = 0.0;
        } else {
            const double q = remL / d;
            alph = alph + q;
            remL = step;
            // z[ind] + alph * (z[ind+1] - z[ind]) with alph promoted to (alph + 0i)
            const double pr = alph * dx - 0.0 * dy, pi = alph * dy + 0.0 * dx;
            h_out[have].re = zx[ind] + pr; h_out[have].im = zy[ind] + pi;
            ++have;
        }
    }
    return NEP_OK;
}

static thread_local NepScratch g_rk_scratch;
static thread_local PinnedRing g_rk_ring;

int32_t nep_rk_bw(int64_t n, int32_t N, const nep_cdouble* dwc, const nep_cdouble* h_c, nep_cdouble* dBw, nep_stream stream) {
    ARGCHK(n > 0 && N >= 0 && dwc && dBw && (N == 0 || h_c));
    hipStream_t st = as_stream(stream);
    int rc = g_rk_scratch.ensure((size_t)(2 * N + 2) * sizeof(cplx));
    if (rc) return rc;
    if (N > 0) { rc = g_rk_ring.upload(g_rk_scratch.dptr, h_c, (size_t)N * sizeof(cplx), st); if (rc) return rc; }
    hipLaunchKernelGGL(k_rk_bw, dim3(grid_for(n * (N + 1), 256)), dim3(256), 0, st, n, (int)N, (const cplx*)dwc,
                       (const cplx*)g_rk_scratch.dptr, (cplx*)dBw);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_block_recur(int64_t n, int32_t N, const nep_cdouble* h_a, const nep_cdouble* h_b, const nep_cdouble* dy,
                        nep_cdouble* dx, nep_stream stream) {
    ARGCHK(n > 0 && N >= 0 && dy && dx);
    if (N == 0) return NEP_OK;
    ARGCHK(h_a && h_b);
    hipStream_t st = as_stream(stream);
    int rc = g_rk_scratch.ensure((size_t)(2 * N + 2) * sizeof(cplx));
    if (rc) return rc;
    std::vector<nep_cdouble> ab(2 * (size_t)N);
    for (int i = 0; i < N; ++i) { ab[i] = h_a[i]; ab[N + i] = h_b[i]; }
    rc = g_rk_ring.upload(g_rk_scratch.dptr, ab.data(), ab.size() * sizeof(cplx), st);
    if (rc) return rc;
    const cplx* da = (const cplx*)g_rk_scratch.dptr;
    hipLaunchKernelGGL(k_block_recur, dim3(grid_for(n, 256)), dim3(256), 0, st, n, (int)N, da, da + N, (const cplx*)dy,
                       (cplx*)dx);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_gemv_hd(const nep_cdouble* dA, int64_t lda, int64_t rows, int32_t k, const nep_cdouble* dx,
                    const nep_cdouble* dd, nep_cdouble* dy, nep_stream stream) {
    ARGCHK(dA && dx && dy && rows > 0 && k >= 1 && lda >= rows);
    hipLaunchKernelGGL(k_gemv_hd, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, as_stream(stream), (const cplx*)dA, lda, rows,
                       (int)k, (const cplx*)dx, (const cplx*)dd, (cplx*)dy);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_rowdot(int64_t rows, int32_t k, const nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb,
                   nep_cdouble* dout, nep_stream stream) {
    ARGCHK(rows > 0 && k >= 1 && dA && dB && dout);
    hipLaunchKernelGGL(k_rowdot, dim3(grid_for(rows, 256)), dim3(256), 0, as_stream(stream), rows, (int)k,
                       (const cplx*)dA, lda, (const cplx*)dB, ldb, (cplx*)dout);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_hadamard(int64_t rows, int32_t k, nep_cdouble* dA, int64_t lda, const nep_cdouble* dB, int64_t ldb,
                     nep_stream stream) {
    ARGCHK(rows > 0 && k >= 1 && dA && dB);
    hipLaunchKernelGGL(k_hadamard, dim3(grid_for(rows * k, 256)), dim3(256), 0, as_stream(stream), rows, (int)k,
                       (cplx*)dA, lda, (const cplx*)dB, ldb);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_absvec(int64_t len, const nep_cdouble* dx, nep_cdouble* dout, nep_stream stream) {
    ARGCHK(len > 0 && dx && dout);
    hipLaunchKernelGGL(k_absvec, dim3(grid_for(len, 256)), dim3(256), 0, as_stream(stream), len, (const cplx*)dx, (cplx*)dout);
    LAUNCHCHK();
    return NEP_OK;
}

int32_t nep_rowmajor_colnorms(int64_t rows, int32_t k, const nep_cdouble* dXT, int64_t ld, double* h_out,
                              nep_stream stream) {
    ARGCHK(rows > 0 && k >= 1 && ld >= k && dXT && h_out);
    hipStream_t st = as_stream(stream);
    const int nb = (int)std::min<int64_t>((rows + 3) / 4, 1024);
    int rc = g_util_scratch.ensure(((size_t)nb * k + k) * sizeof(double));
    if (rc) return rc;
    double* partial = (double*)g_util_scratch.dptr;
    double* outd = partial + (size_t)nb * k;
    hipLaunchKernelGGL(k_rm_colnorm_partial, dim3(nb), dim3(256), 256 * sizeof(double), st, rows, (int)k,
                       (const cplx*)dXT, ld, partial);
    LAUNCHCHK();
    hipLaunchKernelGGL(k_sum_partials_rows, dim3(k), dim3(256), 0, st, nb, (int)k, partial, outd);
    LAUNCHCHK();
    HIPCHK(hipMemcpyAsync(h_out, outd, (size_t)k * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < k; ++j) h_out[j] = sqrt(h_out[j]);
    return NEP_OK;
}

int32_t nep_rowmajor_to_colmajor(int64_t rows, int32_t k, const nep_cdouble* dsrc, int64_t lds,
                                 const int32_t* h_cols, int32_t ncols, nep_cdouble* ddst, int64_t ldd,
                                 nep_stream stream) {
    ARGCHK(rows > 0 && k > 0 && lds >= k && ldd >= rows);
    if (!h_cols) ncols = k;
    ARGCHK(ncols >= 0);
    if (ncols == 0) return NEP_OK;
    int* dcols = nullptr;
    if (h_cols) {
        for (int i = 0; i < ncols; ++i) ARGCHK(h_cols[i] >= 0 && h_cols[i] < k);
        int rc = g_util_scratch.ensure((size_t)ncols * sizeof(int));
        if (rc) return rc;
        dcols = (int*)g_util_scratch.dptr;
        HIPCHK(hipMemcpyAsync(dcols, h_cols, (size_t)ncols * sizeof(int), hipMemcpyHostToDevice, as_stream(stream)));
    }
    dim3 grid((unsigned)((rows + 63) / 64), (unsigned)((ncols + 15) / 16));
    hipLaunchKernelGGL(k_rm2cm, grid, dim3(256), 0, as_stream(stream), rows, (const cplx*)dsrc, lds,
                       (const int*)dcols, (int)ncols, (cplx*)ddst, ldd);
    LAUNCHCHK();
    if (h_cols) HIPCHK(hipStreamSynchronize(as_stream(stream)));  // scratch reuse safety
    return NEP_OK;
}

}  // extern "C"
