// libnepmi355: one infinite-Arnoldi step as ONE call (the inner statements of src/method_iar.jl:94-109 between two
// eigenvalue checks).  Every piece is an entry point of this library already; this file only sequences them natively so
// that the host language issues one foreign call per Arnoldi step instead of a dozen (a Python host spends ~0.25 ms per
// step in ctypes marshalling at gun size, as much as the device needs for the step itself).
//
//   y[:,1]  = compute_Mlincomb!(nep, sigma, y[:,1:k], a[1:k], 1)      K1  nep_mlincomb_dev on column k of the basis
//   y[:,1]  = -lin_solve(M0inv, y[:,1])                                K5  nep_lu_solve [+ UMFPACK-style refinement, blind plan]
//   vv      = reshape(y[:,1:k+1])  with  y[:,2:k+1] ./ (1:k)'          nep_iar_shift_scale
//   h, beta = orthogonalize_and_normalize!(VV, vv, h, DGKS)            K6  nep_orth_dev (decision on the device)
//   H[:,k]  -> pinned host memory behind an event                      (the host reads it when it needs eig(H_k))
//
// Refinement without read-backs: the step takes `refine_steps` correction sweeps blindly and RECORDS UMFPACK's componentwise
// backward error omega of every iterate x_0 .. x_r behind the H row (4 doubles at entries k+2, k+3 of the row).  The host
// replays UMFPACK's stopping rule on the recorded values when the row arrives and re-runs the call with checked solves in
// the (never yet observed) case that the rule would have asked for more sweeps than were taken -- no step of the
// recurrence waits for the device (the checked form drained the queue on 17 of the 100 gun solves).
#include "common.h"
#include <vector>
#include <stdlib.h>
#include <time.h>

struct nep_iar {
    nep_spmf* spmf; nep_lu* lu;
    int64_t n, ldv; int32_t m, mt;
    cplx* dV; const cplx* dCtab; int64_t ldc; const int64_t* d_active;
    cplx* dz; cplx* dW;                      // z (n) ; refinement work r, x (2n)
    std::vector<double> cabs; std::vector<nep_cdouble> cf;
    cplx* dH; nep_cdouble* hH;               // m rows of (m+4): device / pinned host; row k-1 = h[0..k), beta, flags, omegas
    double* d_cabs; cplx* d_ccf;             // |f_t(sigma)|, f_t(sigma) resident on the device (refinement residuals)
    cplx* hH_dev;                            // device address of the pinned block (NULL: not mapped, rows travel by memcpy)
    int32_t method;
    std::vector<hipEvent_t> ev;              // ev[k]: H column k is in pinned memory
    // the recorded residual of the kept iterate (a pure check: nothing of the recurrence depends on it) runs on a side stream
    // next to the projections of the Gram-Schmidt pass; the pass' first write to the vector waits for it
    hipStream_t side = nullptr; hipEvent_t e_solved = nullptr, e_checked = nullptr;
    hipEvent_t e_upload = nullptr; bool upload_waited = false;      // |f_t|, f_t were uploaded on the NULL stream
    hipStream_t last = nullptr;
    bool no_events = false;                  // nep_iar_steps_graph: the steps are being captured, their events are recorded behind the graph launch
    std::vector<hipGraphExec_t> graphs;      // chunk graphs launched so far (destroyed with the object: a graph must outlive its execution)
    // step k's last kernel (k_orth_finish_vc) forms step k + 1's coefficient product and block shift: dWT (n x mt) holds the
    // product for step `wt_for` (0: none)
    cplx* dWT = nullptr; int32_t wt_for = 0;
};

extern "C" {

int32_t nep_iar_create(nep_spmf* spmf, nep_lu* lu, int64_t n, int32_t m, nep_cdouble* dV, int64_t ldv,
                       const nep_cdouble* dCtab, int64_t ldc, const int64_t* d_active, nep_cdouble* dwork3n,
                       const double* h_cabs, const nep_cdouble* h_cf, int32_t mt, nep_cdouble* dH, nep_cdouble* h_pinnedH,
                       int32_t orth_method, nep_iar** out) {
    ARGCHK(out != nullptr);
    *out = nullptr;
    ARGCHK(spmf && lu && dV && dCtab && dwork3n && dH && h_pinnedH);
    ARGCHK(n > 0 && m >= 1 && ldv >= n * (int64_t)(m + 1) && ldc >= m && mt >= 1 && (orth_method == 0 || orth_method == 1));
    nep_iar* s = new nep_iar();
    s->spmf = spmf; s->lu = lu; s->n = n; s->ldv = ldv; s->m = m; s->mt = mt;
    s->dV = (cplx*)dV; s->dCtab = (const cplx*)dCtab; s->ldc = ldc; s->d_active = d_active;
    s->dz = (cplx*)dwork3n; s->dW = (cplx*)dwork3n + n;
    s->d_cabs = nullptr; s->d_ccf = nullptr;
    if (h_cabs && h_cf) {
        s->cabs.assign(h_cabs, h_cabs + mt); s->cf.assign(h_cf, h_cf + mt);
        void* p = nullptr;
        if (nep_pool_alloc(&p, (size_t)mt * 24 + 64)) { delete s; return NEP_ERR_HIP; }
        s->d_cabs = (double*)p; s->d_ccf = (cplx*)((char*)p + (((size_t)mt * 8 + 15) & ~(size_t)15));
        // asynchronous (pinned staging): a synchronous copy would queue behind the factorisation's build kernels
        static thread_local PinnedRing ring;
        int rcu = ring.upload(s->d_cabs, h_cabs, (size_t)mt * 8, nullptr);
        if (!rcu) rcu = ring.upload(s->d_ccf, h_cf, (size_t)mt * 16, nullptr);
        if (rcu) { nep_pool_free(p); delete s; return rcu; }
        // the steps run on the caller's stream, which need not be ordered behind the NULL stream (non-blocking streams)
        if (hipEventCreateWithFlags(&s->e_upload, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(s->e_upload, nullptr);
        else (void)hipGetLastError();
        // MEASURED: slower (gun iar 45.1 ms per call against 42.8 ms): the two cross-stream dependencies per step cost more on
        // the device than the 11 us kernel they take off the critical path.  Opt-in (NEP_IAR_RESID_OVERLAP=1).
        static const int overlap = getenv("NEP_IAR_RESID_OVERLAP") ? atoi(getenv("NEP_IAR_RESID_OVERLAP")) : 0;
        if (overlap && hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) == hipSuccess) {
            if (hipEventCreateWithFlags(&s->e_solved, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&s->e_checked, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError(); (void)hipStreamDestroy(s->side); s->side = nullptr;
            }
        } else (void)hipGetLastError();
    }
    s->dH = (cplx*)dH; s->hH = h_pinnedH; s->method = orth_method;
    const int fuse_vc = getenv("NEP_IAR_FUSE_VC") ? atoi(getenv("NEP_IAR_FUSE_VC")) : 1;      // (read per object: tests compare both forms)
    if (fuse_vc && mt <= 4) {
        void* p = nullptr;
        if (nep_pool_alloc(&p, (size_t)n * mt * sizeof(cplx)) == 0) s->dWT = (cplx*)p;
    }
    s->ev.assign(m + 1, nullptr);
    s->hH_dev = nullptr;
    if (!getenv("NEP_IAR_NO_MIRROR")) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, h_pinnedH, 0) == hipSuccess) s->hH_dev = (cplx*)dp; else (void)hipGetLastError();
    }
    *out = s;
    return NEP_OK;
}

int32_t nep_iar_destroy(nep_iar* s) {
    if (!s) return NEP_OK;
    for (hipEvent_t e : s->ev) if (e) (void)hipEventDestroy(e);
    if (!s->graphs.empty()) {
        if (s->last) (void)hipStreamSynchronize(s->last);          // a graph must not be destroyed while it executes
        for (hipGraphExec_t g : s->graphs) (void)hipGraphExecDestroy(g);
    }
    if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }
    if (s->e_solved) (void)hipEventDestroy(s->e_solved);
    if (s->e_checked) (void)hipEventDestroy(s->e_checked);
    if (s->e_upload) (void)hipEventDestroy(s->e_upload);
    // steps (speculative ones past convergence included) may still be queued on the last stream: the block goes back to the
    // pool behind them
    if (s->d_cabs) nep_pool_free_on(s->d_cabs, s->last, s->last != nullptr);
    if (s->dWT) nep_pool_free_on(s->dWT, s->last, s->last != nullptr);
    delete s;
    return NEP_OK;
}

int32_t nep_iar_step(nep_iar* s, int32_t k, int32_t refine_steps, nep_stream stream) {
    // refine_steps + NEP_IAR_SKIP_FINAL_RECORD (0x100): the backward error of the iterate that is KEPT is not recorded in steps
    // whose index is no multiple of 8 (a pure check, one pass over the matrices per step: the caller asks for this once its
    // refinement count has settled -- the policy FactorizeLinSolver.solve_dev applies to its checked solves, 7 of 8 taken on trust)
    const bool skip_final = (refine_steps & 0x100) != 0 && (k % 8) != 0;
    refine_steps &= 0xff;
    ARGCHK(s && k >= 1 && k <= s->m && refine_steps >= 0 && refine_steps <= 3);
    ARGCHK(refine_steps == 0 || !s->cabs.empty());
    hipStream_t st = as_stream(stream);
    const int64_t n = s->n;
    s->last = st;
    if (s->e_upload && !s->upload_waited) { HIPCHK(hipStreamWaitEvent(st, s->e_upload, 0)); s->upload_waited = true; }
    cplx* col = s->dV + (int64_t)(k - 1) * s->ldv;      // column k-1: the n x k block of the reference's reshape
    cplx* vv = s->dV + (int64_t)k * s->ldv;
    int32_t shifted = 0;
    int rc;
    if (s->dWT && s->wt_for == k) {       // the previous step's last kernel left the coefficient product and the shifted block
        rc = nep_spmv_wt(s->spmf, (const nep_cdouble*)s->dWT, (nep_cdouble*)s->dz, st);
        shifted = 1;
    } else
        rc = nep_mlincomb_dev_shift(s->spmf, k, (const nep_cdouble*)s->dCtab, s->ldc, (const nep_cdouble*)col, n, (nep_cdouble*)s->dz,
                                    (nep_cdouble*)(vv + n), &shifted, st);
    s->wt_for = 0;
    if (rc) return rc;
    cplx* hrow = s->dH + (int64_t)(k - 1) * (s->m + 4);
    unsigned long long* bits = (unsigned long long*)(hrow + k + 2);      // zero: dH is zero-filled by the caller, a row is used once
    const bool record = !s->cabs.empty();
    if (refine_steps == 0) {
        rc = nep_lu_solve(s->lu, 1, (const nep_cdouble*)s->dz, n, (nep_cdouble*)vv, n, -1.0, stream);
        if (rc) return rc;
    } else {
        cplx* r = s->dW; cplx* x = s->dW + n;
        rc = nep_lu_solve(s->lu, 1, (const nep_cdouble*)s->dz, n, (nep_cdouble*)x, n, 1.0, stream);
        for (int i = 0; i < refine_steps && !rc; ++i) {
            rc = nep_cw_resid_dev(s->spmf, s->d_cabs, (const nep_cdouble*)s->d_ccf, (const nep_cdouble*)x, (const nep_cdouble*)s->dz,
                                  (nep_cdouble*)r, bits + i, 1.0, st);
            if (rc) break;
            if (i == refine_steps - 1)
                rc = nep_lu_solve_add(s->lu, 1, (const nep_cdouble*)r, n, (const nep_cdouble*)x, n, (nep_cdouble*)vv, n, -1.0, stream);
            else
                rc = nep_lu_solve_add(s->lu, 1, (const nep_cdouble*)r, n, (const nep_cdouble*)x, n, (nep_cdouble*)x, n, 1.0, stream);
        }
        if (rc) return rc;
    }
    void* before_write = nullptr;
    if (record && !skip_final) {       // omega of the iterate that is kept (stored negated in vv)
        hipStream_t cs = st;
        if (s->side) {  // reads vv, z; writes dW and the omega word of this row: ordered behind the solve, ahead of the first update
            HIPCHK(hipEventRecord(s->e_solved, st));
            HIPCHK(hipStreamWaitEvent(s->side, s->e_solved, 0));
            cs = s->side;
        }
        rc = nep_cw_resid_dev(s->spmf, s->d_cabs, (const nep_cdouble*)s->d_ccf, (const nep_cdouble*)vv, (const nep_cdouble*)s->dz,
                              (nep_cdouble*)s->dW, bits + refine_steps, -1.0, cs);
        if (rc) return rc;
        if (s->side) { HIPCHK(hipEventRecord(s->e_checked, s->side)); before_write = (void*)s->e_checked; }
    }
    if (!shifted) {
        rc = nep_iar_shift_scale(n, k, (const nep_cdouble*)col, (nep_cdouble*)vv, stream);
        if (rc) return rc;
    }
    cplx* mirror = s->hH_dev ? s->hH_dev + (int64_t)(k - 1) * (s->m + 4) : nullptr;
    if (s->dWT && k < s->m) {           // + step k + 1's coefficient product and block shift (column k + 1, blocks 1 ..)
        rc = nep_orth_dev_iar_next((const nep_cdouble*)s->dV, s->ldv, n, k, s->d_active, (nep_cdouble*)vv, (nep_cdouble*)hrow, s->method,
                                   (nep_cdouble*)mirror, k + 4, before_write, (const nep_cdouble*)s->dCtab, s->ldc, s->mt,
                                   (nep_cdouble*)s->dWT, (nep_cdouble*)(vv + s->ldv + n), stream);
        if (!rc) s->wt_for = k + 1;
    } else
        rc = nep_orth_dev_mirror_ev((const nep_cdouble*)s->dV, s->ldv, n * (int64_t)(k + 1), k, s->d_active, (nep_cdouble*)vv,
                                    (nep_cdouble*)hrow, s->method, (nep_cdouble*)mirror, k + 4, before_write, stream);
    if (rc) return rc;
    if (!mirror)
        HIPCHK(hipMemcpyAsync(s->hH + (int64_t)(k - 1) * (s->m + 4), hrow, (size_t)(k + 4) * sizeof(cplx), hipMemcpyDeviceToHost, st));
    if (s->no_events) return NEP_OK;
    if (!s->ev[k]) HIPCHK(hipEventCreateWithFlags(&s->ev[k], hipEventDisableTiming | hipEventBlockingSync));
    HIPCHK(hipEventRecord(s->ev[k], st));
    return NEP_OK;
}

// steps k0 .. k0+count-1 in one call: a host language with a global interpreter lock hands the lock to its other threads
// (eigen workers, convergence checks) for the duration instead of fighting for it after every step
int32_t nep_iar_steps(nep_iar* s, int32_t k0, int32_t count, int32_t refine_steps, nep_stream stream) {
    ARGCHK(s && count >= 1);
    for (int32_t k = k0; k < k0 + count; ++k) {
        int rc = nep_iar_step(s, k, refine_steps, stream);
        if (rc) return rc;
    }
    return NEP_OK;
}

// The same steps as ONE hipGraph launch: the chunk's ~19 launches per step are captured (thread-local capture on the caller's
// stream), instantiated and replayed once.  Why: a dependent kernel dispatch costs ~4.6 us through the stream path on this part and
// ~1.5-2 us as a node of a graph (scripts/ub/ub_launch_chain.hip: 2.99 -> 1.53 us for a one-word kernel with arguments), and an
// Arnoldi step at gun size is 19 dependent dispatches around ~100-300 us of work.  The host has the time (nep_iar_run sleeps half of a
// call).  The steps' events are recorded behind the graph launch (an event recorded inside a capture cannot be waited for by a
// stream outside it): every step of the chunk completes, for its waiters, when the chunk does.  Anything that cannot be captured
// (a scratch block that has to grow: device synchronisation) ends the capture; the chunk is then issued the plain way.
// *captured = 1 when the graph route was taken.
int32_t nep_iar_steps_graph(nep_iar* s, int32_t k0, int32_t count, int32_t refine_steps, nep_stream stream, int32_t* captured) {
    ARGCHK(s && count >= 1);
    if (captured) *captured = 0;
    hipStream_t st = as_stream(stream);
    const int32_t wt_for0 = s->wt_for; const bool waited0 = s->upload_waited; hipStream_t last0 = s->last;
    bool ok = false;
    if (st != nullptr && hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        s->no_events = true;
        int rc = NEP_OK;
        for (int32_t k = k0; k < k0 + count && !rc; ++k) rc = nep_iar_step(s, k, refine_steps, stream);
        s->no_events = false;
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        hipGraphExec_t ge = nullptr;
        if (!rc && e == hipSuccess && g && hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) == hipSuccess && ge) {
            if (hipGraphLaunch(ge, st) == hipSuccess) { s->graphs.push_back(ge); ok = true; }
            else (void)hipGraphExecDestroy(ge);
        }
        if (g) (void)hipGraphDestroy(g);
        if (!ok) (void)hipGetLastError();
    } else (void)hipGetLastError();
    if (!ok) {                              // not capturable (or the NULL stream): the plain way, from the state the chunk started with
        s->wt_for = wt_for0; s->upload_waited = waited0; s->last = last0;
        return nep_iar_steps(s, k0, count, refine_steps, stream);
    }
    for (int32_t k = k0; k < k0 + count; ++k) {
        if (!s->ev[k]) HIPCHK(hipEventCreateWithFlags(&s->ev[k], hipEventDisableTiming | hipEventBlockingSync));
        HIPCHK(hipEventRecord(s->ev[k], st));
    }
    if (captured) *captured = 1;
    return NEP_OK;
}

// blocks the calling thread until column k of H has reached the pinned buffer
int32_t nep_iar_wait(nep_iar* s, int32_t k) {
    ARGCHK(s && k >= 1 && k <= s->m && s->ev[k]);
    // NEP_IAR_POLL_LAST: 2 (default) = every waiter polls, 1 = only those of the last 13 steps, 0 = all sleep on the interrupt
    static const int poll_last = getenv("NEP_IAR_POLL_LAST") ? atoi(getenv("NEP_IAR_POLL_LAST")) : 2;
    if (poll_last == 2 || (poll_last && k > s->m - 13)) {
        // the waiters poll (20 us naps) instead of sleeping on the event's interrupt: an interrupt-driven wait on this stack now
        // and then wakes 20-35 ms late (seen as "wait eig" tails and as 20-35 ms hipDeviceSynchronize calls on an idle device,
        // scripts/diag/tail_kernels.py) -- the checks are consumed in order, so a late waiter near the end delays the call --
        // and the naps cost LESS CPU than the runtime's own wait (0.19 s instead of 0.43 s of CPU per headline call: at most
        // LAG + 1 waiters exist at any time)
        for (;;) {
            const hipError_t e = hipEventQuery(s->ev[k]);
            if (e == hipSuccess) return NEP_OK;
            if (e != hipErrorNotReady) HIPCHK(e);
            struct timespec ts = {0, 20000};
            nanosleep(&ts, nullptr);
        }
    }
    HIPCHK(hipEventSynchronize(s->ev[k]));
    return NEP_OK;
}

// orders `stream` behind step k without involving the host
int32_t nep_iar_stream_wait(nep_iar* s, int32_t k, nep_stream stream) {
    ARGCHK(s && k >= 1 && k <= s->m && s->ev[k]);
    HIPCHK(hipStreamWaitEvent(as_stream(stream), s->ev[k], 0));
    return NEP_OK;
}

}  // extern "C"
