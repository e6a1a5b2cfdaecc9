// libnepmi355: the whole infinite-Arnoldi run as ONE foreign call.
//
// replaces: the body of `iar(::Type{T}, nep; ...)`, src/method_iar.jl:66-182, after `create_linsolver` (which stays with the
//           host language: the factorisation seam is LinSolvers.jl's) -- the Arnoldi recurrence (:94-109), `eigen(H[1:k,1:k])`
//           (:112), the Ritz block `Q = VV*Z` (:115), `estimate_error` of every Ritz pair (:133-135), the convergence count,
//           the sort of the errors and the extraction of the returned pairs (:137-160).
//
// Why it exists: the measured pipeline of this library (nep_iar_step + nep_hess_eig*_batch_dev + nep_gemm_ts_dev +
// nep_resid_batch_dev on three streams) used to be sequenced by the Python host only.  A host that binds the four plug-in
// seams (NEP / LinSolver / orthogonalisation / error measure) one call at a time gets a correct but host-synchronous loop;
// with this entry point a `iar(nep::DeviceSPMF; ...)` method is one `ccall`, and the Python host calls the same function.
//
// One host thread, nothing of the recurrence ever waits for the device:
//   recurrence   steps are enqueued on the caller's stream in chunks (nep_iar_steps); H's rows reach mapped pinned memory from
//                the last kernel of each step;
//   (A) eig      the Hessenberg eigen-decompositions of consecutive check steps go out as batches (one workgroup per step) on an
//                eig stream ordered behind the batch's last step by an event -- no host involvement;
//   (B) checks   when a batch's eigenvalues are in the pinned mirror the host forms lambda = sigma + gamma / D, asks the caller's
//                callback for f_t(lambda) (the only thing this library cannot evaluate: the scalar functions of the SPMF are the
//                host language's closures) and enqueues the Ritz GEMM (B operand = the device eigenvector block) and the
//                residual batch on a low-priority check stream;
//   (C) results  the 2 kc squared norms come back behind an event; errors, convergence count, sort.
// The three stages are polled between chunks of steps.  A run whose recorded refinement omegas / DGKS flags / eig status words
// ask for what the enqueued work did not do returns NEP_ERR_RETRY: the caller re-runs through its step-synchronous route
// (the reference's own loop over the four seams), exactly what the Python host did on such a miss.
#include "common.h"
#include <algorithm>
#include <deque>
#include <map>
#include <mutex>
#include <vector>
#include <math.h>
#include <time.h>

extern "C" {
int32_t nep_iar_create(nep_spmf* spmf, nep_lu* lu, int64_t n, int32_t m, nep_cdouble* dV, int64_t ldv, const nep_cdouble* dCtab,
                       int64_t ldc, const int64_t* d_active, nep_cdouble* dwork3n, const double* h_cabs, const nep_cdouble* h_cf,
                       int32_t mt, nep_cdouble* dH, nep_cdouble* h_pinnedH, int32_t orth_method, nep_iar** out);
int32_t nep_iar_steps_graph(nep_iar* s, int32_t k0, int32_t count, int32_t refine_steps, nep_stream stream, int32_t* captured);
}

namespace {

constexpr double EPS = 2.220446049250313e-16;

__global__ void k_iar_active(int64_t* act, int64_t n, int m1) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m1) act[j] = (int64_t)(j + 1) * n;
}
// column 0, block 0 of the basis = v / ||v|| (the norm is taken on the host: n numbers, once per run)
__global__ void k_iar_start(cplx* __restrict__ V, const cplx* __restrict__ v, double inv, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) V[i] = cmake(v[i].x * inv, v[i].y * inv);
}

// Column j of the basis is written by its step in full over its (j + 1) n active rows; rows beyond are never part of the arithmetic, but
// the Gram-Schmidt update kernels mask at TILE granularity (csrc/orth.hip: `r0 < act`), so the rows of the tile that straddles the
// end of a column's active part are read and must be zero.  Instead of a zero fill of the whole (m + 1) x n (m + 1) block (1.6 GB
// for gun at m = 100, 0.45 ms on the critical path of every call) only a slack of ZSLACK rows behind every column's active part is
// cleared (m + 1 columns x 128 KB).  NEP_IAR_FULL_ZERO=1 restores the full fill; NEP_IAR_POISON=1 (tests) fills the block with NaN
// patterns first, so that a kernel that read anything else would show.
constexpr int64_t ZSLACK = 8192;
__global__ __launch_bounds__(256) void k_iar_zero_slack(cplx* __restrict__ V, int64_t ldv, int64_t n, int m1) {
    const int j = blockIdx.y;
    const int64_t a = (int64_t)(j + 1) * n;
    const int64_t r = a + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < m1 && r < ldv && r < a + ZSLACK) V[(int64_t)j * ldv + r] = cmake(0.0, 0.0);
}

// streams of a run, kept per device for the life of the process (a fresh stream per call costs a hardware-queue assignment, and
// the pool allocator's event hand-off is per stream)
struct RunStreams { hipStream_t check = nullptr, eig = nullptr; hipStream_t probed_for = (hipStream_t)-1; bool shared = false; int retries = 0; };
std::mutex g_rs_mu;
std::map<int, RunStreams> g_rs;

// first of up to `ncand` fresh streams (priority prio) that shares a hardware queue with none of `others` (probed, ~1.5 ms per
// pair, once per process, device and caller stream); `keep` (may be NULL) is tried first.  The runtime maps streams onto a small
// pool of hardware queues and gives no way to ask which: two streams on one queue serialise (measured: the convergence checks on
// the recurrence's queue cost 15 ms per headline call).
int pick_stream(hipStream_t keep, int prio, int ncand, bool probe, const std::vector<hipStream_t>& others, hipStream_t* out, bool* shared) {
    std::vector<hipStream_t> cand;
    if (keep) cand.push_back(keep);
    hipStream_t pick = nullptr;
    for (int c = 0; c < ncand && !pick; ++c) {
        hipStream_t s = nullptr;
        if ((size_t)c < cand.size()) s = cand[c];
        else { HIPCHK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, prio)); cand.push_back(s); }
        bool clash = false;
        for (size_t i = 0; probe && i < others.size() && !clash; ++i) {
            int32_t a = 0;
            int rc = nep_stream_pair_serializes((nep_stream)others[i], (nep_stream)s, &a); if (rc) return rc;
            clash = a != 0;
        }
        if (!clash) pick = s;
    }
    if (getenv("NEP_IAR_RUN_TRACE")) fprintf(stderr, "nep_iar_run: stream pick (prio %d): %zu candidates tried, %s\n", prio, cand.size(), pick ? "free queue found" : "none free");
    if (!pick) { pick = cand[0]; *shared = true; }   // no free hardware queue: better a shared one than none (probed again by the next run)
    for (hipStream_t s : cand) if (s != pick) (void)hipStreamDestroy(s);
    *out = pick;
    return NEP_OK;
}

int run_streams(hipStream_t main, RunStreams* out) {
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_rs_mu);
    RunStreams& r = g_rs[dev];
    // (a pick that found no free hardware queue is not final: the next runs probe again, three times at most)
    if (!r.check || !r.eig || r.probed_for != main || (r.shared && r.retries < 3)) {
        if (r.shared) ++r.retries;
        r.shared = false;
        int least = 0, greatest = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int ncand = getenv("NEP_IAR_EIG_CANDIDATES") ? atoi(getenv("NEP_IAR_EIG_CANDIDATES")) : 8;
        const bool probe = !(getenv("NEP_IAR_EIG_PROBE") && atoi(getenv("NEP_IAR_EIG_PROBE")) == 0);
        // the convergence checks are off the critical path, but their stream gets NORMAL priority (NEP_IAR_CHECK_PRIO = 1: the device's
        // lowest, -1: its highest).  MEASURED (round 6, 16 fresh processes): with a lowest-priority stream half of the processes ran
        // the headline call in 57-69 ms instead of 32 -- both side streams then executed as if they shared the recurrence's queue,
        // although the probe below had found them free; with normal priority 8 of 8 processes ran at 31.5-32.1 ms.  (The Python host's
        // torch.cuda.Stream(priority=1) was clamped to normal by torch and never had a low-priority queue.)
        const int cprio = getenv("NEP_IAR_CHECK_PRIO") ? (atoi(getenv("NEP_IAR_CHECK_PRIO")) > 0 ? least : (atoi(getenv("NEP_IAR_CHECK_PRIO")) < 0 ? greatest : 0)) : 0;
        int rc = pick_stream(r.check, cprio, ncand, probe, {main}, &r.check, &r.shared); if (rc) return rc;
        // the decompositions are 3 ms one-workgroup kernels next to both
        // (NEP_IAR_EIG_PRIO: 0 = normal (default), -1 = the device's highest: a decomposition is ONE workgroup that needs half a CU's
        // LDS and has to find a CU between the recurrence's kernels)
        const int eprio = getenv("NEP_IAR_EIG_PRIO") ? (atoi(getenv("NEP_IAR_EIG_PRIO")) < 0 ? greatest : (atoi(getenv("NEP_IAR_EIG_PRIO")) > 0 ? least : 0)) : 0;
        rc = pick_stream(r.eig, eprio, ncand, probe, {main, r.check}, &r.eig, &r.shared); if (rc) return rc;
        r.probed_for = main;
    }
    *out = r;
    return NEP_OK;
}

// per-run host / device blocks that do not depend on the data, kept between runs of one shape (hipHostMalloc of the three pinned
// blocks alone is ~1 ms)
struct Arena {
    int dev = -1; int64_t n = 0; int32_t m = 0, mt = 0;
    cplx* Hdev = nullptr; cplx* wdev = nullptr; cplx* Ctab = nullptr; cplx* work3n = nullptr; int64_t* active = nullptr;
    cplx* v0 = nullptr; double* d_norms = nullptr; void* eigwork = nullptr; int64_t wsz = 0; int bmax = 0;
    cplx* Hpin = nullptr; cplx* wpin = nullptr; double* npin = nullptr; cplx* stage = nullptr;   // pinned
    void release() {
        for (void* p : {(void*)Hdev, (void*)wdev, (void*)Ctab, (void*)work3n, (void*)active, (void*)v0, (void*)d_norms, eigwork})
            if (p) nep_pool_free(p);
        for (void* p : {(void*)Hpin, (void*)wpin, (void*)npin, (void*)stage}) if (p) (void)hipHostFree(p);
        *this = Arena();
    }
};
std::mutex g_arena_mu;
std::vector<Arena> g_arenas;

int arena_acquire(int64_t n, int32_t m, int32_t mt, int bmax, Arena* out) {
    int dev = 0; HIPCHK(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_arena_mu);
        for (size_t i = 0; i < g_arenas.size(); ++i)
            if (g_arenas[i].dev == dev && g_arenas[i].n == n && g_arenas[i].m == m && g_arenas[i].mt == mt && g_arenas[i].bmax == bmax) {
                *out = g_arenas[i]; g_arenas.erase(g_arenas.begin() + i); return NEP_OK;
            }
    }
    Arena a; a.dev = dev; a.n = n; a.m = m; a.mt = mt; a.bmax = bmax;
    int64_t wsz = 0; int rc = nep_hess_eig_worksize(m, &wsz); if (rc) return rc;
    a.wsz = (wsz + 15) / 16 * 16;
    void* p = nullptr;
#define AL(dst, type, bytes) do { rc = nep_pool_alloc(&p, (size_t)(bytes)); if (rc) { a.release(); return rc; } dst = (type)p; } while (0)
    AL(a.Hdev, cplx*, (size_t)m * (m + 4) * 16);
    AL(a.wdev, cplx*, (size_t)m * (m + 2) * 16);
    AL(a.Ctab, cplx*, (size_t)m * mt * 16);
    AL(a.work3n, cplx*, (size_t)3 * n * 16);
    AL(a.active, int64_t*, (size_t)(m + 1) * 8);
    AL(a.v0, cplx*, (size_t)n * 16);
    AL(a.d_norms, double*, (size_t)m * 2 * m * 8);
    AL(a.eigwork, void*, (size_t)bmax * a.wsz);
#undef AL
    const size_t stage_bytes = std::max<size_t>((size_t)n * 16, (size_t)m * mt * 16);
    if (hipHostMalloc((void**)&a.Hpin, (size_t)m * (m + 4) * 16, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void**)&a.wpin, (size_t)m * (m + 2) * 16, hipHostMallocMapped) != hipSuccess ||
        hipHostMalloc((void**)&a.npin, (size_t)m * 2 * m * 8, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&a.stage, stage_bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); a.release();
        nep_set_error("nep_iar_run: pinned host allocation failed");
        return NEP_ERR_HIP;
    }
    *out = a;
    return NEP_OK;
}
void arena_return(Arena& a) {
    std::lock_guard<std::mutex> lk(g_arena_mu);
    if (g_arenas.size() >= 2) { Arena old = g_arenas.front(); g_arenas.erase(g_arenas.begin()); old.release(); }
    g_arenas.push_back(a);
}

// UMFPACK's stopping rule replayed on the recorded omegas of a step that took `plan` sweeps without looking
// (nonlineareigenproblems.jl_amd/linsolvers.py FactorizeLinSolver.review_recorded is the same rule for the step-at-a-time hosts;
// the rule itself: UMFPACK's umfpack_solve refinement loop behind `Afact \ x`, src/LinSolvers.jl:114-122)
struct Refine {
    int umf = 0; int recorded = -1; int hint = -1; bool hint_off = false; double last_omega = 0.0;
    int plan() const {
        if (umf <= 0) return 0;
        if (recorded >= 0) return recorded;
        return std::min(umf, hint < 0 ? 2 : hint);
    }
    bool settled() const { return recorded >= 0 || hint >= 0; }
    bool review(const double* w, int plan, bool final_recorded) {
        if (!final_recorded) {
            double w_prev = INFINITY;
            for (int step = 0; step < plan; ++step) {
                const double om = w[step];
                if (isfinite(om) && om <= 2.0 * EPS) { recorded = std::max(step, 1); return true; }
                if (!isfinite(om) || om > 0.5 * w_prev) { hint = -1; hint_off = true; return false; }
                w_prev = om;
            }
            return true;
        }
        double w_prev = INFINITY; int ret = -1;
        for (int step = 0; step <= umf; ++step) {
            if (step > plan) {
                if (isfinite(w[plan]) && w[plan] <= 4.0 * EPS) { ret = plan; break; }
                hint = -1; hint_off = true; return false;
            }
            const double om = w[step];
            if (om <= 2.0 * EPS) { ret = step; break; }
            if (om > 0.5 * w_prev) { ret = om > w_prev ? step - 1 : step; break; }
            if (step == umf) { ret = step; break; }
            w_prev = om;
        }
        if (ret < 0) ret = plan;                                   // (NaN omegas fall through every test)
        last_omega = w[plan];
        recorded = std::max(ret, 1);
        const bool ok = isfinite(w[plan]) && (ret == plan || w[plan] <= std::max(4.0 * EPS, w[ret]));
        if (ok) { if (!hint_off) hint = recorded; } else { hint = -1; hint_off = true; }
        return ok;
    }
};

struct Batch { std::vector<int> kcs; cplx* Zb = nullptr; int kmax = 0; hipEvent_t evW = nullptr, evZ = nullptr; };
struct Check { int kc = 0; std::vector<double> lam; cplx* QT = nullptr; std::vector<double> F; hipEvent_t ev = nullptr; cplx* Zb_owner = nullptr; };

inline cplx hc(const nep_cdouble& z) { cplx r; r.x = z.re; r.y = z.im; return r; }
// (a NotReady answer is recorded as the thread's last error by the runtime: taken off again, or the next launch check reports it)
inline bool ev_done(hipEvent_t e) {
    if (hipEventQuery(e) == hipSuccess) return true;
    (void)hipGetLastError();
    return false;
}

struct Run {
    // inputs
    nep_spmf* spmf; nep_lu* lu; int64_t n; nep_iar_opts o; nep_fv_eval fv; void* ctx; const double* h_fro;
    hipStream_t st; RunStreams rs; Arena a; bool have_arena = false;
    int32_t m, mt; int64_t ldv;
    cplx* V = nullptr; bool own_V = true; nep_iar* step = nullptr;
    Refine ref;
    std::vector<char> filled; std::vector<int> plans;
    std::vector<hipEvent_t> ev_pool;
    // state of the last consumed check
    std::vector<double> s_lam; std::vector<int> s_idx; cplx* s_QT = nullptr; int s_kq = 0; int conv = 0; int k_checked = 0;
    std::vector<double> s_err;
    double* h_err = nullptr;
    int retry = 0;

    hipEvent_t event() {
        if (!ev_pool.empty()) { hipEvent_t e = ev_pool.back(); ev_pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        return e;
    }
    void done(hipEvent_t e) { if (e) ev_pool.push_back(e); }

    // rows 1..kk of the recorded H block: breakdown / DGKS flags and the refinement record (the H entries themselves stay on the
    // device: the eigen-decompositions read them there)
    int fail_kind = 0, fail_at = 0;          // test hook NEP_IAR_RUN_FAIL_AT=kind:step (1 refinement record, 2 DGKS flag, 3 eig status)
    int fill_H(int kk) {
        for (int j = 1; j <= kk; ++j) {
            if (filled[j]) continue;
            if (fail_at == j && (fail_kind == 1 || fail_kind == 2)) {
                retry = fail_kind; nep_set_error("nep_iar_run: injected miss of kind %d at step %d", fail_kind, j); return NEP_ERR_RETRY;
            }
            const cplx* row = a.Hpin + (int64_t)(j - 1) * (m + 4);
            const int flags = (int)row[j + 1].y;
            if (flags & 2) { nep_set_error("orthogonalisation breakdown in step %d: ||w|| = %g", j, row[j].x); return NEP_ERR_BREAKDOWN; }
            if ((flags & 1) && o.orth_method == 0) { retry = 2; nep_set_error("nep_iar_run: step %d wanted another DGKS pass", j); return NEP_ERR_RETRY; }
            if (ref.umf > 0) {
                const int plan = plans[j];
                const bool final_rec = !((plan & 0x100) && (j % 8) != 0);
                if (!ref.review((const double*)(row + j + 2), plan & 0xff, final_rec)) {
                    retry = 1; nep_set_error("nep_iar_run: the refinement record of step %d asks for more sweeps than were taken", j);
                    return NEP_ERR_RETRY;
                }
            }
            filled[j] = 1;
        }
        return NEP_OK;
    }
    ~Run() {
        if (step) (void)nep_iar_destroy(step);
        if (V && own_V) nep_pool_free_on(V, st, true);
        if (s_QT) nep_pool_free_on(s_QT, rs.check, true);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        if (have_arena) arena_return(a);
    }
};

}  // namespace

extern "C" {

// UMFPACK's stopping rule on a recorded omega sequence (see struct Refine): for hosts that drive nep_iar_step themselves, and for
// the test that compares this restatement with the Python host's (linsolvers.py review_recorded)
int32_t nep_refine_review(int32_t umfpack_refinements, int32_t plan, int32_t final_recorded, const double* w4, int32_t hint_in,
                          int32_t out[4]) {
    ARGCHK(w4 && out && plan >= 0 && plan <= 3 && umfpack_refinements >= 0);
    Refine r; r.umf = umfpack_refinements; r.hint = hint_in;
    const bool ok = r.review(w4, plan, final_recorded != 0);
    out[0] = ok ? 1 : 0; out[1] = r.recorded; out[2] = r.hint; out[3] = r.hint_off ? 1 : 0;
    return NEP_OK;
}

int32_t nep_iar_run(nep_spmf* spmf, nep_lu* lu, int64_t n, const nep_iar_opts* opts, const nep_cdouble* h_v0,
                    const nep_cdouble* h_Ctab, int32_t mt, const double* h_cabs, const nep_cdouble* h_cf, const double* h_fro,
                    nep_fv_eval fv, void* ctx, nep_cdouble* h_lam, nep_cdouble* dQ, nep_cdouble* h_Q, double* h_err,
                    nep_cdouble* dV_basis, nep_iar_result* res, nep_stream stream) {
    ARGCHK(spmf && lu && opts && h_v0 && h_Ctab && fv && h_lam && res);
    ARGCHK(n > 0 && mt >= 1 && opts->maxit >= 1 && opts->check_error_every >= 1);
    ARGCHK(opts->orth_method == 0 || opts->orth_method == 1);
    ARGCHK(opts->errmeasure == 0 || (opts->errmeasure == 1 && h_fro));
    ARGCHK(opts->umfpack_refinements <= 0 || (h_cabs && h_cf));
    memset(res, 0, sizeof(*res));
    res->refine_plan = -1;
    struct timespec t_e_; clock_gettime(CLOCK_MONOTONIC, &t_e_);
    const double t_entry = t_e_.tv_sec * 1e3 + t_e_.tv_nsec * 1e-6;
    const int32_t m = opts->maxit;
    if (m > 128) { nep_set_error("nep_iar_run: maxit = %d exceeds the device eigen-decomposition's limit 128", m); return NEP_ERR_UNSUPPORTED; }
    const int BMAX = std::max(1, getenv("NEP_IAR_EIG_BATCH") ? atoi(getenv("NEP_IAR_EIG_BATCH")) : 16);
    const int LASTB = std::max(1, getenv("NEP_IAR_EIG_LAST") ? atoi(getenv("NEP_IAR_EIG_LAST")) : 8);
    const double T100 = getenv("NEP_IAR_EIG_MS100") ? atof(getenv("NEP_IAR_EIG_MS100")) : 3.3;
    const double TSTEP = getenv("NEP_IAR_EIG_MSSTEP") ? atof(getenv("NEP_IAR_EIG_MSSTEP")) : 0.35;
    const int cee = opts->check_error_every;
    const double neigs = opts->neigs;
    const bool unthrottled = isinf(neigs) && neigs > 0;

    Run R; R.spmf = spmf; R.lu = lu; R.n = n; R.o = *opts; R.fv = fv; R.ctx = ctx; R.h_fro = h_fro; R.st = as_stream(stream);
    R.m = m; R.mt = mt; R.ldv = n * (int64_t)(m + 1); R.h_err = h_err;
    int rc = run_streams(R.st, &R.rs); if (rc) return rc;
    rc = arena_acquire(n, m, mt, BMAX, &R.a); if (rc) return rc;
    R.have_arena = true;
    Arena& a = R.a;
    hipStream_t st = R.st, cst = R.rs.check, est = R.rs.eig;
    R.ref.umf = opts->umfpack_refinements > 0 ? opts->umfpack_refinements : 0;
    R.ref.hint = (opts->refine_hint >= 0 && !(getenv("NEP_REFINE_HINT") && atoi(getenv("NEP_REFINE_HINT")) == 0)) ? opts->refine_hint : -1;
    R.filled.assign(m + 1, 0); R.plans.assign(m + 1, 0);
    if (const char* fa = getenv("NEP_IAR_RUN_FAIL_AT")) { if (sscanf(fa, "%d:%d", &R.fail_kind, &R.fail_at) != 2) R.fail_kind = R.fail_at = 0; }
    if (h_err) for (int64_t i = 0; i < (int64_t)m * m; ++i) h_err[i] = NAN;

    // ---- set-up (method_iar.jl:76-86): basis, start vector, derivative table, H blocks
    if (dV_basis) { R.V = (cplx*)dV_basis; R.own_V = false; }
    else {
        void* p = nullptr;
        rc = nep_pool_alloc(&p, (size_t)(m + 1) * R.ldv * 16); if (rc) return rc;
        R.V = (cplx*)p;
    }
    double nrm2 = 0.0;
    for (int64_t i = 0; i < n; ++i) { a.stage[i] = hc(h_v0[i]); nrm2 += h_v0[i].re * h_v0[i].re + h_v0[i].im * h_v0[i].im; }
    ARGCHK(nrm2 > 0.0 && isfinite(nrm2));
    HIPCHK(hipMemcpyAsync(a.v0, a.stage, (size_t)n * 16, hipMemcpyHostToDevice, st));
    if (getenv("NEP_IAR_FULL_ZERO") && atoi(getenv("NEP_IAR_FULL_ZERO")))
        HIPCHK(hipMemsetAsync(R.V, 0, (size_t)(m + 1) * R.ldv * 16, st));
    else {
        if (getenv("NEP_IAR_POISON") && atoi(getenv("NEP_IAR_POISON"))) HIPCHK(hipMemsetAsync(R.V, 0xFF, (size_t)(m + 1) * R.ldv * 16, st));
        hipLaunchKernelGGL(k_iar_zero_slack, dim3((unsigned)(ZSLACK / 256), (unsigned)(m + 1)), dim3(256), 0, st, R.V, R.ldv, n, m + 1);
    }
    hipLaunchKernelGGL(k_iar_start, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, R.V, (const cplx*)a.v0, 1.0 / sqrt(nrm2), n);
    hipLaunchKernelGGL(k_iar_active, dim3((unsigned)((m + 1 + 63) / 64)), dim3(64), 0, st, a.active, n, m + 1);
    HIPCHK(hipMemsetAsync(a.Hdev, 0, (size_t)m * (m + 4) * 16, st));
    LAUNCHCHK();
    memset(a.Hpin, 0, (size_t)m * (m + 4) * 16);
    memset(a.wpin, 0, (size_t)m * (m + 2) * 16);
    {   // the table goes up through its own pinned block (the stage block is still being read by the start vector's copy)
        static thread_local PinnedRing ring;
        rc = ring.upload(a.Ctab, h_Ctab, (size_t)m * mt * 16, st); if (rc) return rc;
    }
    rc = nep_iar_create(spmf, lu, n, m, (nep_cdouble*)R.V, R.ldv, (const nep_cdouble*)a.Ctab, m, a.active, (nep_cdouble*)a.work3n,
                        R.ref.umf > 0 ? h_cabs : nullptr, R.ref.umf > 0 ? h_cf : nullptr, mt, (nep_cdouble*)a.Hdev,
                        (nep_cdouble*)a.Hpin, opts->orth_method, &R.step);
    if (rc) return rc;

    // ---- batch plan of the decompositions (see checker_dev of the Python host, iar.py, for the measurements behind it)
    std::vector<char> plan_end(m + 2, 0);
    if (unthrottled) {
        std::vector<int> allk;
        for (int kk = 1; kk <= m; ++kk) if (kk % cee == 0 || kk == m) allk.push_back(kk);
        int e_ = (int)allk.size(); int size = std::min(LASTB, e_);
        while (e_ > 0) {
            plan_end[allk[e_ - 1]] = 1; e_ -= size;
            if (e_ > 0) {
                const double q = allk[e_ - 1] / 100.0;
                size = (int)std::min<double>(std::min(BMAX, e_), std::max(1.0, ceil(2.0 * T100 * q * q / (TSTEP * cee))));
            }
        }
    }
    const bool trace = getenv("NEP_IAR_RUN_TRACE") != nullptr;
    auto now = []() { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; };
    double t_fv = 0;
    std::deque<int> pendA; std::deque<Batch> stA; std::deque<Check> stC;
    int slots = unthrottled ? m + 1 : std::max(BMAX, 4);
    const int CHUNK = unthrottled ? (getenv("NEP_IAR_BATCH") ? std::max(1, atoi(getenv("NEP_IAR_BATCH"))) : 8) : 4;
    const int P = std::min(256, std::max(1, 3072 / mt));        // panel width of nep_resid_batch_dev's output layout
    int k = 1; bool done = false; int status = NEP_OK;
    const bool use_graph = getenv("NEP_IAR_GRAPH") && atoi(getenv("NEP_IAR_GRAPH")) != 0;
    int n_graph = 0;
    auto finished = [&]() { return R.conv >= neigs; };

    auto batch_ready = [&]() -> int {
        if (pendA.empty()) return 0;
        int cnt = 1;
        while (cnt < (int)pendA.size() && cnt < BMAX && pendA[cnt] - pendA[cnt - 1] == pendA[1] - pendA[0] && !(unthrottled && plan_end[pendA[cnt - 1]])) ++cnt;
        if (!unthrottled || plan_end[pendA[cnt - 1]] || cnt >= BMAX || done) return cnt;
        return 0;
    };
    auto launch_batch = [&](int count) -> int {
        Batch b;
        for (int i = 0; i < count; ++i) { b.kcs.push_back(pendA.front()); pendA.pop_front(); }
        const int k0 = b.kcs[0], nb = count; b.kmax = b.kcs.back();
        const int kstep = nb > 1 ? b.kcs[1] - k0 : 0;
        int r = nep_iar_stream_wait(R.step, b.kmax, (nep_stream)est); if (r) return r;
        cplx* wrow = a.wdev + (int64_t)(k0 - 1) * (m + 2);
        cplx* mrow = a.wpin + (int64_t)(k0 - 1) * (m + 2);
        r = nep_hess_eigvals_batch_dev(nb, k0, kstep, (const nep_cdouble*)a.Hdev, m + 4, (nep_cdouble*)wrow, (int64_t)kstep * (m + 2),
                                       a.eigwork, a.wsz, (nep_cdouble*)mrow, (int64_t)kstep * (m + 2), (nep_stream)est);
        if (r) return r;
        b.evW = R.event(); if (!b.evW) return NEP_ERR_HIP;
        HIPCHK(hipEventRecord(b.evW, est));
        void* pz = nullptr;
        r = nep_pool_alloc(&pz, (size_t)nb * b.kmax * b.kmax * 16); if (r) return r;
        b.Zb = (cplx*)pz;
        r = nep_hess_eigvecs_batch_dev(nb, k0, kstep, (nep_cdouble*)wrow, (int64_t)kstep * (m + 2), (nep_cdouble*)b.Zb, b.kmax,
                                       (int64_t)b.kmax * b.kmax, a.eigwork, a.wsz, (nep_cdouble*)mrow, (int64_t)kstep * (m + 2), (nep_stream)est);
        if (r) { nep_pool_free_on(b.Zb, est, true); return r; }
        b.evZ = R.event(); if (!b.evZ) return NEP_ERR_HIP;
        HIPCHK(hipEventRecord(b.evZ, est));
        stA.push_back(b);
        return NEP_OK;
    };
    // (B) of one step: Ritz values, f_t(lambda), Ritz block, residual batch -- all on the check stream
    auto launch_check = [&](const Batch& b, int bi, bool first_of_batch) -> int {
        const int kc = b.kcs[bi];
        int r = R.fill_H(kc); if (r) return r;
        const cplx* w = a.wpin + (int64_t)(kc - 1) * (m + 2);
        if (w[kc].x != 0.0 || (R.fail_kind == 3 && R.fail_at == kc)) { R.retry = 3; nep_set_error("nep_iar_run: the QR iteration of step %d gave up", kc); return NEP_ERR_RETRY; }
        Check c; c.kc = kc; c.lam.resize(2 * (size_t)kc); c.F.resize(2 * (size_t)mt * kc);
        const double sr = R.o.sigma.re, si = R.o.sigma.im, gr = R.o.gamma.re, gi = R.o.gamma.im;
        for (int s = 0; s < kc; ++s) {                         // lambda = sigma + gamma / D   (method_iar.jl:116)
            const double dr = w[s].x, di = w[s].y, den = dr * dr + di * di;
            c.lam[2 * s] = sr + (gr * dr + gi * di) / den;
            c.lam[2 * s + 1] = si + (gi * dr - gr * di) / den;
        }
        const double tf0 = trace ? now() : 0.0;
        const int32_t frc = R.fv(R.ctx, kc, (const nep_cdouble*)c.lam.data(), (nep_cdouble*)c.F.data());
        if (trace) t_fv += now() - tf0;
        if (frc != 0) {
            nep_set_error("nep_iar_run: the f_t(lambda) callback failed in step %d", kc); return NEP_ERR_ARG;
        }
        if (first_of_batch) HIPCHK(hipStreamWaitEvent(cst, b.evZ, 0));
        void* pq = nullptr;
        r = nep_pool_alloc(&pq, (size_t)n * kc * 16); if (r) return r;
        c.QT = (cplx*)pq;
        r = nep_gemm_ts_dev((const nep_cdouble*)R.V, R.ldv, n, kc, (const nep_cdouble*)(b.Zb + (int64_t)bi * b.kmax * b.kmax), b.kmax, 0, kc,
                            (nep_cdouble*)c.QT, kc, 1, (nep_stream)cst);
        double* d_out = a.d_norms + (int64_t)(kc - 1) * 2 * m;
        if (!r) r = nep_resid_batch_dev(spmf, kc, (const nep_cdouble*)c.F.data(), (const nep_cdouble*)c.QT, kc, d_out, (nep_stream)cst);
        if (r) { nep_pool_free_on(c.QT, cst, true); return r; }
        HIPCHK(hipMemcpyAsync(a.npin + (int64_t)(kc - 1) * 2 * m, d_out, (size_t)2 * kc * 8, hipMemcpyDeviceToHost, cst));
        c.ev = R.event(); if (!c.ev) return NEP_ERR_HIP;
        HIPCHK(hipEventRecord(c.ev, cst));
        if (bi == (int)b.kcs.size() - 1) c.Zb_owner = b.Zb;     // the batch's eigenvector block is free once its last GEMM has run
        stC.push_back(std::move(c));
        return NEP_OK;
    };
    // (C): errors of step kc (errmeasure.jl:128-130,186-190), convergence count, sort (method_iar.jl:133-160)
    auto consume = [&](Check& c) -> int {
        const int kc = c.kc;
        const cplx* w = a.wpin + (int64_t)(kc - 1) * (m + 2);
        if (w[kc + 1].x != 0.0) { R.retry = 3; nep_set_error("nep_iar_run: an inverse iteration of step %d did not grow", kc); return NEP_ERR_RETRY; }
        const double* sq = a.npin + (int64_t)(kc - 1) * 2 * m;
        std::vector<double> e(kc);
        for (int j0 = 0; j0 < kc; j0 += P) {
            const int kk = std::min(P, kc - j0);
            for (int j = 0; j < kk; ++j) {
                const double rn = sqrt(sq[2 * j0 + j]), qn = sqrt(sq[2 * j0 + kk + j]);
                double den = qn;
                if (R.o.errmeasure == 1) {
                    double d = 0.0;
                    for (int t = 0; t < mt; ++t) d += R.h_fro[t] * hypot(c.F[2 * ((size_t)(j0 + j) * mt + t)], c.F[2 * ((size_t)(j0 + j) * mt + t) + 1]);
                    den *= d;
                }
                e[j0 + j] = rn / den;
            }
        }
        int conv = 0;
        for (int s = 0; s < kc; ++s) if (e[s] < R.o.tol) ++conv;
        std::vector<int> idx(kc);
        for (int s = 0; s < kc; ++s) idx[s] = s;
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) {     // NaN last, as sortperm / argsort place it
            const double ex = isnan(e[x]) ? INFINITY : e[x], ey = isnan(e[y]) ? INFINITY : e[y];
            return ex < ey;
        });
        R.s_err.assign(kc, 0.0);
        for (int s = 0; s < kc; ++s) R.s_err[s] = e[idx[s]];
        if (R.h_err) for (int s = 0; s < kc; ++s) R.h_err[(int64_t)s * m + (kc - 1)] = R.s_err[s];      // err[kc, s] of a column-major m x m
        int nret = kc;
        if (kc == m || conv >= neigs) nret = (int)std::min<double>(kc, neigs);
        R.s_lam.resize(2 * (size_t)nret); R.s_idx.assign(idx.begin(), idx.begin() + nret);
        for (int s = 0; s < nret; ++s) {
            const int src = (kc == m || conv >= neigs) ? idx[s] : s;
            R.s_lam[2 * s] = c.lam[2 * src]; R.s_lam[2 * s + 1] = c.lam[2 * src + 1];
        }
        if (!(kc == m || conv >= neigs)) for (int s = 0; s < nret; ++s) R.s_idx[s] = idx[s];
        if (R.s_QT) nep_pool_free_on(R.s_QT, cst, true);
        R.s_QT = c.QT; c.QT = nullptr; R.s_kq = kc;
        R.conv = conv; R.k_checked = kc;
        R.done(c.ev); c.ev = nullptr;
        return NEP_OK;
    };
    auto drop_check = [&](Check& c) {
        if (c.QT) nep_pool_free_on(c.QT, cst, true);
        if (c.Zb_owner) nep_pool_free_on(c.Zb_owner, cst, true);
        R.done(c.ev);
    };
    auto drop_batch = [&](Batch& b) {
        if (b.Zb) nep_pool_free_on(b.Zb, est, true);
        R.done(b.evW); R.done(b.evZ);
    };

    // ---- the cooperative loop
    hipEvent_t te0 = nullptr, te1 = nullptr; bool te1_rec = false;
    if (trace) { (void)hipEventCreate(&te0); (void)hipEventCreate(&te1); (void)hipEventRecord(te0, st); }
    const double t_loop = now();
    double tt_steps = 0, tt_A = 0, tt_B = 0, tt_C = 0, tt_sleep = 0, t_enq_done = 0; double tq = 0;
    while (status == NEP_OK) {
        bool progressed = false;
        tq = now();
        // recurrence: as many steps as there are free check slots (at most CHUNK) in one go
        if (!done) {
            if (k <= m && !finished()) {
                int nb = 0;
                while (nb < CHUNK && k + nb <= m) {
                    const bool due = ((k + nb) % cee == 0) || (k + nb == m);
                    if (due) { if (slots <= 0) break; --slots; }
                    ++nb;
                }
                if (nb > 0) {
                    int plan = R.ref.plan();
                    if (plan > 0 && R.ref.settled() && !getenv("NEP_IAR_RECORD_ALL")) plan |= 0x100;
                    // chunks after the first as hipGraphs (NEP_IAR_GRAPH, default off for the NULL stream: it cannot be captured)
                    if (use_graph && k > 1) { int32_t cap = 0; status = nep_iar_steps_graph(R.step, k, nb, plan, stream, &cap); n_graph += cap; }
                    else status = nep_iar_steps(R.step, k, nb, plan, stream);
                    if (status) break;
                    for (int kk = k; kk < k + nb; ++kk) {
                        R.plans[kk] = plan;
                        if (kk % cee == 0 || kk == m) pendA.push_back(kk);
                    }
                    k += nb; progressed = true;
                }
            } else done = true;
            if (k > m) { if (!done && trace) { t_enq_done = now() - t_loop; if (te1 && !te1_rec) { (void)hipEventRecord(te1, st); te1_rec = true; } } done = true; }
        }
        if (trace) { const double t = now(); tt_steps += t - tq; tq = t; }
        // (A)
        while (!finished()) {
            const int cnt = batch_ready();
            if (!cnt) break;
            status = launch_batch(cnt); progressed = true;
            if (status) break;
        }
        if (status) break;
        if (trace) { const double t = now(); tt_A += t - tq; tq = t; }
        // (B)
        while (!stA.empty() && !finished() && ev_done(stA.front().evW)) {
            progressed = true;
            Batch b = stA.front(); stA.pop_front();
            for (int bi = 0; bi < (int)b.kcs.size() && status == NEP_OK; ++bi) {
                status = launch_check(b, bi, bi == 0);
                if (status == NEP_OK) ++slots;
            }
            if (status) { drop_batch(b); break; }
            R.done(b.evW); R.done(b.evZ);          // (the block Zb now travels with the batch's last check)
        }
        if (status) break;
        if (trace) { const double t = now(); tt_B += t - tq; tq = t; }
        // (C)
        while (!stC.empty() && !finished() && ev_done(stC.front().ev)) {
            progressed = true;
            Check c = std::move(stC.front()); stC.pop_front();
            status = consume(c);
            if (c.Zb_owner) nep_pool_free_on(c.Zb_owner, cst, true);
            c.Zb_owner = nullptr;
            if (status) { drop_check(c); break; }
        }
        if (status) break;
        if (finished()) {
            done = true; pendA.clear();
            while (!stA.empty()) { drop_batch(stA.front()); stA.pop_front(); }
            while (!stC.empty()) { drop_check(stC.front()); stC.pop_front(); }
        }
        if (trace) { const double t = now(); tt_C += t - tq; tq = t; }
        if (done && pendA.empty() && stA.empty() && stC.empty()) break;
        if (!progressed) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); if (trace) { const double t = now(); tt_sleep += t - tq; tq = t; } }
    }
    if (trace)
        fprintf(stderr, "nep_iar_run trace (ms): set-up %.2f | loop %.2f: steps %.2f (all enqueued at %.2f) A %.2f B %.2f (callback %.2f) C %.2f sleep %.2f\n",
                t_loop - t_entry, now() - t_loop, tt_steps, t_enq_done, tt_A, tt_B, t_fv, tt_C, tt_sleep);
    if (trace && use_graph) fprintf(stderr, "nep_iar_run: %d chunks went out as hipGraphs\n", n_graph);
    if (trace && te0 && te1 && te1_rec) {
        float ms_ = 0.0f;
        if (hipEventSynchronize(te1) == hipSuccess && hipEventElapsedTime(&ms_, te0, te1) == hipSuccess)
            fprintf(stderr, "nep_iar_run: the recurrence's %d steps took %.2f ms on the device\n", m, ms_);
        (void)hipGetLastError();
    }
    if (te0) (void)hipEventDestroy(te0);
    if (te1) (void)hipEventDestroy(te1);
    // whatever is still queued (a failure, or speculative work beyond convergence) must not outlive the blocks it uses
    while (!stA.empty()) { drop_batch(stA.front()); stA.pop_front(); }
    while (!stC.empty()) { drop_check(stC.front()); stC.pop_front(); }
    (void)hipStreamSynchronize(est);
    (void)hipStreamSynchronize(cst);
    res->retry_reason = R.retry;
    res->refine_plan = R.ref.hint_off ? -1 : (R.ref.recorded >= 0 ? R.ref.recorded : R.ref.hint);
    res->refine_hint_off = R.ref.hint_off ? 1 : 0;
    // speculative steps beyond the converged one (finite neigs) still write the arena's blocks: drained before it changes hands
    if (status || !unthrottled) (void)hipStreamSynchronize(st);
    if (status) return status;

    // ---- what the reference returns (method_iar.jl:162-181)
    int nret = (int)(R.s_lam.size() / 2);
    const bool noconv = (R.conv < neigs) && !unthrottled;
    if (!noconv) nret = std::min(nret, R.conv);
    res->k = R.k_checked > 0 ? R.k_checked : k - 1;
    res->nconv = R.conv; res->nret = nret;
    for (int s = 0; s < nret; ++s) { h_lam[s].re = R.s_lam[2 * s]; h_lam[s].im = R.s_lam[2 * s + 1]; }
    if (nret > 0 && (dQ || h_Q) && R.s_QT) {
        cplx* dst = (cplx*)dQ; void* tmp = nullptr;
        if (!dst) { rc = nep_pool_alloc(&tmp, (size_t)n * nret * 16); if (rc) return rc; dst = (cplx*)tmp; }
        std::vector<int32_t> cols(R.s_idx.begin(), R.s_idx.begin() + nret);
        rc = nep_rowmajor_to_colmajor(n, R.s_kq, (const nep_cdouble*)R.s_QT, R.s_kq, cols.data(), nret, (nep_cdouble*)dst, n, (nep_stream)cst);
        if (!rc && h_Q) {
            if (hipMemcpyAsync(h_Q, dst, (size_t)n * nret * 16, hipMemcpyDeviceToHost, cst) != hipSuccess) rc = NEP_ERR_HIP;
        }
        if (hipStreamSynchronize(cst) != hipSuccess && !rc) rc = NEP_ERR_HIP;
        if (tmp) nep_pool_free(tmp);
        if (rc) { if (rc == NEP_ERR_HIP) nep_set_error("nep_iar_run: returning the eigenvector block failed"); return rc; }
    }
    if (noconv) { nep_set_error("Number of iterations exceeded. maxit=%d.", m); return NEP_ERR_NOCONV; }
    return NEP_OK;
}

}  // extern "C"
