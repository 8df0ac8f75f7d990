// C ABI of libtheta_hip.so (see include/theta_hip.h).  Host-side glue only: argument checks,
// HBM residency of a search instance, kernel launches on the context's stream, HIP-event timing,
// and the merge of the device tie list.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <memory>

#include "n2.hpp"
#include "n3_core.hpp"
#include "n3_sieve.hpp"
#include "bnb.hpp"

// ---- implemented in n2.hip / n3.hip / batch.hip ------------------------------------------------
void n2_launch_search(const N2Dev &P, const SearchArgs &A, unsigned long long begin, unsigned long long end,
                      int per_thread, hipStream_t st, unsigned long long sample_stride = 0);
void n2_launch_enumerate(const N2Dev &P, unsigned long long begin, unsigned long long count, unsigned char *out,
                         hipStream_t st);
void n2_launch_unrank_list(const N2Dev &P, const TieRecord *recs, int count, unsigned char *out, hipStream_t st);

int n3_build_host(int m, int tau, const int32_t *lb_in, const int32_t *ub_in, N3Host &h);
int n3_run_dp(const N3Dev &P, u128 *cnt, unsigned *overflow_dev, unsigned long long *total_dev, hipStream_t st);
void n3_launch_tasks(const N3Dev &P, u128 begin, u128 end, uint64_t per_task, int ntasks, N3Task *tasks,
                     unsigned *stbuf, hipStream_t st);
void n3_launch_task_list(const N3Dev &P, const uint64_t *spec, int ntasks, N3Task *tasks, unsigned *stbuf, hipStream_t st);
void n3_launch_search(const N3Dev &P, const SearchArgs &A, const N3Task *tasks, const unsigned *stbuf, int ntasks,
                      uint64_t per_task, hipStream_t st);
void n3_launch_unrank_list(const N3Dev &P, const TieRecord *recs, int count, unsigned char *out, hipStream_t st);
int n3_enumerate_burst_levels(const N3Dev &P);
void n3_launch_enumerate_burst(const N3Dev &P, const N3Task *tasks, const unsigned *stbuf, int ntasks, uint64_t per_task,
                               unsigned char *out, hipStream_t st);
void n3_launch_enumerate(const N3Dev &P, const N3Task *tasks, const unsigned *stbuf, int ntasks, uint64_t per_task,
                         unsigned char *out, hipStream_t st);
void n3_launch_nan_scan(const unsigned char *ok, const double *nll, unsigned long long count, u128 base, double near, SearchCounters *ctr,
                        TieRecord *deg, unsigned deg_cap, hipStream_t st);
void n3_launch_collinear_scan(const unsigned char *C, unsigned long long count, int m, u128 base, SearchCounters *ctr, TieRecord *deg,
                              unsigned deg_cap, hipStream_t st);

void batch_launch_solve(int n, int m, int tau, const double *r, const double *rN, double max_normal, int B,
                        const unsigned char *C, unsigned char *ok, double *mu, double *nll, double *vals,
                        hipStream_t st);
void batch_launch_score(int n, int m, int B, const double *Cw, const double *mu, const double *r, int r_stride, double *nll,
                        double *vals, unsigned char *valid, hipStream_t st);
void batch_launch_boundary_min(int m, int tau, const double *r, const double *rN, int B, const unsigned char *C,
                               double *bound, hipStream_t st);
void batch_launch_score_masked(int n, int m, int tau, int B, int S, const unsigned char *C, const double *w,
                               const double *r, const double *mu, const unsigned long long *mask, double *nll,
                               double *rsum_scratch, hipStream_t st, double rsum_host, bool rsum_host_valid, double rlogw_host,
                               bool rlogw_valid, double wsum_host, double wmin_host, double wmax_host);

// ---- error string ---------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void theta_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *theta_last_error(void) { return g_err; }

// ---- context -------------------------------------------------------------------------------------
extern "C" int theta_create(int device_id, theta_ctx **out) {
    if (!out) {
        theta_set_error("theta_create: null output pointer");
        return THETA_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        theta_set_error("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return THETA_ERR_HIP;
    }
    if (device_id < 0 || device_id >= ndev) {
        theta_set_error("device %d out of range (%d devices)", device_id, ndev);
        return THETA_ERR_ARG;
    }
    HIP_ENTER(device_id);
    theta_ctx *c = new theta_ctx();
    c->device = device_id;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    c->cu_count = prop.multiProcessorCount;
    c->hbm_bytes = prop.totalGlobalMem;
    snprintf(c->name, sizeof(c->name), "%s (%s)", prop.name, prop.gcnArchName);
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipEventCreate(&c->ev2));
    *out = c;
    return THETA_OK;
}

extern "C" int theta_device_count(int *n_out) {
    if (!n_out) {
        theta_set_error("theta_device_count: null output pointer");
        return THETA_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    *n_out = e == hipSuccess ? ndev : 0;
    if (e != hipSuccess || ndev == 0) {
        (void)hipGetLastError();
        theta_set_error("no HIP device available (%s)", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return THETA_ERR_HIP;
    }
    return THETA_OK;
}

extern "C" void theta_destroy(theta_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipEventDestroy(c->ev0);
    (void)hipEventDestroy(c->ev1);
    (void)hipEventDestroy(c->ev2);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" int theta_synchronize(theta_ctx *c) {
    if (!c) {
        theta_set_error("null context");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(c->device);
    HIP_TRY(hipDeviceSynchronize());
    return THETA_OK;
}

extern "C" int theta_device_info(theta_ctx *c, char *name, int cap, int *cu, uint64_t *hbm_bytes) {
    if (!c) {
        theta_set_error("null context");
        return THETA_ERR_ARG;
    }
    if (name && cap > 0) snprintf(name, cap, "%s", c->name);
    if (cu) *cu = c->cu_count;
    if (hbm_bytes) *hbm_bytes = c->hbm_bytes;
    return THETA_OK;
}

// ---- search instance ------------------------------------------------------------------------------
#define LIST_CAP (1u << 20)
#define SUS_CAP (1u << 20)
#define DEG_CAP (1u << 20)
#define LINE_CAP (1u << 19)      // n=3 sieve: tasks per call that may report a prefix with collinear rows (a call holds at most 2^18 tasks; redone slices report again)
#define SURV_CAP (1u << 24)           /* contenders per slice of the sieve (2.4 GB of the 288 GB, allocated on first use) */
#define SIEVE_SLICE (1ull << 31)     /* candidates per sieve launch; the finish kernel runs in between and lowers the minimum */
#define SIEVE_MAX_SLICES 64

struct theta_problem {
    theta_ctx *ctx = nullptr;
    int n = 0, m = 0, tau = 0;
    double max_normal = 1.0;
    N2Host n2h;
    N2Dev n2{};
    N3Host n3h;
    N3Dev n3{};
    uint64_t total[2] = {0, 0};
    std::vector<TieRecord> suspects;   // rejected candidates near the minimum, from the last theta_search
    uint64_t suspects_dropped = 0;     // ... and how many more did not fit the device list
    std::vector<TieRecord> degenerate; // n=3 rank-deficient candidates (collinear rows; all-zero tumour columns among them), from the last theta_search
    uint64_t degenerate_dropped = 0;
    double hint = INFINITY;            // upper bound of the minimum known to the caller (theta_problem_hint), one-shot
    uint64_t opt_per_task = 0;         // n=3 candidates per wave task (0: automatic), theta_problem_set_option
    int opt_per_thread = 0;            // n=2 candidates per thread (0: automatic)
    int opt_nan_sweep = 0;             // n=3: after a search, every candidate of the range through the reference's own procedure; the ones it
                                       // reports with a NaN likelihood join the degenerate list (nan_sweep below; 2e8-5e8 candidates/s)
    int opt_sieve = 1;                 // n=3: sieve + finish kernels (n3_sieve.hip); 0 = the fused kernel of n3.hip only
    int auto64 = 0;                    // n=3 sieve: 1 = the packed-FP32 screen listed too many contenders on this problem (its margin, 2e-5 sum r + 1,
                                       // is coarse against the spread of the NLL in a range): later calls run the double instantiation
    bool opt_auto64 = true;            // ... unless switched off (option "n3_auto_f64")
    bool count_saturated = false;      // n=3: the space holds 2^128 matrices or more (total = 2^128 - 1)
    bool mix_only = false;             // n=3: more than 64 rows within the bounds -- no ranks: theta_mix_search and the batch operators only
    bool table_pending = false;        // n=3: the counting table is built at the first call that takes ranks (ensure_table): the space provably holds
                                       // 2^128 matrices or more, so its count is known without it
    unsigned opt_surv_cap = 0;         // n=3: contenders a slice may list before it counts as overflowed (0: SURV_CAP; smaller
                                       // values make the tests walk the redo ladder: sieve again -> 8 parts -> fused kernel)
    int device = 0;                                     // (= ctx->device: the destructor must not need the context)
    uint64_t last_survivors = 0, last_fallback = 0;   // of the last search: contenders listed by the sieve / candidates redone fused
    SearchCounters last_redo{};                        // ... what the fused kernel did on the redone slices (kept apart from the main counters)
    double last_redo_ms = 0.0;
    bool last_sieve64 = false;                         // ... and whether the sieve ran in FP64 (n3_force_f64)
    uint64_t last_launches = 0;                        // launches of the search kernel behind kernel_ms (sieve: one per slice)
    std::vector<double> h_r, h_rN;                     // the counts as given (sorted order): the per-depth constants of theta_bnb
    DevBuf d_misc2;                                    // task specifications of a search over several ranges
    std::vector<uint64_t> boot_spec;                   // ... and of the small first slice of a hinted n=3 call (kept: the copy is asynchronous)
    // the mixture-space search (theta_mix_search): its buffers, allocated at the first call and kept; the lines of the alphabet's grid
    DevBuf d_mix_stack, d_mix_work, d_mix_leaves, d_mix_ctr, d_mix_mat, d_mix_lines, d_mix_slot, d_mix_iv, d_mix_seen;
    unsigned long long mix_seen_mask = 0;
    std::vector<MixLine> mix_lines;
    unsigned mix_chunk = 0;
    unsigned long long mix_stack_cap = 0, mix_leaf_cap = 0;
    uint64_t mix_mat_cap = 0;
    int opt_mix_g = 0, opt_mix_G = 1;                  // options mix_shard_rank / mix_shard_world: this rank's share of the boxes
    int opt_mix_niches = 1;                            // option mix_dive_niches
    double opt_mix_blend = 0.2;                        // option mix_dive_blend
    uint64_t opt_mix_beam = 512;                       // option mix_beam: boxes a dive (THETA_MIX_DIVE) keeps per level
    uint64_t opt_mix_max_steps = 1ull << 22;           // option mix_max_steps: steps one (leaf, corner) walk over the intervals may take
    double opt_mix_max_ms = 0.0;                       // option mix_max_ms: give up (THETA_ERR_CAPACITY) beyond this much wall time (0: never)
    uint64_t opt_mix_max_boxes = 0;                    // option mix_max_boxes: give up (THETA_ERR_CAPACITY) beyond this many boxes tested (0: never)
    DevBuf d_r, d_rN, d_small, d_P, d_PR, d_PN, d_cnt, d_ctr, d_stat, d_list, d_tasks, d_stbuf, d_misc, d_smask, d_dynmask, d_sus, d_deg, d_surv, d_survcnt, d_survacc, d_line, d_scan, d_sweep;
};

static int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t st) {
    int rc = b.alloc(bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, st));
    return THETA_OK;
}

extern "C" void theta_problem_destroy(theta_problem *p) {
    if (!p) return;
    // (p->device, not p->ctx->device: a garbage collector may destroy the context first -- round 3: a dangling read here set
    // the device to garbage, and the error stayed in HIP's per-thread state until the next hipGetLastError() of a search)
    (void)hipSetDevice(p->device);
    delete p;
    (void)hipGetLastError();
}

// A lower bound of the number of matrices of an n = 3 space, on the host in a millisecond: the matrices along which a + b never
// decreases.  Every one of them is a matrix Enumerator._generate_next_C_3 yields, given valid rows within the adjusted bounds
// and the symmetry rule (a <= b in the first off-diagonal row, Enumerator.py:181-183, 199-202): between consecutive rows either
// nothing changes or some component increases (:258-260), and the ratio window (:225-239) always holds the ratio 1 -- a step with
// dx > 0 > dy and dx + dy >= 0 raises its lower end to |dy| / dx <= 1, a step with dx < 0 < dy lowers its upper end to dy / |dx| >= 1,
// all others leave it alone.  Unlike the matrices that are monotone in both components (polynomially many in m) this family is
// exponential: 4^m on the anti-diagonal a + b = 7 alone.  log2 of the count, long double (1e4932 is room enough for any m <= 256).
static double n3_count_lower_bound_log2(const N3Host &h, int m, int tau) {
    struct Row { int a, b; };
    std::vector<Row> rows;
    for (int a = 0; a <= h.K; a++)
        for (int b = 0; b <= h.K; b++)
            if ((tau - a) * (tau - b) >= 0) rows.push_back({a, b});
    const int R = (int)rows.size(), S = 2 * h.K + 1;
    auto in = [&](int i, const Row &r) { return r.a >= h.lb[i] && r.a <= h.ub[i] && r.b >= h.lb[i] && r.b <= h.ub[i]; };
    // f[0][r]: sequences ending in row r whose rows so far all have a == b (the symmetry rule still binds); f[1][r]: the others
    std::vector<long double> f0(R, 0.0L), f1(R, 0.0L), c0(S + 1), c1(S + 1);
    for (int r = 0; r < R; r++)
        if (in(0, rows[r]) && rows[r].a <= rows[r].b) (rows[r].a == rows[r].b ? f0 : f1)[r] = 1.0L;
    for (int i = 1; i < m; i++) {
        // cumulative over a + b: c[s + 1] = sum of f over the rows with a + b <= s
        std::fill(c0.begin(), c0.end(), 0.0L);
        std::fill(c1.begin(), c1.end(), 0.0L);
        for (int r = 0; r < R; r++) {
            c0[rows[r].a + rows[r].b + 1] += f0[r];
            c1[rows[r].a + rows[r].b + 1] += f1[r];
        }
        for (int s = 0; s < S; s++) {
            c0[s + 1] += c0[s];
            c1[s + 1] += c1[s];
        }
        for (int r = 0; r < R; r++) {
            const Row &w = rows[r];
            const long double below0 = c0[w.a + w.b + 1], below1 = c1[w.a + w.b + 1];
            f0[r] = f1[r] = 0.0L;
            if (!in(i, w)) continue;
            if (w.a == w.b) {
                f0[r] = below0;                       // still all-diagonal
                f1[r] = below1;
            } else {
                f1[r] = below1 + (w.a < w.b ? below0 : 0.0L);      // the first off-diagonal row needs a < b
            }
        }
    }
    long double tot = 0.0L;
    for (int r = 0; r < R; r++) tot += f0[r] + f1[r];
    return tot > 0.0L ? (double)log2l(tot) : -INFINITY;
}

extern "C" int theta_count_lower_bound(int m, int tau, const int32_t *lb, const int32_t *ub, double *log2_count) {
    if (m < 1 || m > N3_MAX_M_WIDE || !lb || !ub || !log2_count) {
        theta_set_error("theta_count_lower_bound: bad argument");
        return THETA_ERR_ARG;
    }
    N3Host h;
    h.m = m;
    h.lb.assign(lb, lb + m);
    h.ub.assign(ub, ub + m);
    for (int i = 1; i < m; i++)                       // Enumerator._check_bound_order (Enumerator.py:90-113), as n3_build_host does
        if (h.lb[i] < h.lb[i - 1]) h.lb[i] = h.lb[i - 1];
    for (int i = m - 2; i >= 0; i--)
        if (h.ub[i] > h.ub[i + 1]) h.ub[i] = h.ub[i + 1];
    h.K = 0;
    for (int i = 0; i < m; i++) {
        if (h.lb[i] < 0 || h.ub[i] > N3_MAX_COPY) {
            theta_set_error("theta_count_lower_bound: bounds outside [0, %d]", N3_MAX_COPY);
            return THETA_ERR_ARG;
        }
        h.K = std::max(h.K, h.ub[i]);
    }
    *log2_count = n3_count_lower_bound_log2(h, m, tau);
    return THETA_OK;
}

// The counting table of an n = 3 problem whose creation deferred it (table_pending): built now, before the first kernel that reads it.
static int ensure_table(theta_problem *p) {
    if (!p || !p->table_pending) return THETA_OK;
    HIP_ENTER(p->ctx->device);
    hipStream_t st = p->ctx->stream;
    N3Dev &D = p->n3;
    const size_t per_level = (size_t)D.Q * 2 * (D.NT + 1) * (D.NT + 1);
    int rc = p->d_cnt.alloc(per_level * p->m * sizeof(u128));
    if (rc) return rc;
    D.cnt = (const u128 *)p->d_cnt.p;
    HIP_TRY(hipMemsetAsync(p->d_misc.p, 0, 64, st));
    n3_run_dp(D, (u128 *)p->d_cnt.p, (unsigned *)p->d_misc.p, (unsigned long long *)((char *)p->d_misc.p + 16), st);
    unsigned char hostmisc[64];
    HIP_TRY(hipMemcpyAsync(hostmisc, p->d_misc.p, 64, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    unsigned hov;
    memcpy(&hov, hostmisc, 4);
    if (!hov) {        // (the host's lower bound said 2^128 or more: the table must have saturated)
        theta_set_error("internal: the counting table of a space counted as saturated did not saturate");
        return THETA_ERR_HIP;
    }
    p->table_pending = false;
    return THETA_OK;
}

extern "C" int theta_problem_create(theta_ctx *ctx, int n, int m, int tau, const int64_t *r, const int64_t *rN,
                                    const int32_t *lb, const int32_t *ub, double max_normal, theta_problem **out) {
    if (!ctx || !r || !rN || !lb || !ub || !out) {
        theta_set_error("theta_problem_create: null argument");
        return THETA_ERR_ARG;
    }
    if (n != 2 && n != 3) {
        theta_set_error("n must be 2 or 3 (got %d)", n);
        return THETA_ERR_ARG;
    }
    int max_m = (n == 2) ? THETA_MAX_M : N3_MAX_M_WIDE;
    if (m < 2 || m > max_m) {
        theta_set_error("m must be in [2, %d] for n=%d (got %d)", max_m, n, m);
        return THETA_ERR_ARG;
    }
    if (tau < 0 || tau > THETA_MAX_COPY) {
        theta_set_error("tau out of range");
        return THETA_ERR_ARG;
    }
    if (!(max_normal >= 0.0 && max_normal <= 1.0)) {
        theta_set_error("max_normal must be in [0,1]");
        return THETA_ERR_ARG;
    }
    std::vector<double> rd(m), rnd(m);
    long double N = 0, Rt = 0;
    for (int i = 0; i < m; i++) {
        if (rN[i] <= 0 || r[i] < 0) {
            theta_set_error("interval %d: need normal count > 0 and tumour count >= 0 (got %lld, %lld)", i,
                            (long long)rN[i], (long long)r[i]);
            return THETA_ERR_ARG;
        }
        rd[i] = (double)r[i];
        rnd[i] = (double)rN[i];
        N += rN[i];
        Rt += r[i];
    }
    if (N >= 9.0e15L || Rt >= 9.0e15L) {
        theta_set_error("read-count totals exceed 2^53");
        return THETA_ERR_ARG;
    }
    long double k0 = 0;
    for (int i = 0; i < m; i++)
        if (r[i] > 0) k0 -= (long double)r[i] * logl((long double)rN[i] / N);

    HIP_ENTER(ctx->device);
    std::unique_ptr<theta_problem> owner(new theta_problem());   // freed (with all its device buffers) on every error return
    theta_problem *p = owner.get();
    p->ctx = ctx;
    p->device = ctx->device;
    p->n = n;
    p->m = m;
    p->tau = tau;
    p->max_normal = max_normal;
    hipStream_t st = ctx->stream;
    int rc;
#define TRY(x)          \
    do {                \
        rc = (x);       \
        if (rc) return rc; \
    } while (0)
    p->h_r = rd;
    p->h_rN = rnd;
    TRY(upload(p->d_r, rd.data(), m * sizeof(double), st));
    TRY(upload(p->d_rN, rnd.data(), m * sizeof(double), st));
    TRY(p->d_ctr.alloc(sizeof(SearchCounters)));
    TRY(p->d_stat.alloc((size_t)THETA_STAT_SLOTS * THETA_STAT_STRIDE));
    TRY(p->d_list.alloc((size_t)LIST_CAP * sizeof(TieRecord)));
    TRY(p->d_sus.alloc((size_t)SUS_CAP * sizeof(TieRecord)));
    TRY(p->d_deg.alloc((size_t)DEG_CAP * sizeof(TieRecord)));
    if (n == 3) TRY(p->d_line.alloc((size_t)LINE_CAP * 2 * sizeof(unsigned long long)));

    if (n == 2) {
        TRY(n2_build_host(m, lb, ub, p->n2h));
        const N2Host &h = p->n2h;
        // small arrays: lb[m] ub[m] (bytes) then lbpos (shorts)
        std::vector<unsigned char> small(2 * (size_t)m + 64, 0);
        for (int i = 0; i < m; i++) {
            small[i] = (unsigned char)h.lb[i];
            small[m + i] = (unsigned char)h.ub[i];
        }
        size_t off = ((2 * (size_t)m + 7) / 8) * 8;
        small.resize(off + (N2_KVS + 1) * sizeof(short));
        short *lbpos = (short *)(small.data() + off);
        for (int v = 0; v <= N2_KVS; v++) {
            int pos = m;
            for (int i = 0; i < m; i++)
                if (h.lb[i] >= v) {
                    pos = i;
                    break;
                }
            lbpos[v] = (short)pos;
        }
        TRY(upload(p->d_small, small.data(), small.size(), st));
        TRY(upload(p->d_P, h.P.data(), h.P.size() * sizeof(unsigned long long), st));
        std::vector<double> PR(m + 1, 0.0), PN(m + 1, 0.0);
        for (int i = 0; i < m; i++) {
            PR[i + 1] = PR[i] + rd[i];
            PN[i + 1] = PN[i] + rnd[i];
        }
        TRY(upload(p->d_PR, PR.data(), PR.size() * sizeof(double), st));
        TRY(upload(p->d_PN, PN.data(), PN.size() * sizeof(double), st));
        N2Dev &D = p->n2;
        D.m = m;
        D.kv = h.kv;
        D.tau = tau;
        D.max_normal = max_normal;
        D.N = (double)N;
        D.Rtot = (double)Rt;
        D.K0 = (double)k0;
        D.PR = (const double *)p->d_PR.p;
        D.PN = (const double *)p->d_PN.p;
        D.P = (const unsigned long long *)p->d_P.p;
        D.lb = (const unsigned char *)p->d_small.p;
        D.ub = D.lb + m;
        D.lbpos = (const short *)((const unsigned char *)p->d_small.p + off);
        D.total = h.total;
        D.quick = 1;
        if (const char *e = getenv("THETA_N2_NO_DISMISS")) D.quick = atoi(e) == 0;
        if (const char *e = getenv("THETA_N2_QUICK_FLAGS")) D.quick = atoi(e);      // (experiments)
        D.first_zero_r = m;
        for (int i = m - 1; i >= 0; i--)
            if (r[i] == 0) D.first_zero_r = i;
        p->total[0] = h.total;
        p->total[1] = 0;
    } else {
        const auto tc0 = std::chrono::steady_clock::now();
        auto tc_say = [&](const char *what) {
            if (getenv("THETA_CREATE_DEBUG"))
                fprintf(stderr, "create: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
        };
        TRY(n3_build_host(m, tau, lb, ub, p->n3h));
        tc_say("host tables");
        const N3Host &h = p->n3h;
        if (h.mix_only) {
            // more than 64 rows (a, b) within the bounds (full bounds [0, 8] and beyond): no child masks, no counting table, no ranks.
            // What remains: the row table and the bounds on the device for theta_mix_search; the space counts as "2^128 or more".
            std::vector<unsigned char> small(2 * (size_t)m + h.rowtab.size(), 0);
            for (int i = 0; i < m; i++) {
                small[i] = (unsigned char)h.lb[i];
                small[m + i] = (unsigned char)h.ub[i];
            }
            memcpy(small.data() + 2 * m, h.rowtab.data(), h.rowtab.size());
            TRY(upload(p->d_small, small.data(), small.size(), st));
            N3Dev &D = p->n3;
            memset(&D, 0, sizeof(D));
            D.m = m;
            D.K = h.K;
            D.Q = h.Q;
            D.tau = tau;
            D.N = (double)N;
            D.Rtot = (double)Rt;
            D.K0 = (double)k0;
            D.r = (const double *)p->d_r.p;
            D.rN = (const double *)p->d_rN.p;
            D.lb = (const unsigned char *)p->d_small.p;
            D.ub = D.lb + m;
            D.rowtab = D.lb + 2 * m;
            p->mix_only = true;
            p->count_saturated = true;
            p->total[0] = p->total[1] = ~0ull;
            p->opt_sieve = 0;
            HIP_TRY(hipStreamSynchronize(st));
            *out = owner.release();
            return THETA_OK;
        }
        std::vector<unsigned char> small(2 * (size_t)m + h.ridx.size() + h.rowtab.size(), 0);
        for (int i = 0; i < m; i++) {
            small[i] = (unsigned char)h.lb[i];
            small[m + i] = (unsigned char)h.ub[i];
        }
        memcpy(small.data() + 2 * m, h.ridx.data(), h.ridx.size());
        memcpy(small.data() + 2 * m + h.ridx.size(), h.rowtab.data(), h.rowtab.size());
        TRY(upload(p->d_small, small.data(), small.size(), st));
        TRY(upload(p->d_smask, h.smask.data(), h.smask.size() * sizeof(unsigned long long), st));
        {   // the ratio masks, and behind them the full ratio-rank table once more (n3_sieve.hip reads it there: sv_child_dyn)
            std::vector<unsigned long long> dm(h.dynmask.size() + (h.ridx.size() + 7) / 8, 0ull);
            memcpy(dm.data(), h.dynmask.data(), h.dynmask.size() * sizeof(unsigned long long));
            memcpy(dm.data() + h.dynmask.size(), h.ridx.data(), h.ridx.size());
            TRY(upload(p->d_dynmask, dm.data(), dm.size() * sizeof(unsigned long long), st));
        }
        N3Dev &D = p->n3;
        D.m = m;
        D.K = h.K;
        D.Q = h.Q;
        D.tau = tau;
        D.NT = h.NT;
        D.L = 1;   // (decided below, once the candidate count is known; the counting DP does not depend on it)
        D.warm_blend = 0.9;
        D.mu_tol = 0.0;
        D.conv_l2 = 1e-4;    // first-pass threshold on the squared decrement before the last step (contenders are polished)
        if (const char *e = getenv("THETA_N3_WARM_BLEND")) D.warm_blend = atof(e);
        D.force64 = 0;
        if (const char *e = getenv("THETA_N3_FORCE_F64")) D.force64 = atoi(e) != 0;
        if (const char *e = getenv("THETA_N3_CONV_L2")) D.conv_l2 = atof(e);
        D.no_dismiss = 0;
        if (const char *e = getenv("THETA_N3_NO_DISMISS")) D.no_dismiss = atoi(e) != 0;
        D.prefix_bound = 1;
        if (const char *e = getenv("THETA_N3_PREFIX_BOUND")) D.prefix_bound = atoi(e) != 0;
        D.no_second = 0;
        if (const char *e = getenv("THETA_N3_SECOND")) D.no_second = atoi(e) == 0;
        if (const char *e = getenv("THETA_N3_SIEVE")) p->opt_sieve = atoi(e) != 0;
        D.N = (double)N;
        D.Rtot = (double)Rt;
        D.K0 = (double)k0;
        D.r = (const double *)p->d_r.p;
        D.rN = (const double *)p->d_rN.p;
        D.lb = (const unsigned char *)p->d_small.p;
        D.ub = D.lb + m;
        D.ridx = D.lb + 2 * m;
        D.rowtab = D.ridx + h.ridx.size();
        D.smask = (const unsigned long long *)p->d_smask.p;
        D.dynmask = (const unsigned long long *)p->d_dynmask.p;
        D.swmask = h.swmask;
        size_t per_level = (size_t)h.Q * 2 * (h.NT + 1) * (h.NT + 1);
        size_t cnt_bytes = per_level * m * sizeof(u128);
        if (cnt_bytes > (size_t)ctx->hbm_bytes / 2) {
            theta_set_error("n=3 counting table needs %zu bytes (device has %llu)", cnt_bytes,
                            (unsigned long long)ctx->hbm_bytes);
            return THETA_ERR_HIP;
        }
        tc_say("uploads");
        TRY(p->d_misc.alloc(64));
        // A LARGE table (m = 200, K = 7: 2 GB, 200 launches of 2 ms) of a space that provably holds 2^128 matrices or more -- its count
        // is "2^128 - 1: that many or more" whatever the table says -- waits for the first call that takes ranks (ensure_table):
        // the mixture-space search, which serves such a space whole, never reads it.  (THETA_N3_LAZY_TABLE=0: always built here.)
        const bool lazy_ok = !(getenv("THETA_N3_LAZY_TABLE") && atoi(getenv("THETA_N3_LAZY_TABLE")) == 0);
        if (lazy_ok && cnt_bytes >= ((size_t)64 << 20) && n3_count_lower_bound_log2(h, m, tau) >= 130.0) {
            p->table_pending = true;
            p->count_saturated = true;
            p->total[0] = p->total[1] = ~0ull;
            D.cnt = nullptr;
            tc_say("counting DP deferred");
        } else {
            TRY(p->d_cnt.alloc(cnt_bytes));
            tc_say("table allocated");
            D.cnt = (const u128 *)p->d_cnt.p;
            HIP_TRY(hipMemsetAsync(p->d_misc.p, 0, 64, st));
            unsigned *ovf = (unsigned *)p->d_misc.p;
            unsigned long long *tot = (unsigned long long *)((char *)p->d_misc.p + 16);
            n3_run_dp(D, (u128 *)p->d_cnt.p, ovf, tot, st);
            unsigned char hostmisc[64];
            HIP_TRY(hipMemcpyAsync(hostmisc, p->d_misc.p, 64, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipGetLastError());
            tc_say("counting DP");
            unsigned hov;
            memcpy(&hov, hostmisc, 4);
            // (a space of 2^128 matrices or more: the counting table saturates -- n3_dp_kernel -- and theta_problem_count reports
            // 2^128 - 1, "at least that many"; the first 2^128 - 1 ranks of the reference's order are searched like any others)
            p->count_saturated = hov != 0;
            memcpy(p->total, hostmisc + 16, 16);
        }
        D.total_lo = p->total[0];
        D.total_hi = p->total[1];
        // leaf levels enumerated by the lanes (one byte of the 64-bit leaf code each, at most 8): enough of them that a
        // prefix has >= 32 k leaves on average (growth factor of the candidate count per interval ^ L), so that the per-prefix
        // work (group tile, 64 unranks) is amortised -- but no more, every leaf row is a likelihood term of its own.
        // m=50: K=6 -> 6, K<=5 -> 8 (measured on MI355X)
        int L = 8;
        {
            const double lg = log((double)p->total[1] * 18446744073709551616.0 + (double)p->total[0]) / (double)m;
            if (lg > 0.0 && 6.0 * lg >= log(32768.0)) L = 6;
        }
        if (const char *e = getenv("THETA_N3_LEAF_LEVELS")) {
            int v = atoi(e);
            if (v >= 1 && v <= 8) L = v;
        }
        if (L > m - 1) L = m - 1;
        D.L = L;
        TRY(p->d_tasks.alloc((size_t)N3_MAX_TASKS * sizeof(N3Task)));
        TRY(p->d_stbuf.alloc(((size_t)N3_MAX_TASKS * N3_STB + 16 * N3_STB) * sizeof(unsigned)));      // (+ the anchor path of n3_launch_tasks)
        tc_say("task buffers");
    }
    HIP_TRY(hipStreamSynchronize(st));
#undef TRY
    *out = owner.release();
    return THETA_OK;
}

/*
 * Run-time switches of a search instance (the THETA_N3_* environment variables only set their defaults at creation).
 */
extern "C" int theta_problem_set_option(theta_problem *p, const char *name, double value) {
    if (!p || !name) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    const std::string k(name);
    if (k == "n3_force_f64") p->n3.force64 = value != 0.0;
    else if (k == "n3_no_dismiss") p->n3.no_dismiss = value != 0.0;
    else if (k == "n3_prefix_bound") p->n3.prefix_bound = value != 0.0;
    else if (k == "n3_second") p->n3.no_second = value == 0.0;
    else if (k == "n3_conv_l2" && value > 0.0) p->n3.conv_l2 = value;
    else if (k == "n3_mu_tol" && value >= 0.0) p->n3.mu_tol = value;
    else if (k == "n3_warm_blend" && value >= 0.0 && value <= 1.0) p->n3.warm_blend = value;
    else if (k == "n3_sieve") p->opt_sieve = value != 0.0;
    else if (k == "n3_auto_f64") {
        p->opt_auto64 = value != 0.0;
        p->auto64 = 0;
    }
    else if (k == "n3_nan_sweep") p->opt_nan_sweep = value != 0.0;
    else if (k == "n3_contender_cap" && value >= 0.0 && value <= (double)SURV_CAP) p->opt_surv_cap = (unsigned)value;
    else if (k == "n3_per_task" && (value == 0.0 || (value >= 64 && value <= 65535))) p->opt_per_task = (uint64_t)value;
    else if (k == "mix_shard_world" && value >= 1.0 && value <= 4096.0) {
        p->opt_mix_G = (int)value;
        if (p->opt_mix_g >= p->opt_mix_G) p->opt_mix_g = 0;
    }
    else if (k == "mix_shard_rank" && value >= 0.0 && value < (double)p->opt_mix_G) p->opt_mix_g = (int)value;
    else if (k == "mix_max_boxes" && value >= 0.0) p->opt_mix_max_boxes = (uint64_t)value;
    else if (k == "mix_max_ms" && value >= 0.0) p->opt_mix_max_ms = value;
    else if (k == "mix_max_steps" && value >= 1.0) p->opt_mix_max_steps = (uint64_t)value;
    else if (k == "mix_dive_niches") p->opt_mix_niches = value != 0.0;
    else if (k == "mix_dive_blend" && value >= 0.0 && value <= 1.0) p->opt_mix_blend = value;
    else if (k == "mix_beam" && value >= 8.0 && value <= (double)MIX_BEAM_LIMIT) p->opt_mix_beam = (uint64_t)value;
    else if (k == "n2_no_dismiss") p->n2.quick = value == 0.0;
    else if (k == "n2_per_thread" && (value == 0.0 || (value >= 1 && value <= 512))) p->opt_per_thread = (int)value;
    else {
        theta_set_error("unknown option or value out of range: %s = %g", name, value);
        return THETA_ERR_ARG;
    }
    return THETA_OK;
}

extern "C" int theta_problem_count(theta_problem *p, uint64_t count[2]) {
    if (!p || !count) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    count[0] = p->total[0];
    count[1] = p->total[1];
    return THETA_OK;
}

static inline u128 mk128(const uint64_t v[2]) { return ((u128)v[1] << 64) | v[0]; }

static int check_range(theta_problem *p, const uint64_t rb[2], const uint64_t re[2], u128 &b, u128 &e) {
    if (!p || !rb || !re) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    if (p->mix_only) {
        theta_set_error("%d distinct rows (a, b) lie within the bounds: more than the 64 the rank-walking kernels hold; such a space is searched "
                        "whole (theta_mix_search, do_optimization_single), it has no ranks", p->n3.Q);
        return THETA_ERR_ARG;
    }
    if (p->table_pending) {
        int rc = ensure_table(p);
        if (rc) return rc;
    }
    u128 tot = mk128(p->total);
    if (tot == 0) {
        theta_set_error("no valid copy number profiles exist within the bounds");
        return THETA_ERR_NO_CANDIDATES;
    }
    b = mk128(rb);
    e = mk128(re);
    if (b > e || e > tot) {
        theta_set_error("rank range out of bounds");
        return THETA_ERR_ARG;
    }
    return THETA_OK;
}

// Runs the fused kernel over [b, e).  dump arrays are device pointers or null.
static int enumerate_device(theta_problem *p, u128 b, uint64_t count, unsigned char *d_out, double *kernel_ms);

// n=3 sieve path: the rank-deficient candidates (n3_core.hpp: N3Line) of the tasks that met a prefix with collinear rows.  The
// sieve kernel only reports such tasks (a per-candidate test in its hot loops cost 1-5 % of the kernel); here their rank
// ranges -- runs of consecutive tasks -- are materialised with the generator and scanned, one thread per candidate, and the
// ranks of the candidates whose rows all lie on one line join the degenerate list on the device.  Rare: none of the 2^31
// candidates of a bench step, 873 of the 21 050 matrices of the m=6, K=3 space.
typedef std::vector<std::pair<u128, uint64_t>> TaskList;      // (first rank, candidates) of every task of a multi-range search, in rank order
static int list_deficient(theta_problem *p, u128 b, u128 e, uint64_t per_task, unsigned line_count, SearchCounters &hc, const TaskList *tl = nullptr) {
    theta_ctx *ctx = p->ctx;
    hipStream_t st = ctx->stream;
    std::vector<unsigned long long> raw((size_t)2 * line_count);
    HIP_TRY(hipMemcpyAsync(raw.data(), p->d_line.p, raw.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (tl) {
        // several ranges: the flagged tasks by their first rank, each materialised and scanned by itself (runs of tasks that follow
        // each other without a gap together)
        std::vector<size_t> ix;
        for (unsigned i = 0; i < line_count; i++) {
            const u128 base = ((u128)raw[2 * i + 1] << 64) | raw[2 * i];
            auto it = std::lower_bound(tl->begin(), tl->end(), base, [](const std::pair<u128, uint64_t> &t, u128 v) { return t.first < v; });
            if (it != tl->end() && it->first == base) ix.push_back((size_t)(it - tl->begin()));
        }
        std::sort(ix.begin(), ix.end());
        ix.erase(std::unique(ix.begin(), ix.end()), ix.end());
        const size_t cb = (size_t)p->m * 2;
        for (size_t i = 0; i < ix.size();) {
            size_t j = i + 1;
            u128 rb = (*tl)[ix[i]].first, re = rb + (*tl)[ix[i]].second;
            while (j < ix.size() && ix[j] == ix[j - 1] + 1 && (*tl)[ix[j]].first == re && (uint64_t)(re - rb) * cb < ((uint64_t)256 << 20)) {
                re += (*tl)[ix[j]].second;
                j++;
            }
            const uint64_t count = (uint64_t)(re - rb);
            if (p->d_scan.bytes < count * cb) {
                p->d_scan.release();
                int rc = p->d_scan.alloc(count * cb);
                if (rc) return rc;
            }
            int rc = enumerate_device(p, rb, count, (unsigned char *)p->d_scan.p, nullptr);
            if (rc) return rc;
            n3_launch_collinear_scan((const unsigned char *)p->d_scan.p, count, p->m, rb, (SearchCounters *)p->d_ctr.p, (TieRecord *)p->d_deg.p,
                                     DEG_CAP, st);
            i = j;
        }
        HIP_TRY(hipMemcpyAsync(&hc.deg_count, (char *)p->d_ctr.p + offsetof(SearchCounters, deg_count), sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        return THETA_OK;
    }
    std::vector<uint64_t> tix(line_count);
    for (unsigned i = 0; i < line_count; i++) tix[i] = (uint64_t)(((((u128)raw[2 * i + 1]) << 64 | raw[2 * i]) - b) / per_task);
    std::sort(tix.begin(), tix.end());
    tix.erase(std::unique(tix.begin(), tix.end()), tix.end());
    const size_t cb = (size_t)p->m * 2;
    const uint64_t run_max = std::max<uint64_t>(1, ((uint64_t)256 << 20) / (per_task * cb));      // <= 256 MB of records at a time
    for (size_t i = 0; i < tix.size();) {
        size_t j = i + 1;
        while (j < tix.size() && tix[j] == tix[j - 1] + 1 && tix[j] - tix[i] < run_max) j++;
        const u128 rb = b + (u128)tix[i] * per_task;
        u128 re = b + (u128)(tix[j - 1] + 1) * per_task;
        if (re > e) re = e;
        const uint64_t count = (uint64_t)(re - rb);
        if (p->d_scan.bytes < count * cb) {
            p->d_scan.release();
            int rc = p->d_scan.alloc(count * cb);
            if (rc) return rc;
        }
        int rc = enumerate_device(p, rb, count, (unsigned char *)p->d_scan.p, nullptr);
        if (rc) return rc;
        n3_launch_collinear_scan((const unsigned char *)p->d_scan.p, count, p->m, rb, (SearchCounters *)p->d_ctr.p, (TieRecord *)p->d_deg.p,
                                 DEG_CAP, st);
        i = j;
    }
    HIP_TRY(hipMemcpyAsync(&hc.deg_count, (char *)p->d_ctr.p + offsetof(SearchCounters, deg_count), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return THETA_OK;
}

// n=3: the candidates of [b, e) the reference reports with a NaN likelihood.  For about one full-rank candidate in a million the
// reference's hybrj stops, unconverged, at a nu inside [0,1]^3 whose components do not sum to one; _solve_n3plus takes it
// (Optimizer.py:150-153), M3 makes a mu with a negative entry of it, L3 the logarithm of a negative number -- and the NaN tuple
// joins `best` wherever it stands (Misc.py:44-46).  Which candidates those are hangs on the trajectory of MINPACK's iteration;
// nothing short of running it tells (tools/nan_hunt.py: 89 of 1.4e8 candidates, none of them near its space's minimum).  So
// this sweep runs the reference's procedure, restated (solve_batch_n3_kernel), over EVERY candidate of the range -- generator ->
// solver -> scan, device resident, 2e8-5e8 candidates/s against the search's 3e10-9e10 -- and appends the ranks to the degenerate
// list.  Option "n3_nan_sweep"; theta_amd.search switches it on for spaces of up to 2^33 matrices (more than the reference could
// walk in a year) and reports when it did not.
static int nan_sweep(theta_problem *p, u128 b, u128 e, double near, SearchCounters &hc) {
    theta_ctx *ctx = p->ctx;
    hipStream_t st = ctx->stream;
    const size_t cb = (size_t)p->m * 2;
    const uint64_t chunk = (uint64_t)1 << 20;
    if (p->d_scan.bytes < chunk * cb) {
        p->d_scan.release();
        int rc = p->d_scan.alloc(chunk * cb);
        if (rc) return rc;
    }
    if (p->d_sweep.bytes < chunk * (1 + 3 * 8 + 8)) {
        int rc = p->d_sweep.alloc(chunk * (1 + 3 * 8 + 8));
        if (rc) return rc;
    }
    double *d_nll = (double *)p->d_sweep.p, *d_mu = d_nll + chunk;
    unsigned char *d_ok = (unsigned char *)(d_mu + 3 * chunk);
    for (u128 at = b; at < e; at += chunk) {
        const uint64_t c = (uint64_t)std::min<u128>((u128)chunk, e - at);
        int rc = enumerate_device(p, at, c, (unsigned char *)p->d_scan.p, nullptr);
        if (rc) return rc;
        batch_launch_solve(3, p->m, p->tau, (const double *)p->d_r.p, (const double *)p->d_rN.p, p->max_normal, (int)c,
                           (const unsigned char *)p->d_scan.p, d_ok, d_mu, d_nll, nullptr, st);
        n3_launch_nan_scan(d_ok, d_nll, c, at, near, (SearchCounters *)p->d_ctr.p, (TieRecord *)p->d_deg.p, DEG_CAP, st);
    }
    HIP_TRY(hipMemcpyAsync(&hc.deg_count, (char *)p->d_ctr.p + offsetof(SearchCounters, deg_count), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return THETA_OK;
}

// the per-wave statistics of the search kernels live in THETA_STAT_SLOTS copies of the counter block (SearchArgs::stat): add them up
static void fold_stats(const unsigned char *slots, SearchCounters &c) {
    for (int i = 0; i < THETA_STAT_SLOTS; i++) {
        SearchCounters s;
        memcpy(&s, slots + (size_t)i * THETA_STAT_STRIDE, sizeof(s));
        c.evaluated += s.evaluated;
        c.accepted += s.accepted;
        c.degenerate += s.degenerate;
        c.iterations += s.iterations;
        c.terms += s.terms;
        c.final_terms += s.final_terms;
        c.dismissed += s.dismissed;
        c.terms64 += s.terms64;
        c.finish_iterations += s.finish_iterations;
        c.sieve_pterms += s.sieve_pterms;
        c.sieve_children += s.sieve_children;
        c.sieve_pruned += s.sieve_pruned;
        for (int k = 0; k < 8; k++) c.prof[k] += s.prof[k];
    }
}

// (a witnessed call, theta_search_witness: device buffer of the records, sampling shift, capacity)
struct WitnessReq {
    SvWitness *d;
    unsigned shift;
    unsigned long long cap;
};

static int run_search(theta_problem *p, u128 b, u128 e, double window, double *dump_nll, double *dump_mu,
                      SearchCounters &hc, std::vector<TieRecord> &recs, double &kernel_ms, double &setup_ms,
                      unsigned long long &dropped_out, const WitnessReq *wit = nullptr,
                      const std::vector<std::pair<u128, u128>> *ranges = nullptr) {      // (n = 3 sieve path: several rank ranges [first, last) in rank order instead of [b, e))
    theta_ctx *ctx = p->ctx;
    hipStream_t st = ctx->stream;
    SearchArgs A;
    A.ctr = (SearchCounters *)p->d_ctr.p;
    A.stat = (SearchCounters *)p->d_stat.p;
    A.list = (TieRecord *)p->d_list.p;
    A.list_cap = LIST_CAP;
    A.sus = (TieRecord *)p->d_sus.p;
    A.sus_cap = SUS_CAP;
    A.deg = (TieRecord *)p->d_deg.p;
    A.deg_cap = DEG_CAP;
    A.line = (unsigned long long *)p->d_line.p;
    A.line_cap = p->d_line.p ? LINE_CAP : 0;
    A.window = window;
    A.dump_nll = dump_nll;
    A.dump_mu = dump_mu;
    A.wit = wit ? wit->d : nullptr;
    A.wit_begin_lo = (unsigned long long)b;
    A.wit_begin_hi = (unsigned long long)(b >> 64);
    A.wit_cap = wit ? wit->cap : 0;
    A.wit_shift = wit ? wit->shift : 0;
    memset(&hc, 0, sizeof(hc));
    hc.best_bits = order_bits(p->hint);   // a caller-supplied upper bound keeps tie and suspect lists short
    p->hint = INFINITY;
    hc.rej_bits = order_bits(INFINITY);
    kernel_ms = setup_ms = 0.0;
    unsigned long long list_dropped = 0;
    memset(&p->last_redo, 0, sizeof(p->last_redo));
    p->last_redo_ms = 0.0;
    p->last_sieve64 = false;
    p->last_fallback = 0;
    p->last_launches = 0;
    const unsigned surv_cap = p->opt_surv_cap ? p->opt_surv_cap : SURV_CAP;      // (the list is allocated for SURV_CAP)
    uint64_t sieve_per_task_last = 0;                                            // candidates per task of the last pass, if the sieve ran it
    TaskList host_tasks;                                                         // (several ranges: the tasks as the host cut them)
    std::vector<unsigned char> stat_host((size_t)THETA_STAT_SLOTS * THETA_STAT_STRIDE);
    for (int pass = 0; pass < 3; pass++) {
        std::vector<std::pair<int, int>> slices;     // n=3 fast path: (first task, tasks) of every sieve launch of this pass
        uint64_t sieve_per_task = 0;
        HIP_TRY(hipMemcpyAsync(p->d_ctr.p, &hc, sizeof(hc), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(p->d_stat.p, 0, (size_t)THETA_STAT_SLOTS * THETA_STAT_STRIDE, st));
        HIP_TRY(hipEventRecord(ctx->ev0, st));
        if (p->n == 2) {
            unsigned long long nb = (unsigned long long)b, ne = (unsigned long long)e, cnt = ne - nb;
            // candidates per thread: a thread unranks its first candidate (binary searches in the count table) and then
            // steps; 128 per thread amortise that (2.3x faster than 16 at m=50, k=6), fewer only for small ranges so
            // that there are still ~65536 threads
            unsigned long long per = cnt / 65536ull;
            if (per < 4) per = 4;
            if (per > 128) per = 128;
            if (const char *e = getenv("THETA_N2_PER_THREAD")) {
                long long v = atoll(e);
                if (v >= 1 && v <= 512) per = (unsigned long long)v;
            }
            if (p->opt_per_thread > 0) per = (unsigned long long)p->opt_per_thread;
            HIP_TRY(hipEventRecord(ctx->ev1, st));
            // a search that starts without a minimum (no hint, first pass): a sample of 4096 candidates spread over the range
            // gives it one first (n2.hip: sample launch), so that the threads of the search proper do not all begin "within the
            // window" of +inf
            if (hc.best_bits == order_bits(INFINITY) && !dump_nll && p->n2.quick && cnt >= 65536ull)
                n2_launch_search(p->n2, A, nb, ne, (int)per, st, cnt / 4096ull);
            n2_launch_search(p->n2, A, nb, ne, (int)per, st);
        } else {
            u128 cnt = e - b;
            if (ranges) {
                cnt = 0;
                for (const auto &rg : *ranges) cnt += rg.second - rg.first;
            }
            // candidates per wave task: 8192 for launches up to 2^29 candidates, 16384 beyond (the sieve's waves fetch tasks as
            // they finish them, so the tail of a launch is one task long: 16384 measured 3 % faster than 32768 on the search
            // leg, 8192 pays more for the task set-up than it wins); tuned, profiles/r4/NOTES.md
            uint64_t per_task = 8192;
            while (per_task < 16384 && (u128)per_task * 65536u < cnt) per_task <<= 1;
            if (const char *e = getenv("THETA_N3_PER_TASK")) {
                long long v = atoll(e);
                if (v >= 64 && v <= 65535) per_task = (uint64_t)v;   // the kernel keeps in-task offsets in 16 bits
            }
            if (p->opt_per_task > 0) per_task = p->opt_per_task;
            u128 nt = (cnt + per_task - 1) / per_task;
            if (ranges) {
                nt = 0;
                for (const auto &rg : *ranges) nt += (rg.second - rg.first + per_task - 1) / per_task;
            }
            if (nt > N3_MAX_TASKS) {
                theta_set_error("n=3 rank range too large for one call: at most %llu candidates (split the range; "
                                "theta_amd.Problem.search does)", (unsigned long long)N3_MAX_TASKS * per_task);
                return THETA_ERR_ARG;
            }
            int ntasks = (int)nt;
            N3Dev PS = p->n3;
            const int sieve_levels = (p->opt_sieve && !dump_nll) ? n3_sieve_levels(p->n3) : 0;   // (FP64 mode: the sieve's double instantiation)
            if (p->m > N3_MAX_M && sieve_levels == 0) {
                theta_set_error("n=3 with more than %d intervals runs on the sieve path only (no --GET_VALUES dump, n3_sieve = 1)", N3_MAX_M);
                return THETA_ERR_ARG;
            }
            if (ranges && sieve_levels == 0) {
                theta_set_error("a search over several rank ranges runs on the sieve path only (n = 3, m >= 8, n3_sieve = 1)");
                return THETA_ERR_ARG;
            }
            if (sieve_levels > 0) {
                // fast path (n3_sieve.hip): sieve kernel per slice of the range, finish kernel on its contenders in between
                PS.L = sieve_levels;
                if (p->opt_auto64 && p->auto64) PS.force64 = 1;      // (same finalists either way: only the screen's margin differs)
                if (!p->d_surv.p) {
                    int rc = p->d_surv.alloc((size_t)SURV_CAP * sizeof(SvSurvivor));
                    if (rc) return rc;
                    if ((rc = p->d_survcnt.alloc((SV_TASKCTR_OFF + SIEVE_MAX_SLICES) * sizeof(unsigned)))) return rc;   // (+ the launches' task counters)
                    if ((rc = p->d_survacc.alloc(SIEVE_MAX_SLICES * sizeof(unsigned)))) return rc;
                }
                HIP_TRY(hipMemsetAsync(p->d_survcnt.p, 0, (SV_TASKCTR_OFF + SIEVE_MAX_SLICES) * sizeof(unsigned), st));
                HIP_TRY(hipMemsetAsync(p->d_survacc.p, 0, SIEVE_MAX_SLICES * sizeof(unsigned), st));
                if (ranges) {
                    // the tasks of several ranges: cut on the host, unranked one by one (n3_task_list_kernel)
                    host_tasks.clear();
                    std::vector<uint64_t> spec;
                    spec.reserve((size_t)3 * ntasks);
                    for (const auto &rg : *ranges)
                        for (u128 at = rg.first; at < rg.second; at += per_task) {
                            const uint64_t c = (uint64_t)std::min<u128>((u128)per_task, rg.second - at);
                            host_tasks.push_back({at, c});
                            spec.push_back((uint64_t)at);
                            spec.push_back((uint64_t)(at >> 64));
                            spec.push_back(c);
                        }
                    if (p->d_misc2.bytes < spec.size() * sizeof(uint64_t)) {
                        int rc = p->d_misc2.alloc(spec.size() * sizeof(uint64_t));
                        if (rc) return rc;
                    }
                    HIP_TRY(hipMemcpyAsync(p->d_misc2.p, spec.data(), spec.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                    n3_launch_task_list(PS, (const uint64_t *)p->d_misc2.p, ntasks, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
                    HIP_TRY(hipStreamSynchronize(st));        // (`spec` leaves scope)
                } else {
                    n3_launch_tasks(PS, b, e, per_task, ntasks, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
                }
                HIP_TRY(hipEventRecord(ctx->ev1, st));
                int per_slice = (int)std::max<uint64_t>(1, SIEVE_SLICE / per_task);
                while ((ntasks + per_slice - 1) / per_slice > SIEVE_MAX_SLICES - 6) per_slice *= 2;   // (+ 4 bootstrap slices; the last counter is the redo's)
                // The sieve judges candidates against the running minimum, and only the finish kernel lowers it.  A call that
                // starts without one (no hint, first pass) works its way up through short slices first -- 1, 8, 64, 512 tasks --
                // so that the bulk of the range is sieved against a minimum that some candidate really attains.
                slices.clear();
                {
                    int t = 0;
                    if (hc.best_bits == order_bits(INFINITY))
                        for (int sz = 1; sz <= 512 && t < ntasks; sz *= 8) {
                            const int k = std::min(sz, ntasks - t);
                            slices.push_back({t, k});
                            t += k;
                        }
                    // A call that starts WITH a minimum may still run into a region whose candidates beat it (a stale minimum: a
                    // tenth of the bench's candidates looked like contenders in one stretch of twenty).  One short slice first --
                    // 2048 tasks, one round of the resident waves, 1/32 of a full call -- lets the finish kernel lower the
                    // minimum on a sample of the range before the bulk is judged against it.
                    // (round 5: and before that one, FOUR tasks -- 2^16 candidates of the range, a fraction of a millisecond --, so that even the
                    // short slice is judged against a minimum this range attains: with a stale hint it listed a tenth of its 3e7
                    // candidates as contenders, which is what made one step in twenty cost twice the median)
                    // (round 6: those four tasks as 64 SMALL ones -- the same 4 per_task candidates, behind the call's task list --: four waves
                    // of 16 384 candidates each held the chip for a millisecond, 1.4 % of a bench step, with 2 044 waves' slots idle; 64
                    // waves of 1 024 take 70 us.  Not under "n3_contender_cap" (the redo ladder re-cuts slices by whole tasks).)
                    if (t == 0 && ntasks >= 8 * 2048) {
                        const bool small_boot = !ranges && !p->opt_surv_cap && per_task % 16 == 0 && ntasks + 64 <= N3_MAX_TASKS &&
                                                (u128)4 * per_task <= cnt && !getenv("THETA_N3_NO_SMALL_BOOT");
                        if (small_boot) {
                            p->boot_spec.resize(3 * 64);
                            const uint64_t each = per_task / 16;
                            for (int i = 0; i < 64; i++) {
                                const u128 at = b + (u128)i * each;
                                p->boot_spec[3 * i] = (uint64_t)at;
                                p->boot_spec[3 * i + 1] = (uint64_t)(at >> 64);
                                p->boot_spec[3 * i + 2] = each;
                            }
                            if (p->d_misc2.bytes < p->boot_spec.size() * sizeof(uint64_t)) {
                                int rc = p->d_misc2.alloc(p->boot_spec.size() * sizeof(uint64_t));
                                if (rc) return rc;
                            }
                            HIP_TRY(hipMemcpyAsync(p->d_misc2.p, p->boot_spec.data(), p->boot_spec.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
                            n3_launch_task_list(PS, (const uint64_t *)p->d_misc2.p, 64, (N3Task *)p->d_tasks.p + ntasks,
                                                (unsigned *)p->d_stbuf.p + (size_t)ntasks * N3_STB, st);
                            slices.push_back({ntasks, 64});
                        } else {
                            slices.push_back({0, 4});
                        }
                        slices.push_back({4, 2044});
                        t = 2048;
                    }
                    while (t < ntasks) {
                        const int k = std::min(per_slice, ntasks - t);
                        slices.push_back({t, k});
                        t += k;
                    }
                }
                unsigned *cnts = (unsigned *)p->d_survcnt.p;
                for (size_t sl = 0; sl < slices.size(); sl++) {
                    const int t0 = slices[sl].first, nts = slices[sl].second;
                    n3_launch_sieve(PS, A, (const N3Task *)p->d_tasks.p + t0, (const unsigned *)p->d_stbuf.p + (size_t)t0 * N3_STB, nts,
                                    (SvSurvivor *)p->d_surv.p, surv_cap, cnts + sl, st);
                    n3_launch_finish(PS, A, (const SvSurvivor *)p->d_surv.p, surv_cap, cnts + sl, (unsigned *)p->d_survacc.p + sl, st);
                }
                sieve_per_task = per_task;
                sieve_per_task_last = per_task;
                p->last_sieve64 = PS.force64 != 0;
            } else {
                n3_launch_tasks(p->n3, b, e, per_task, ntasks, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
                HIP_TRY(hipEventRecord(ctx->ev1, st));
                n3_launch_search(p->n3, A, (const N3Task *)p->d_tasks.p, (const unsigned *)p->d_stbuf.p, ntasks, per_task, st);
            }
        }
        p->last_launches += slices.empty() ? 1 : (uint64_t)slices.size();
        HIP_TRY(hipEventRecord(ctx->ev2, st));
        SearchCounters got;
        unsigned hcnt[SIEVE_MAX_SLICES], hacc[SIEVE_MAX_SLICES];
        HIP_TRY(hipMemcpyAsync(&got, p->d_ctr.p, sizeof(got), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(stat_host.data(), p->d_stat.p, stat_host.size(), hipMemcpyDeviceToHost, st));
        if (!slices.empty()) {
            HIP_TRY(hipMemcpyAsync(hcnt, p->d_survcnt.p, slices.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(hacc, p->d_survacc.p, slices.size() * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        }
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        fold_stats(stat_host.data(), got);
        if (p->n == 2) got.terms64 = got.terms;
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
        kernel_ms += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        setup_ms += ms;
        const SearchCounters raw = got;          // (as the device wrote them)
        // the sieve does not count what its bound finishes: every regular candidate ends dismissed or listed as a contender
        auto sieve_dismissed = [&](const SearchCounters &k) -> unsigned long long {
            if (p->n3.no_dismiss) return 0ull;
            const unsigned long long gone = k.degenerate + k.sieve_survivors;
            return k.evaluated > gone ? k.evaluated - gone : 0ull;
        };
        if (!slices.empty()) got.dismissed = raw.dismissed + sieve_dismissed(raw);
        // The packed-FP32 screen carries a margin of 2e-5 sum r + 1 NLL units; where the candidates of a range differ by less than
        // that -- 200 intervals of which a range varies the last dozen: BASELINE config 5's shape -- it lists percent of them as
        // contenders, the list overflows and the ladder below redoes slices.  The double instantiation's margin is a few units: from
        // the next call on this problem runs it (6x faster there: 0.30 -> 0.05 s per 2^30 candidates).  Decided on the BULK slices
        // (their minimum is fresh: a stale one is the short first slice's to absorb), and only with the list at its full capacity.
        if (!slices.empty() && p->opt_auto64 && !p->auto64 && !p->n3.force64 && !p->n3.no_dismiss && !p->opt_surv_cap && slices.size() >= 2) {
            unsigned long long listed = 0, tasks_bulk = 0;
            for (size_t sl = 1; sl < slices.size(); sl++) {
                listed += hcnt[sl];
                tasks_bulk += (unsigned long long)slices[sl].second;
            }
            if (tasks_bulk >= 8 * 2048 && (double)listed > 0.005 * (double)tasks_bulk * (double)sieve_per_task) p->auto64 = 1;
        }
        // A slice of the sieve whose contender list overflowed is redone.  The usual cause is a STALE minimum: the slice runs into
        // a region whose candidates beat the running minimum it was judged against, so a large share of them looks like a
        // contender.  The finish kernel has meanwhile gone through the 2^24 that were listed and lowered the minimum, so the same
        // slice, sieved again, lists few; if it overflows again it is cut into 8 parts (each part's finish lowers the minimum
        // for the next); a part that still overflows -- a genuine stretch of 2^24 near-ties -- goes to the fused kernel, which
        // is complete by itself.  Records join the same device lists (theta_search drops duplicates by rank).
        uint64_t redone = 0, redone_accepted_by_finish = 0;
        double redo_ms = 0.0;
        for (size_t sl = 0; sl < slices.size(); sl++) {
            if (hcnt[sl] <= surv_cap) continue;
            if (ranges) {          // (the ladder below re-cuts a contiguous range; the caller searches these ranges one by one instead)
                theta_set_error("%u contenders in one slice of a search over several ranges exceed the list (%u): search the ranges one by one", hcnt[sl], surv_cap);
                return THETA_ERR_CAPACITY;
            }
            redone_accepted_by_finish += hacc[sl];
            const int t0 = slices[sl].first, nts = slices[sl].second;
            const u128 sb = b + (u128)t0 * sieve_per_task;
            u128 se = sb + (u128)nts * sieve_per_task;
            if (se > e) se = e;
            redone += (uint64_t)(se - sb);
            N3Dev PS = p->n3;
            PS.L = n3_sieve_levels(p->n3);
            if (p->opt_auto64 && p->auto64) PS.force64 = 1;
            unsigned *cnt2 = (unsigned *)p->d_survcnt.p + (SIEVE_MAX_SLICES - 1), *acc2 = (unsigned *)p->d_survacc.p + (SIEVE_MAX_SLICES - 1);
            // sieve + finish over tasks [ta, ta + na) of the rebuilt task list of this slice; returns the contender count
            auto again = [&](int ta, int na, unsigned &count) -> int {
                HIP_TRY(hipMemsetAsync(cnt2, 0, sizeof(unsigned), st));
                HIP_TRY(hipMemsetAsync(cnt2 + SV_TASKCTR_OFF, 0, sizeof(unsigned), st));
                HIP_TRY(hipEventRecord(ctx->ev1, st));
                n3_launch_sieve(PS, A, (const N3Task *)p->d_tasks.p + ta, (const unsigned *)p->d_stbuf.p + (size_t)ta * N3_STB, na,
                                (SvSurvivor *)p->d_surv.p, surv_cap, cnt2, st);
                n3_launch_finish(PS, A, (const SvSurvivor *)p->d_surv.p, surv_cap, cnt2, acc2, st);
                HIP_TRY(hipEventRecord(ctx->ev2, st));
                HIP_TRY(hipMemcpyAsync(&count, cnt2, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
                redo_ms += ms;
                return THETA_OK;
            };
            n3_launch_tasks(PS, sb, se, sieve_per_task, nts, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
            unsigned count = 0;
            int rc2 = again(0, nts, count);
            if (rc2) return rc2;
            if (count <= surv_cap) continue;
            const int part = (nts + 7) / 8;
            for (int ta = 0; ta < nts; ta += part) {
                const int na = std::min(part, nts - ta);
                if ((rc2 = again(ta, na, count))) return rc2;
                if (count <= surv_cap) continue;
                if (p->m > N3_MAX_M) {      // (the fused kernel holds one interval per lane)
                    theta_set_error("n=3, m = %d: %u contenders in one slice exceed the list (%u): pass a hint (theta_problem_hint) or "
                                    "search a shorter range", p->m, count, surv_cap);
                    return THETA_ERR_CAPACITY;
                }
                const u128 fb = sb + (u128)ta * sieve_per_task;
                u128 fe = fb + (u128)na * sieve_per_task;
                if (fe > se) fe = se;
                n3_launch_tasks(p->n3, fb, fe, sieve_per_task, na, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
                HIP_TRY(hipEventRecord(ctx->ev1, st));
                n3_launch_search(p->n3, A, (const N3Task *)p->d_tasks.p, (const unsigned *)p->d_stbuf.p, na, sieve_per_task, st);
                HIP_TRY(hipEventRecord(ctx->ev2, st));
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
                redo_ms += ms;
                // (the sieve's task list of this slice is rebuilt for the parts that follow)
                n3_launch_tasks(PS, sb, se, sieve_per_task, nts, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
            }
        }
        if (redone) {
            // The redone slices were counted once by the sieve; what the fused kernel did on them is reported APART
            // (stats->redo_*): the main counters stay those of the sieve + finish kernels, so that every candidate is counted
            // once in evaluated / dismissed / iterations / terms and the executed-FLOP figures of the two kernels do not mix.
            // `accepted`: the finish kernel's count for an overflowed slice covers the listed part only -- replaced by the
            // fused kernel's, which saw the whole slice.
            SearchCounters after;
            HIP_TRY(hipMemcpyAsync(&after, p->d_ctr.p, sizeof(after), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(stat_host.data(), p->d_stat.p, stat_host.size(), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            fold_stats(stat_host.data(), after);
            SearchCounters &R = p->last_redo;
            R.evaluated = after.evaluated - raw.evaluated;
            R.accepted = after.accepted - raw.accepted;
            R.degenerate = after.degenerate - raw.degenerate;
            R.iterations = after.iterations - raw.iterations;
            R.terms = after.terms - raw.terms;
            R.final_terms = after.final_terms - raw.final_terms;
            R.terms64 = after.terms64 - raw.terms64;
            R.sieve_pterms = after.sieve_pterms - raw.sieve_pterms;
            R.sieve_children = after.sieve_children - raw.sieve_children;
            R.sieve_pruned = after.sieve_pruned - raw.sieve_pruned;
            R.sieve_survivors = after.sieve_survivors - raw.sieve_survivors;
            R.finish_iterations = after.finish_iterations - raw.finish_iterations;
            R.dismissed = after.dismissed - raw.dismissed;
            const unsigned long long acc = got.accepted - std::min<unsigned long long>(got.accepted, redone_accepted_by_finish) + R.accepted;
            SearchCounters merged = after;                     // lists, minima and list counts: the latest
            merged.evaluated = got.evaluated;
            merged.degenerate = got.degenerate;
            merged.iterations = got.iterations;
            merged.terms = got.terms;
            merged.final_terms = got.final_terms;
            merged.dismissed = got.dismissed;
            merged.terms64 = got.terms64;
            merged.sieve_pterms = got.sieve_pterms;
            merged.sieve_children = got.sieve_children;
            merged.sieve_pruned = got.sieve_pruned;
            merged.sieve_survivors = got.sieve_survivors;
            merged.finish_iterations = got.finish_iterations;
            merged.accepted = acc;
            got = merged;
            p->last_redo_ms += redo_ms;
        }
        p->last_fallback += redone;
        if (got.list_count <= LIST_CAP || pass == 2) {
            unsigned long long dropped = got.list_count > LIST_CAP ? got.list_count - LIST_CAP : 0;
            hc = got;
            list_dropped = dropped;
            break;
        }
        // The list overflowed while the running minimum was still loose: go again with the minimum
        // found so far as the starting threshold (the kernels lower best_bits even when a record is dropped).
        SearchCounters keep;
        memset(&keep, 0, sizeof(keep));
        keep.best_bits = got.best_bits;
        keep.rej_bits = order_bits(INFINITY);
        hc = keep;
    }
    unsigned nrec = std::min<unsigned>(hc.list_count, LIST_CAP);
    recs.resize(nrec);
    if (nrec) HIP_TRY(hipMemcpyAsync(recs.data(), p->d_list.p, (size_t)nrec * sizeof(TieRecord), hipMemcpyDeviceToHost, st));
    unsigned nsus = std::min<unsigned>(hc.sus_count, SUS_CAP);
    p->suspects.resize(nsus);
    p->suspects_dropped = hc.sus_count > SUS_CAP ? hc.sus_count - SUS_CAP : 0;
    if (nsus) HIP_TRY(hipMemcpyAsync(p->suspects.data(), p->d_sus.p, (size_t)nsus * sizeof(TieRecord), hipMemcpyDeviceToHost, st));
    unsigned line_lost = 0;
    if (p->n == 3 && sieve_per_task_last > 0 && hc.line_count > 0) {
        if (hc.line_count > LINE_CAP) line_lost = hc.line_count - LINE_CAP;      // (reported like an overflow of the list itself)
        else {
            int rc = list_deficient(p, b, e, sieve_per_task_last, hc.line_count, hc, ranges ? &host_tasks : nullptr);
            if (rc) return rc;
        }
    }
    if (p->n == 3 && p->opt_nan_sweep && !dump_nll && e > b) {
        // (listed as well: whatever the procedure reports at or below the search's minimum + window -- the sweep then settles the
        // range's result by itself; without a finite minimum only the NaN outcomes are listed)
        const double best = order_unbits(hc.best_bits);
        int rc = THETA_OK;
        if (ranges) {
            for (const auto &rg : *ranges)
                if ((rc = nan_sweep(p, rg.first, rg.second, best < INFINITY ? best + window : -INFINITY, hc))) return rc;
        } else {
            rc = nan_sweep(p, b, e, best < INFINITY ? best + window : -INFINITY, hc);
        }
        if (rc) return rc;
    }
    unsigned ndeg = std::min<unsigned>(hc.deg_count, DEG_CAP);
    p->degenerate.resize(ndeg);
    p->degenerate_dropped = (hc.deg_count > DEG_CAP ? hc.deg_count - DEG_CAP : 0) + line_lost;
    if (ndeg) HIP_TRY(hipMemcpyAsync(p->degenerate.data(), p->d_deg.p, (size_t)ndeg * sizeof(TieRecord), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    dropped_out = list_dropped;
    return THETA_OK;
}

// FP64 operations per likelihood term (an FMA counts 2, rcp / log / div count 1); see DESIGN.md "Roofline".
static const double FLOPS_PER_TERM_ITER_N3 = 25.0;  // 2 sub, 2 fma (q), rcp + 2 fma, mul, 2 fma (grad), 3 mul, 3 fma (Hessian)
static const double FLOPS_PER_TERM_ITER_N2 = 13.0;  // fma (den), rcp + 2 fma, mul, fma (f), mul, fma (f')
static const double FLOPS_PER_TERM_ITER_N3_F32 = 24.0;  // packed pass: 2 sub, 2 fma (q), rcp, log + fma (value), mul, 2 fma (grad), 3 mul, 3 fma
static const double FLOPS_PER_FINAL_TERM_N3 = 9.0;  // f32 screen: 2 sub, 2 fma, log, fma
// sieve (n3_sieve.hip), shared sums of a last-level node, per term: 2 fma (q), rcp, log + fma (L), mul (t), add + 2 fma (T), mul (tw),
// add (W00), 2 mul + 2 add (W01, W02), 3 fma (W11, W12, W22)
static const double FLOPS_PER_TERM_SIEVE_SHARED = 26.0;
// ... and per child: its own term (26) + the restriction to its slice, the 2x2 solve, decrement, value and bound (6 + 6 fma, det 3,
// z.w 2 fma, rcp, 2 x 3 + 2 (d), 3 (l2), log + fma, step 5, bound 8)
static const double FLOPS_PER_SIEVE_CHILD = 90.0;
static const double FLOPS_PER_FINAL_TERM_N2 = 5.0;  // fma, log, fma

// the per-call statistics of theta_search / theta_search_witness from the device counters
static void fill_search_stats(theta_problem *p, const SearchCounters &hc, unsigned long long dropped, double kms, double sms,
                              theta_search_stats *stats) {
    const double best = order_unbits(hc.best_bits);
    stats->evaluated = hc.evaluated;
    stats->accepted = hc.accepted;
    stats->degenerate = hc.degenerate;
    stats->iterations = hc.iterations;
    stats->terms = hc.terms;
    stats->dismissed = hc.dismissed;
    stats->list_overflow = dropped;
    const double per = (p->n == 2) ? FLOPS_PER_TERM_ITER_N2 : FLOPS_PER_TERM_ITER_N3;
    const double fin = (p->n == 2) ? FLOPS_PER_FINAL_TERM_N2 : FLOPS_PER_FINAL_TERM_N3;
    auto count_flops = [&](const SearchCounters &k, uint64_t &f64, uint64_t &f32) {
        if (p->n == 2) {
            f64 = (uint64_t)(per * (double)k.terms + fin * (double)k.final_terms);
            f32 = 0;
        } else if (p->last_sieve64) {
            // n=3, the sieve in FP64 (n3_force_f64): every evaluation on doubles.  Per term of a full evaluation (sv_step, round 4's
            // form with rho = sqrt R): 2 sub, 2 fma (q), 2 fma (the Newton-Raphson step of the reciprocal), 2 fma (value and the
            // error bound of its logarithms), 3 mul (rho / q, alpha, beta), 2 fma (gradient), 3 fma (Hessian) = 27; per term of a
            // node's shared sums (sv_parent): 2 fma (q), 2 fma (reciprocal), 2 fma (L, LA), 3 mul, 3 fma (T), 6 fma (W) = 33.  The
            // single-precision operations per term are the seed of the reciprocal and the logarithm of the screened value
            // (v_rcp_f32, v_log_f32 of the same (float) q).  A child: its own term, the restriction to its slice, the 2x2 solve,
            // decrement, value and bound (FLOPS_PER_SIEVE_CHILD) + the Newton-Raphson steps of its three reciprocals and the
            // error-bound terms (+ 16), two logarithms and two reciprocal seeds in single precision.
            const double full = (double)(k.terms - k.terms64);
            // (round 6, the tight modes -- sv_third in n3_sieve.hip: the shared evaluations only shape starting points and are single
            // precision throughout.  A term of a node's sums, sv_parent_third: 2 fma (q), rcp, min, 3 mul + add + 2 fma (T), add, 2 mul,
            // 2 add, 3 fma (W), 6 mul, 6 add, 4 fma (third-order sums) = 45; a child, sv_child_eval_third: ~150 with its cubic
            // correction, and two FP64 fma for its column sums)
            const bool third = p->n3.no_dismiss && p->n3.conv_l2 < 1e-6 && !p->n3.no_second;
            f64 = (uint64_t)(per * ((double)k.terms64 + (double)k.finish_iterations * (double)p->m) +
                             27.0 * full + (third ? 0.0 : 33.0) * (double)k.sieve_pterms +
                             (third ? 4.0 : FLOPS_PER_SIEVE_CHILD + 16.0) * (double)k.sieve_children + fin * (double)k.final_terms);
            f32 = (uint64_t)((third ? 45.0 : 2.0) * (double)k.sieve_pterms + 2.0 * full + (third ? 150.0 : 4.0) * (double)k.sieve_children);
        } else {   // n=3: FP64 iterations (dump, ill-conditioned candidates, the finish kernel: m terms each) + the packed-f32
                   // coarse pass and screen
            f64 = (uint64_t)(per * ((double)k.terms64 + (double)k.finish_iterations * (double)p->m));
            f32 = (uint64_t)(FLOPS_PER_TERM_ITER_N3_F32 * (double)(k.terms - k.terms64) + fin * (double)k.final_terms +
                             FLOPS_PER_TERM_SIEVE_SHARED * (double)k.sieve_pterms + FLOPS_PER_SIEVE_CHILD * (double)k.sieve_children);
        }
    };
    count_flops(hc, stats->flops, stats->flops_f32);
    count_flops(p->last_redo, stats->redo_flops, stats->redo_flops_f32);    // slices redone (contender list full): apart
    stats->redo_kernel_ms = p->last_redo_ms;
    stats->kernel_launches = p->last_launches;
    stats->pruned = hc.sieve_pruned;
    stats->survivors = hc.sieve_survivors;
    stats->fallback_candidates = p->last_fallback;
    stats->best_nll = best;
    stats->rejected_bound = order_unbits(hc.rej_bits);
    stats->rejected_rank[0] = hc.rej_rank_lo;
    stats->rejected_rank[1] = hc.rej_rank_hi;
    stats->kernel_ms = kms;
    stats->setup_ms = sms;
    for (int i = 0; i < 8; i++) stats->phase_cycles[i] = hc.prof[i];
}

static int search_impl(theta_problem *p, u128 b, u128 e, const std::vector<std::pair<u128, u128>> *ranges, double window, int cap, double *nll,
                       double *mu, uint64_t *rank, uint8_t *C, int *n_out, theta_search_stats *stats);

extern "C" int theta_search(theta_problem *p, const uint64_t rank_begin[2], const uint64_t rank_end[2], double window,
                            int cap, double *nll, double *mu, uint64_t *rank, uint8_t *C, int *n_out,
                            theta_search_stats *stats) {
    u128 b, e;
    int rc = check_range(p, rank_begin, rank_end, b, e);
    if (rc) return rc;
    return search_impl(p, b, e, nullptr, window, cap, nll, mu, rank, C, n_out, stats);
}

// theta_search over several rank ranges in ONE pass of the kernels (the survivors of theta_bnb: thousands of short ranges, each of
// which would cost a call's fixed overhead -- task set-up, a dozen synchronisations, the side lists -- by itself).
extern "C" int theta_search_ranges(theta_problem *p, int nranges, const uint64_t *ranges, double window, int cap, double *nll, double *mu,
                                   uint64_t *rank, uint8_t *C, int *n_out, theta_search_stats *stats) {
    if (!p || nranges < 0 || (nranges > 0 && !ranges)) {
        theta_set_error("theta_search_ranges: bad argument");
        return THETA_ERR_ARG;
    }
    if (p->n != 3 || p->mix_only) {
        theta_set_error("theta_search_ranges: n = 3 with at most 64 rows within the bounds only (a space with more has no ranks)");
        return THETA_ERR_ARG;
    }
    if (p->table_pending) {
        int rc = ensure_table(p);
        if (rc) return rc;
    }
    std::vector<std::pair<u128, u128>> rg;
    const u128 total = ((u128)p->total[1] << 64) | p->total[0];
    u128 prev = 0, sum = 0;
    for (int i = 0; i < nranges; i++) {
        const u128 rb = ((u128)ranges[4 * i + 1] << 64) | ranges[4 * i], rc_ = ((u128)ranges[4 * i + 3] << 64) | ranges[4 * i + 2];
        if (rb < prev || rb + rc_ < rb || rb + rc_ > total) {
            theta_set_error("theta_search_ranges: range %d is out of order, overlaps its predecessor or leaves the space", i);
            return THETA_ERR_ARG;
        }
        if (rc_ == 0) continue;
        rg.push_back({rb, rb + rc_});
        prev = rb + rc_;
        sum += rc_;
    }
    if (sum > ((u128)1 << 31)) {
        theta_set_error("theta_search_ranges: at most 2^31 candidates per call (theta_amd.Problem.search_ranges batches)");
        return THETA_ERR_ARG;
    }
    const u128 b = rg.empty() ? 0 : rg.front().first, e = rg.empty() ? 0 : rg.back().second;
    return search_impl(p, b, e, &rg, window, cap, nll, mu, rank, C, n_out, stats);
}

static int search_impl(theta_problem *p, u128 b, u128 e, const std::vector<std::pair<u128, u128>> *ranges, double window, int cap, double *nll,
                       double *mu, uint64_t *rank, uint8_t *C, int *n_out, theta_search_stats *stats) {
    int rc;
    if (!n_out || cap < 0 || (cap > 0 && (!nll || !mu || !rank || !C))) {
        theta_set_error("theta_search: null output");
        return THETA_ERR_ARG;
    }
    if (!(window >= 0.0)) {
        theta_set_error("window must be >= 0");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(p->ctx->device);
    *n_out = 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (b == e) {            // an empty range: nothing found -- and nothing left over from the previous call either
        p->suspects.clear();
        p->degenerate.clear();
        p->suspects_dropped = p->degenerate_dropped = 0;
        p->last_fallback = 0;
        p->hint = INFINITY;
        return THETA_OK;
    }
    SearchCounters hc;
    std::vector<TieRecord> recs;
    double kms, sms;
    unsigned long long dropped = 0;
    rc = run_search(p, b, e, window, nullptr, nullptr, hc, recs, kms, sms, dropped, nullptr, ranges);
    if (rc) return rc;
    double best = order_unbits(hc.best_bits);
    if (stats) fill_search_stats(p, hc, dropped, kms, sms, stats);
    // keep what lies within the window of the final minimum, in rank order
    std::vector<TieRecord> keep;
    for (const TieRecord &t : recs)
        if (t.nll <= best + window) keep.push_back(t);
    auto by_rank = [](const TieRecord &x, const TieRecord &y) {
        return x.rank_hi != y.rank_hi ? x.rank_hi < y.rank_hi : x.rank_lo < y.rank_lo;
    };
    {   // suspects: keep those whose bound is within the window of the FINAL minimum
        std::vector<TieRecord> sk;
        for (const TieRecord &t : p->suspects)
            if (t.nll <= best + window) sk.push_back(t);
        std::sort(sk.begin(), sk.end(), by_rank);
        sk.erase(std::unique(sk.begin(), sk.end(), [](const TieRecord &x, const TieRecord &y) {
                     return x.rank_hi == y.rank_hi && x.rank_lo == y.rank_lo; }), sk.end());
        p->suspects.swap(sk);
    }
    auto same_rank = [](const TieRecord &x, const TieRecord &y) { return x.rank_hi == y.rank_hi && x.rank_lo == y.rank_lo; };
    std::sort(keep.begin(), keep.end(), by_rank);
    keep.erase(std::unique(keep.begin(), keep.end(), same_rank), keep.end());     // (a slice redone by the fused kernel lists twice)
    std::sort(p->degenerate.begin(), p->degenerate.end(), by_rank);
    p->degenerate.erase(std::unique(p->degenerate.begin(), p->degenerate.end(), same_rank), p->degenerate.end());
    *n_out = (int)keep.size();
    if ((int)keep.size() > cap) {
        theta_set_error("%zu candidates within the window but capacity is %d", keep.size(), cap);
        return THETA_ERR_CAPACITY;
    }
    if (keep.empty()) return THETA_OK;
    // materialise their matrices on the device
    hipStream_t st = p->ctx->stream;
    size_t cb = (size_t)p->m * (p->n - 1);
    DevBuf d_rec, d_C;
    rc = d_rec.alloc(keep.size() * sizeof(TieRecord));
    if (rc) return rc;
    rc = d_C.alloc(keep.size() * cb);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(d_rec.p, keep.data(), keep.size() * sizeof(TieRecord), hipMemcpyHostToDevice, st));
    if (p->n == 2) n2_launch_unrank_list(p->n2, (const TieRecord *)d_rec.p, (int)keep.size(), (unsigned char *)d_C.p, st);
    else n3_launch_unrank_list(p->n3, (const TieRecord *)d_rec.p, (int)keep.size(), (unsigned char *)d_C.p, st);
    HIP_TRY(hipMemcpyAsync(C, d_C.p, keep.size() * cb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (size_t i = 0; i < keep.size(); i++) {
        nll[i] = keep[i].nll;
        for (int j = 0; j < p->n; j++) mu[i * p->n + j] = keep[i].mu[j];
        rank[2 * i] = keep[i].rank_lo;
        rank[2 * i + 1] = keep[i].rank_hi;
    }
    return THETA_OK;
}

// Per-candidate dump of the fused kernel (the reference's --GET_VALUES, RunTHetA.py:210-215):
// nll[count] (NaN = None), mu[count*n].  Diagnostic / parity entry point.
extern "C" int theta_search_values(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, double *nll,
                                   double *mu, theta_search_stats *stats) {
    uint64_t re[2];
    u128 b0 = rank_begin ? mk128(rank_begin) : 0;
    u128 e0 = b0 + count;
    re[0] = (uint64_t)e0;
    re[1] = (uint64_t)(e0 >> 64);
    u128 b, e;
    int rc = check_range(p, rank_begin, re, b, e);
    if (rc) return rc;
    if (!nll || !mu) {
        theta_set_error("null output");
        return THETA_ERR_ARG;
    }
    if (count == 0) return THETA_OK;
    HIP_ENTER(p->ctx->device);
    hipStream_t st = p->ctx->stream;
    if (p->n == 3 && p->m > N3_MAX_M) {
        // More intervals than the fused kernel holds (one per lane): the dump is what the REFERENCE reports for each candidate --
        // the generator's matrices through theta_solve_batch's kernel (hybrj, the BFGS decision, M3, L3: RunTHetA.py:185-215 per
        // candidate), chunk by chunk without leaving HBM.  NaN where the reference reports nothing (Optimizer.solve -> None).
        const uint64_t chunk = std::min<uint64_t>(count, 1ull << 20);
        const size_t cb = (size_t)p->m * 2;
        DevBuf d_C, d_ok, d_nll, d_mu;
        if ((rc = d_C.alloc(chunk * cb)) || (rc = d_ok.alloc(chunk)) || (rc = d_nll.alloc(chunk * sizeof(double))) ||
            (rc = d_mu.alloc(chunk * 3 * sizeof(double))))
            return rc;
        std::vector<unsigned char> ok(chunk);
        double kms = 0.0;
        uint64_t accepted = 0;
        for (uint64_t at = 0; at < count; at += chunk) {
            const uint64_t c = std::min<uint64_t>(chunk, count - at);
            double ems = 0.0;
            rc = enumerate_device(p, b + at, c, (unsigned char *)d_C.p, &ems);
            if (rc) return rc;
            HIP_TRY(hipEventRecord(p->ctx->ev0, st));
            batch_launch_solve(3, p->m, p->tau, (const double *)p->d_r.p, (const double *)p->d_rN.p, p->max_normal, (int)c,
                               (const unsigned char *)d_C.p, (unsigned char *)d_ok.p, (double *)d_mu.p, (double *)d_nll.p, nullptr, st);
            HIP_TRY(hipEventRecord(p->ctx->ev1, st));
            HIP_TRY(hipMemcpyAsync(nll + at, d_nll.p, c * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(mu + 3 * at, d_mu.p, c * 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipMemcpyAsync(ok.data(), d_ok.p, c, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipGetLastError());
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, p->ctx->ev0, p->ctx->ev1));
            kms += ems + (double)ms;
            const double qnan = __builtin_nan("");
            for (uint64_t i = 0; i < c; i++) {
                if (ok[i]) {
                    accepted++;
                } else {
                    nll[at + i] = qnan;
                    mu[3 * (at + i)] = mu[3 * (at + i) + 1] = mu[3 * (at + i) + 2] = qnan;
                }
            }
        }
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->evaluated = count;
            stats->accepted = accepted;
            stats->kernel_ms = kms;
            stats->best_nll = __builtin_inf();
            for (uint64_t i = 0; i < count; i++)
                if (nll[i] < stats->best_nll) stats->best_nll = nll[i];
        }
        return THETA_OK;
    }
    DevBuf d_nll, d_mu;
    rc = d_nll.alloc(count * sizeof(double));
    if (rc) return rc;
    rc = d_mu.alloc(count * p->n * sizeof(double));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d_nll.p, 0xff, count * sizeof(double), st));
    HIP_TRY(hipMemsetAsync(d_mu.p, 0xff, count * p->n * sizeof(double), st));
    SearchCounters hc;
    std::vector<TieRecord> recs;
    double kms, sms;
    unsigned long long dropped = 0;
    rc = run_search(p, b, e, 0.0, (double *)d_nll.p, (double *)d_mu.p, hc, recs, kms, sms, dropped);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(nll, d_nll.p, count * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(mu, d_mu.p, count * p->n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->evaluated = hc.evaluated;
        stats->accepted = hc.accepted;
        stats->degenerate = hc.degenerate;
        stats->iterations = hc.iterations;
        stats->terms = hc.terms;
        stats->best_nll = order_unbits(hc.best_bits);
        stats->rejected_bound = order_unbits(hc.rej_bits);
        stats->kernel_ms = kms;
        stats->setup_ms = sms;
    }
    return THETA_OK;
}

// What the n=3 sieve kernel LEAVES a candidate at (no reference counterpart; the evidence behind the bench's unit of work, "a
// candidate passed through the full solve": RunTHetA.py:191-208 per candidate, Optimizer.py:128-165).
extern "C" int theta_search_witness(theta_problem *p, const uint64_t rank_begin[2], const uint64_t rank_end[2], double window,
                                    int every_log2, uint64_t cap, theta_witness *out, uint64_t *n_out, theta_search_stats *stats) {
    static_assert(sizeof(theta_witness) == sizeof(SvWitness), "theta_witness is the kernel's record");
    u128 b, e;
    int rc = check_range(p, rank_begin, rank_end, b, e);
    if (rc) return rc;
    if (!n_out || (cap > 0 && !out) || every_log2 < 0 || every_log2 > 40 || !(window >= 0.0)) {
        theta_set_error("theta_search_witness: bad argument");
        return THETA_ERR_ARG;
    }
    if (p->n != 3 || !p->opt_sieve || n3_sieve_levels(p->n3) == 0) {
        theta_set_error("theta_search_witness: the n=3 sieve path only (n = 3, m >= 8, n3_sieve = 1)");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(p->ctx->device);
    if (stats) memset(stats, 0, sizeof(*stats));
    const u128 span = e - b;
    const uint64_t need = (uint64_t)((span + (((u128)1 << every_log2) - 1)) >> every_log2);
    *n_out = need;
    if (need > cap) {
        theta_set_error("theta_search_witness: %llu records but capacity is %llu", (unsigned long long)need, (unsigned long long)cap);
        return THETA_ERR_CAPACITY;
    }
    if (need == 0) return THETA_OK;
    hipStream_t st = p->ctx->stream;
    DevBuf d_w;
    if ((rc = d_w.alloc((size_t)need * sizeof(SvWitness)))) return rc;
    HIP_TRY(hipMemsetAsync(d_w.p, 0, (size_t)need * sizeof(SvWitness), st));
    WitnessReq wr{(SvWitness *)d_w.p, (unsigned)every_log2, need};
    SearchCounters hc;
    std::vector<TieRecord> recs;
    double kms, sms;
    unsigned long long dropped = 0;
    rc = run_search(p, b, e, window, nullptr, nullptr, hc, recs, kms, sms, dropped, &wr);
    if (rc) return rc;
    if (stats) fill_search_stats(p, hc, dropped, kms, sms, stats);
    HIP_TRY(hipMemcpyAsync(out, d_w.p, (size_t)need * sizeof(SvWitness), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return THETA_OK;
}

// ---- branch and bound above the prefix (bnb.hip) -----------------------------------------------------------------------------
// The walk: levels 0 .. emit_depth - 1 hold pending frontier nodes; the deepest non-empty level is expanded first, a chunk of at
// most `chunk` nodes at a time (breadth first while a level fits its buffer, depth first by chunks when it does not), so every
// level's buffer holds at most chunk x Q children.  Beam mode (beam > 0) is breadth first only and keeps the `beam` smallest
// bounds of every level.
extern "C" int theta_bnb(theta_problem *p, double threshold, uint64_t beam, int follow_collinear, uint64_t max_nodes, uint64_t cap,
                         uint64_t *ranges, uint64_t *n_out, theta_bnb_stats *stats) {
    if (!p || !n_out || (cap > 0 && !ranges)) {
        theta_set_error("theta_bnb: null argument");
        return THETA_ERR_ARG;
    }
    *n_out = 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (p->n != 3) {
        theta_set_error("theta_bnb: n = 3 only (an n = 2 space is exhausted by theta_search)");
        return THETA_ERR_ARG;
    }
    if (p->mix_only) {
        theta_set_error("theta_bnb: %d distinct rows (a, b) lie within the bounds, more than the 64 a child mask holds: the row tree is not walked, "
                        "the space has no ranks (theta_mix_search searches it whole)", p->n3.Q);
        return THETA_ERR_ARG;
    }
    if (p->count_saturated) {
        theta_set_error("theta_bnb: the space holds 2^128 matrices or more; its nodes have no 128-bit ranks");
        return THETA_ERR_OVERFLOW;
    }
    if (threshold != threshold) {
        theta_set_error("theta_bnb: threshold is NaN");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(p->ctx->device);
    hipStream_t st = p->ctx->stream;
    const int m = p->m, Q = p->n3.Q;
    // emit depth: the sieve's prefix depth (the search of a range starts with whole prefixes); small problems go all the way down
    const int sieve_levels = p->opt_sieve ? n3_sieve_levels(p->n3) : 0;
    const int emit_depth = std::max(1, m - (sieve_levels > 0 ? sieve_levels : std::min(m - 1, 4)));
    const uint64_t emit_max = 1ull << 12;            // a node with this few matrices left is cheaper to search than to expand
    const int stride = (m + 3) & ~3;
    // per-depth constants (host, long double): depth dc = rows fixed
    std::vector<double> constc(m + 1), Z0(m + 1);
    {
        long double N = 0, Rt = 0;
        for (int i = 0; i < m; i++) {
            N += p->h_rN[i];
            Rt += p->h_r[i];
        }
        std::vector<long double> tail(m + 1, 0.0L);   // sum over l >= dc of r_l ln(r_l / (Rtot N_l))
        for (int l = m - 1; l >= 0; l--) {
            tail[l] = tail[l + 1];
            if (p->h_r[l] > 0) tail[l] += (long double)p->h_r[l] * logl((long double)p->h_r[l] / (Rt * ((long double)p->h_rN[l] / N)));
        }
        long double Rp = 0, om = 0;
        for (int dc = 1; dc <= m; dc++) {
            Rp += p->h_r[dc - 1];
            om += (long double)p->h_rN[dc - 1] / N;
            Z0[dc] = (double)om;
            long double c = -tail[dc];
            if (Rp > 0) c += Rp * logl(om) + Rp * logl(Rt / Rp);
            constc[dc] = (double)c;
        }
    }
    // buffers
    const size_t node_bytes = sizeof(BnbNode) + (size_t)stride;
    uint64_t chunk = 1ull << 16;
    {
        const double budget = 24e9;                    // bytes of level buffers (of the 288 GB)
        while (chunk > 256 && (double)chunk * Q * node_bytes * (double)(emit_depth + 1) > budget) chunk >>= 1;
    }
    if (beam > chunk) beam = chunk;
    const uint64_t level_cap = chunk * (uint64_t)Q;
    const uint64_t range_cap = std::max<uint64_t>(cap, 1ull << 16);
    struct Level {
        DevBuf nodes, paths;
        uint64_t count = 0, head = 0;
    };
    std::vector<std::unique_ptr<Level>> levels(emit_depth + 1);
    auto level = [&](int d) -> Level * {
        if (!levels[d]) {
            levels[d].reset(new Level());
            const uint64_t c = d == 0 ? 1 : level_cap;
            if (levels[d]->nodes.alloc((size_t)c * sizeof(BnbNode)) || levels[d]->paths.alloc((size_t)c * stride)) return nullptr;
        }
        return levels[d].get();
    };
    DevBuf d_ranges, d_ctr, d_stats, d_bounds, d_scr_nodes, d_scr_paths;
    int rc;
    if ((rc = d_ranges.alloc((size_t)range_cap * sizeof(BnbRange)))) return rc;
    if ((rc = d_ctr.alloc(4 * sizeof(unsigned long long)))) return rc;
    if ((rc = d_stats.alloc((size_t)BNB_STAT_SLOTS * BNB_STAT_STRIDE * sizeof(unsigned long long)))) return rc;
    HIP_TRY(hipMemsetAsync(d_ctr.p, 0, 4 * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(d_stats.p, 0, (size_t)BNB_STAT_SLOTS * BNB_STAT_STRIDE * sizeof(unsigned long long), st));
    if (beam) {
        if ((rc = d_bounds.alloc((size_t)level_cap * sizeof(double)))) return rc;
        if ((rc = d_scr_nodes.alloc((size_t)level_cap * sizeof(BnbNode)))) return rc;
        if ((rc = d_scr_paths.alloc((size_t)level_cap * stride))) return rc;
    }
    // the root: no rows, rank 0
    {
        Level *L0 = level(0);
        if (!L0) return THETA_ERR_HIP;
        BnbNode root;
        memset(&root, 0, sizeof(root));
        root.w0 = NAN;
        root.bound = -INFINITY;
        HIP_TRY(hipMemcpyAsync(L0->nodes.p, &root, sizeof(root), hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(L0->paths.p, 0, stride, st));
        L0->count = 1;
        L0->head = 0;
    }
    const auto t_start = std::chrono::steady_clock::now();
    uint64_t expanded = 0, launches = 0, ranges_written = 0, max_frontier = 1;
    std::vector<uint64_t> frontier(m + 1, 0);
    bool budget_hit = false;
    std::vector<double> hb;
    HIP_TRY(hipEventRecord(p->ctx->ev0, st));
    for (;;) {
        int d = -1;
        for (int k = emit_depth - 1; k >= 0; k--)
            if (levels[k] && levels[k]->head < levels[k]->count) {
                d = k;
                break;
            }
        if (d < 0) break;
        Level *L = levels[d].get();
        Level *Ln = level(d + 1 <= emit_depth ? d + 1 : emit_depth);
        if (!Ln) return THETA_ERR_HIP;
        if (Ln->head >= Ln->count) Ln->head = Ln->count = 0;       // (deepest first: the next level is empty whenever this one is expanded)
        const uint64_t take = std::min<uint64_t>(chunk, L->count - L->head);
        if (max_nodes && expanded + take > max_nodes) {
            budget_hit = true;
            break;
        }
        BnbArgs A;
        A.d = d;
        A.emit_depth = emit_depth;
        A.emit_max = beam ? 0 : emit_max;            // (a dive's ranges are its beam at the emit depth: small subtrees compete like the others)
        A.thr = beam ? INFINITY : threshold;
        A.full_bound = beam ? 1 : 0;
        A.follow_line = follow_collinear ? 1 : 0;
        A.constc = constc[d + 1];
        A.Z0 = Z0[d + 1];
        A.rd = p->h_r[d];
        A.nd = p->h_rN[d] / p->n3.N;
        A.path_stride = stride;
        A.in = (const BnbNode *)L->nodes.p + L->head;
        A.in_path = (const unsigned char *)L->paths.p + (size_t)L->head * stride;
        A.n_in = (unsigned)take;
        A.out = (BnbNode *)Ln->nodes.p + Ln->count;
        A.out_path = (unsigned char *)Ln->paths.p + (size_t)Ln->count * stride;
        A.out_cap = level_cap - Ln->count;
        A.ranges = (BnbRange *)d_ranges.p + ranges_written;
        A.range_cap = range_cap - ranges_written;
        A.counters = (unsigned long long *)d_ctr.p;
        A.stats = (unsigned long long *)d_stats.p;
        HIP_TRY(hipMemsetAsync(d_ctr.p, 0, 2 * sizeof(unsigned long long), st));
        bnb_launch_expand(p->n3, A, st);
        unsigned long long got[2];
        HIP_TRY(hipMemcpyAsync(got, d_ctr.p, sizeof(got), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        launches++;
        expanded += take;
        L->head += take;
        if (getenv("THETA_BNB_DEBUG"))
            fprintf(stderr, "bnb: depth %d, %llu nodes -> %llu children (level holds %llu), %llu ranges\n", d, (unsigned long long)take, got[0],
                    (unsigned long long)Ln->count, got[1]);
        if (got[0] > A.out_cap) {
            theta_set_error("theta_bnb: internal: a level buffer overflowed (%llu children, room for %llu)", got[0], (unsigned long long)A.out_cap);
            return THETA_ERR_HIP;
        }
        if (got[1] > A.range_cap) {
            theta_set_error("theta_bnb: more than %llu surviving rank ranges: the threshold is too far above the minimum (or pass a larger cap)",
                            (unsigned long long)range_cap);
            *n_out = ranges_written + got[1];
            return THETA_ERR_CAPACITY;
        }
        Ln->count += got[0];
        ranges_written += got[1];
        frontier[d + 1] += got[0];
        if (beam && Ln->count > beam) {
            // keep the `beam` smallest bounds of the level (breadth first: the level is complete once its parents are all expanded)
            if (L->head >= L->count) {
                const uint64_t n = Ln->count;
                hb.resize(n);
                bnb_launch_bounds((const BnbNode *)Ln->nodes.p, n, (double *)d_bounds.p, st);
                HIP_TRY(hipMemcpyAsync(hb.data(), d_bounds.p, n * sizeof(double), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                std::vector<double> srt(hb);
                std::nth_element(srt.begin(), srt.begin() + (beam - 1), srt.end());
                const double cut = srt[beam - 1];
                HIP_TRY(hipMemsetAsync((unsigned long long *)d_ctr.p + 2, 0, sizeof(unsigned long long), st));
                bnb_launch_compact((const BnbNode *)Ln->nodes.p, (const unsigned char *)Ln->paths.p, n, stride, cut, (BnbNode *)d_scr_nodes.p,
                                   (unsigned char *)d_scr_paths.p, level_cap, (unsigned long long *)d_ctr.p + 2, st);
                unsigned long long kept = 0;
                HIP_TRY(hipMemcpyAsync(&kept, (unsigned long long *)d_ctr.p + 2, sizeof(kept), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                kept = std::min<unsigned long long>(kept, level_cap);
                HIP_TRY(hipMemcpyAsync(Ln->nodes.p, d_scr_nodes.p, (size_t)kept * sizeof(BnbNode), hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync(Ln->paths.p, d_scr_paths.p, (size_t)kept * stride, hipMemcpyDeviceToDevice, st));
                Ln->count = kept;
            }
        }
        max_frontier = std::max<uint64_t>(max_frontier, Ln->count);
    }
    HIP_TRY(hipEventRecord(p->ctx->ev1, st));
    // the ranges, in rank order, adjacent ones joined
    std::vector<BnbRange> hr(ranges_written);
    if (ranges_written) HIP_TRY(hipMemcpyAsync(hr.data(), d_ranges.p, (size_t)ranges_written * sizeof(BnbRange), hipMemcpyDeviceToHost, st));
    std::vector<unsigned long long> hs((size_t)BNB_STAT_SLOTS * BNB_STAT_STRIDE);
    HIP_TRY(hipMemcpyAsync(hs.data(), d_stats.p, hs.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p->ctx->ev0, p->ctx->ev1));
    std::sort(hr.begin(), hr.end(), [](const BnbRange &a, const BnbRange &b) { return a.base_hi != b.base_hi ? a.base_hi < b.base_hi : a.base_lo < b.base_lo; });
    std::vector<std::pair<u128, u128>> joined;
    long double leaves = 0;
    for (const BnbRange &g : hr) {
        const u128 b = ((u128)g.base_hi << 64) | g.base_lo, c = ((u128)g.count_hi << 64) | g.count_lo;
        leaves += (long double)c;
        if (!joined.empty() && joined.back().first + joined.back().second == b) joined.back().second += c;
        else joined.push_back({b, c});
    }
    if (stats) {
        unsigned long long tot[BNB_STAT_STRIDE] = {0};
        for (int sl = 0; sl < BNB_STAT_SLOTS; sl++)
            for (int k = 0; k < BNB_STAT_STRIDE; k++) tot[k] += hs[(size_t)sl * BNB_STAT_STRIDE + k];
        stats->nodes_expanded = expanded;
        stats->children_bounded = tot[0];
        stats->newton_iterations = tot[1];
        stats->children_pruned = tot[2];
        stats->children_collinear = tot[3];
        stats->children_unbounded = tot[4];
        if (getenv("THETA_BNB_DEBUG")) fprintf(stderr, "bnb: open children: %llu not converged, %llu NaN / huge step, %llu ill-conditioned\n", tot[5], tot[6], tot[7]);
        stats->ranges_raw = ranges_written;
        stats->ranges = joined.size();
        stats->launches = launches;
        stats->max_frontier = max_frontier;
        stats->leaves = (double)leaves;
        stats->kernel_ms = ms;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        stats->emit_depth = emit_depth;
        stats->complete = budget_hit ? 0 : 1;
        stats->chunk = chunk;
        for (int k = 0; k <= m && k <= THETA_MAX_M; k++) stats->frontier[k] = frontier[k];
    }
    *n_out = joined.size();
    if (budget_hit) {
        theta_set_error("theta_bnb: node budget (%llu) spent at %llu nodes expanded; the walk is incomplete", (unsigned long long)max_nodes,
                        (unsigned long long)expanded);
        return THETA_ERR_OVERFLOW;
    }
    if (joined.size() > cap) {
        theta_set_error("theta_bnb: %zu rank ranges but capacity is %llu", joined.size(), (unsigned long long)cap);
        return THETA_ERR_CAPACITY;
    }
    for (size_t i = 0; i < joined.size(); i++) {
        ranges[4 * i] = (uint64_t)joined[i].first;
        ranges[4 * i + 1] = (uint64_t)(joined[i].first >> 64);
        ranges[4 * i + 2] = (uint64_t)joined[i].second;
        ranges[4 * i + 3] = (uint64_t)(joined[i].second >> 64);
    }
    return THETA_OK;
}

// ---- branch and bound over the mixture space (bnb.hip, second half) ------------------------------------------------------------------
// The lines of the alphabet's grid [0, K]^2 that hold at least two rows of the alphabet: the rank-deficient matrices are those whose
// rows all lie on one of them (a matrix of ONE repeated row lies on several).  Canonical direction: dx > 0, or dx = 0 and dy = 1.
static void mix_build_lines(const theta_problem *p, std::vector<MixLine> &lines, unsigned char (&slot_of)[256]) {
    memset(slot_of, 0xff, sizeof(slot_of));
    const N3Host &H = p->n3h;
    int K = 0;
    for (int s = 0; s < p->n3.Q; s++) {
        slot_of[H.rowtab[s]] = (unsigned char)s;
        K = std::max(K, std::max<int>(H.rowtab[s] & 15, H.rowtab[s] >> 4));
    }
    auto gcd = [](int a, int b) { a = abs(a); b = abs(b); while (b) { int t = a % b; a = b; b = t; } return a; };
    lines.clear();
    for (int dx = 0; dx <= K; dx++)
        for (int dy = (dx == 0 ? 1 : -K); dy <= (dx == 0 ? 1 : K); dy++) {
            if (gcd(dx, dy) != 1) continue;
            for (int x0 = 0; x0 <= K; x0++)
                for (int y0 = 0; y0 <= K; y0++) {
                    const int px = x0 - dx, py = y0 - dy;
                    if (px >= 0 && px <= K && py >= 0 && py <= K) continue;           // not the line's first point
                    int T = 0, rows = 0;
                    for (int x = x0, y = y0; x >= 0 && x <= K && y >= 0 && y <= K; x += dx, y += dy) {
                        T++;
                        if (slot_of[x | (y << 4)] != 0xff) rows++;
                    }
                    if (rows < 2) continue;
                    MixLine L;
                    L.x0 = (signed char)x0; L.y0 = (signed char)y0; L.dx = (signed char)dx; L.dy = (signed char)dy; L.T = T;
                    lines.push_back(L);
                }
        }
}

// The buffers of the search live with the problem (round 5 allocated and freed 3.4 GB per call, five or six calls per search).
static int mix_buffers(theta_problem *p, hipStream_t st) {
    if (p->d_mix_ctr.p) return THETA_OK;
    int rc;
    p->mix_chunk = 1u << 15;
    if (const char *e = getenv("THETA_MIX_CHUNK")) p->mix_chunk = (unsigned)std::max(64, atoi(e));
    p->mix_stack_cap = 1ull << 22;        // 256 MB: depth first, a few chunks per level of the tree
    p->mix_leaf_cap = 1ull << 21;         // 128 MB; walked and emptied whenever a batch could fill it
    if (const char *e = getenv("THETA_MIX_STACK")) p->mix_stack_cap = std::max<unsigned long long>(4ull * p->mix_chunk, strtoull(e, nullptr, 10));
    if (const char *e = getenv("THETA_MIX_LEAVES")) p->mix_leaf_cap = std::max<unsigned long long>(4ull * p->mix_chunk, strtoull(e, nullptr, 10));
    unsigned char slot_of[256];
    mix_build_lines(p, p->mix_lines, slot_of);
    std::vector<MixIv> iv((size_t)p->m);
    for (int i = 0; i < p->m; i++) {
        const double r = p->h_r[i], N = p->h_rN[i];
        iv[i].N = N;
        iv[i].r = r;
        iv[i].ts = r / N;
        iv[i].lnN = (double)logl((long double)N);
        iv[i].phimin = r > 0.0 ? (double)((long double)r - (long double)r * logl((long double)r)) : 0.0;
    }
    if ((rc = p->d_mix_stack.alloc(p->mix_stack_cap * sizeof(MixCell))) || (rc = p->d_mix_work.alloc(2ull * p->mix_chunk * sizeof(MixCell) + 2 * MIX_BEAM_LIMIT * sizeof(unsigned short))) ||
        (rc = p->d_mix_leaves.alloc(p->mix_leaf_cap * sizeof(MixCell))) || (rc = p->d_mix_ctr.alloc(MIX_NCTR * sizeof(unsigned long long))) ||
        (rc = upload(p->d_mix_slot, slot_of, sizeof(slot_of), st)) || (rc = upload(p->d_mix_iv, iv.data(), iv.size() * sizeof(MixIv), st)) ||
        (rc = upload(p->d_mix_lines, p->mix_lines.data(), std::max<size_t>(1, p->mix_lines.size()) * sizeof(MixLine), st)))
        return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return THETA_OK;
}

// A stack of boxes, walked depth first in chunks; boxes narrower than `leaf_rel` (relative to the mean read-depth ratio, per unit of
// copy number) are leaves, whose matrices are listed by a budgeted depth-first walk.  Returns the matrices as {a, b} per interval,
// without duplicates, in the reference's enumeration order (lexicographic in the slots).  mode: THETA_MIX_PROPOSE |
// THETA_MIX_LINES | THETA_MIX_LINES_ONLY (include/theta_hip.h).
extern "C" int theta_mix_search(theta_problem *p, double threshold, double leaf_rel, int mode, uint64_t cap, uint8_t *C, uint64_t *n_out,
                                theta_mix_stats *stats) {
    if (!p || !n_out || (cap > 0 && !C)) {
        theta_set_error("theta_mix_search: null argument");
        return THETA_ERR_ARG;
    }
    *n_out = 0;
    if (stats) memset(stats, 0, sizeof(*stats));
    if (p->n != 3) {
        theta_set_error("theta_mix_search: n = 3 only");
        return THETA_ERR_ARG;
    }
    if (!(threshold == threshold) || !(leaf_rel > 0.0 && leaf_rel < 1.0) || (mode & ~15)) {
        theta_set_error("theta_mix_search: bad threshold, leaf size or mode");
        return THETA_ERR_ARG;
    }
    const bool propose = mode & THETA_MIX_PROPOSE, lines_only = mode & THETA_MIX_LINES_ONLY, with_lines = (mode & THETA_MIX_LINES) || lines_only;
    const bool dive = mode & THETA_MIX_DIVE;
    if ((propose && with_lines) || (dive && !propose)) {
        theta_set_error("theta_mix_search: proposals come from the whole alphabet's boxes (no THETA_MIX_LINES with THETA_MIX_PROPOSE), and a dive "
                        "(THETA_MIX_DIVE) only proposes");
        return THETA_ERR_ARG;
    }
    const unsigned beam = dive ? (unsigned)std::min<uint64_t>(std::max<uint64_t>(p->opt_mix_beam, 8), MIX_BEAM_LIMIT) : 0u;
    HIP_ENTER(p->ctx->device);
    hipStream_t st = p->ctx->stream;
    const int m = p->m;
    const auto t_start = std::chrono::steady_clock::now();
    int rc;
    if ((rc = mix_buffers(p, st))) return rc;
    long double N = 0, Rt = 0;
    double rNmin = INFINITY;
    for (int i = 0; i < m; i++) {
        N += p->h_rN[i];
        Rt += p->h_r[i];
        rNmin = std::min(rNmin, p->h_rN[i]);
    }
    const int Q_ = p->n3.Q;
    MixArgs A;
    A.m = m;
    A.Q = p->n3.Q;
    A.tau = p->tau;
    A.r = p->n3.r;
    A.rN = p->n3.rN;
    A.rowtab = p->n3.rowtab;
    A.iv = (const MixIv *)p->d_mix_iv.p;
    A.slot_of = (const unsigned char *)p->d_mix_slot.p;
    A.lb = p->n3.lb;
    A.ub = p->n3.ub;
    A.cst = (double)(-Rt + (Rt > 0 ? Rt * logl(Rt) : 0.0L));
    A.thr = threshold;
    A.lines = (const MixLine *)p->d_mix_lines.p;
    A.n_lines = (int)p->mix_lines.size();
    A.shard_g = p->opt_mix_g;
    A.shard_G = dive ? 1 : p->opt_mix_G;              // (a dive is a few thousand boxes: every rank takes it whole, and alike)
    A.chunk = p->mix_chunk;
    A.dive = dive ? 1 : 0;
    A.dive_blend = p->opt_mix_blend;
    A.dive_niches = p->opt_mix_niches;
    const double tref = (double)(Rt / N);            // the mean read-depth ratio: c.v of a typical interval
    const int Kmax = std::max(1, p->n3.K);
    A.leaf[0] = leaf_rel * tref / std::max(1, p->tau);
    A.leaf[1] = A.leaf[2] = leaf_rel * tref / Kmax;
    A.leaf_line[0] = leaf_rel * tref;
    A.leaf_line[1] = leaf_rel * tref / Kmax;
    A.leaf_line[2] = 1.0;
    // the roots.  Whole alphabet: at the best scale sum_i rN_i c_i.v = Rtot, so tau v0 N <= Rtot and -- a tumour column with an entry
    // >= 1 -- v_j rN_min <= Rtot.  A line: every row in use has 0 < alpha + t beta <= U = Rtot / rN_min; with two distinct rows in use
    // |beta| <= U and -(T - 1) U <= alpha <= T U -- tightened below.  (The matrices of ONE repeated row have the same value at every
    // mixture, the model p_i = rN_i / N: they are listed and valued by the caller apart, Problem.constant_matrices.)
    std::vector<MixCell> roots;
    const double U = (double)(Rt / rNmin);
    // ... and tighter for the lines: the objective is a sum of per-interval terms, each at least its own minimum phi_i(r_i / rN_i), so
    // within the threshold NO interval's term may exceed its minimum by more than the slack S = thr - cst - sum_j phi_j(r_j / rN_j):
    // with u = rN_i q / r_i that is r_i (u - 1 - ln u) <= S, an interval [qlo_i, qhi_i] around the interval's ratio.  Every row in
    // use therefore has qmin <= q <= qmax (the extremes over the intervals), and with two distinct rows in use |beta| <= D = qmax - qmin,
    // qmin - (T - 1) D <= alpha <= qmax + (T - 1) D.  (Against 0 < q <= U this cuts the valleys along which ONE row of the line meets
    // the data -- the repeated-row matrices, valued apart -- from ~ N / rN_min boxes per scale to a handful: 14.0e6 of the 14.9e6 boxes
    // of config 4's final pass were theirs.)
    double qmin = 0.0, qmax = U, qhi_min = INFINITY;
    {
        long double sat = 0;
        for (int i = 0; i < m; i++) {
            const double r = p->h_r[i];
            sat += r > 0 ? (long double)r - (long double)r * logl((long double)r) : 0.0L;       // phi_i at its minimiser: N t = r
        }
        const double S = (double)((long double)threshold - (long double)A.cst - sat);
        if (S >= 0.0 && std::isfinite(S)) {
            double lo_all = INFINITY, hi_all = 0.0;
            for (int i = 0; i < m; i++) {
                const double r = p->h_r[i], Nn = p->h_rN[i];
                double qlo = 0.0, qhi = U;
                if (r > 0.0) {
                    const double sr = S / r;           // u - 1 - ln u <= sr
                    double a = 0.0, b = 1.0;           // the root below 1
                    for (int it = 0; it < 200; it++) {
                        const double u = 0.5 * (a + b);
                        if (u - 1.0 - log(u) > sr) a = u; else b = u;
                    }
                    qlo = a * (r / Nn);
                    a = 1.0;
                    b = 2.0 * sr + 12.0;               // ... and the one above
                    for (int it = 0; it < 200; it++) {
                        const double u = 0.5 * (a + b);
                        if (u - 1.0 - log(u) > sr) b = u; else a = u;
                    }
                    qhi = b * (r / Nn);
                } else {
                    qhi = S / Nn;                      // phi = N q <= S
                }
                lo_all = std::min(lo_all, qlo);
                hi_all = std::max(hi_all, qhi);
                qhi_min = std::min(qhi_min, qhi);
            }
            qmin = std::max(0.0, lo_all * (1.0 - 1e-9));
            qmax = std::min(U, hi_all * (1.0 + 1e-9));
            qhi_min = qhi_min * (1.0 + 1e-9);
        } else if (dive) {
            // a dive has no threshold to take the ranges from: its root is four times the largest ratio of the data (a heuristic like
            // the dive itself: the thresholded walk that follows has the rigorous root, and a guard should the dive have missed)
            double tmax = 0.0;
            for (int i = 0; i < m; i++) tmax = std::max(tmax, p->h_r[i] / p->h_rN[i]);
            double mult = 4.0;
            if (const char *e = getenv("THETA_MIX_DIVE_ROOT")) mult = std::max(1.5, atof(e));
            qmax = std::min(U, mult * tmax);
            qhi_min = qmax;
        }
    }
    const double Dq = std::max(qmax - qmin, 0.0);
    if (!lines_only) {
        MixCell root;
        memset(&root, 0, sizeof(root));
        // (the same ranges bound the whole alphabet's root: tau v0 <= c_i.v <= qhi_i for EVERY interval, and a v_j that multiplies a
        // copy number >= 1 somewhere is at most the largest qhi; a column of zeros leaves its v_j without influence, any value does.
        // Against Rtot / rN_min that is 300 times less per side on the bench's data: sixteen levels of one or two boxes each.)
        root.hi[0] = std::min((double)(Rt / (std::max(1, p->tau) * N)), qhi_min / std::max(1, p->tau));
        root.hi[1] = root.hi[2] = qmax;
        roots.push_back(root);
    }
    if (with_lines)
        for (size_t l = 0; l < p->mix_lines.size(); l++) {
            MixCell root;
            memset(&root, 0, sizeof(root));
            const int T = p->mix_lines[l].T;
            root.lo[0] = qmin - (double)(T - 1) * Dq;
            root.hi[0] = qmax + (double)(T - 1) * Dq;
            root.lo[1] = -Dq;
            root.hi[1] = Dq;
            root.line = (unsigned short)(l + 1);
            roots.push_back(root);
        }
    // a sharded search (options mix_shard_rank / mix_shard_world): every rank walks the large boxes alike, then keeps its own
    A.shard_mult = 8.0;
    if (const char *e = getenv("THETA_MIX_SHARD_MULT")) A.shard_mult = std::max(2.0, atof(e));
    if (roots.size() > p->mix_stack_cap) {
        theta_set_error("theta_mix_search: %zu roots", roots.size());
        return THETA_ERR_CAPACITY;
    }
    unsigned long long h_ctr[MIX_NCTR];
    memset(h_ctr, 0, sizeof(h_ctr));
    h_ctr[MIX_TOP0] = roots.size();
    h_ctr[MIX_MINB] = h_ctr[MIX_MINB_LINE] = ~0ull;
    unsigned long long *d_ctr = (unsigned long long *)p->d_mix_ctr.p;
    MixCell *d_stack = (MixCell *)p->d_mix_stack.p, *d_work = (MixCell *)p->d_mix_work.p, *d_leaves = (MixCell *)p->d_mix_leaves.p;
    HIP_TRY(hipMemcpyAsync(d_ctr, h_ctr, sizeof(h_ctr), hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_stack, roots.data(), roots.size() * sizeof(MixCell), hipMemcpyHostToDevice, st));
    HIP_TRY(hipEventRecord(p->ctx->ev0, st));
    // the list of matrices grows with the need (raw listings: every matrix is listed by each leaf and corner whose budget it meets)
    if (!propose && !p->d_mix_mat.p) {
        p->mix_mat_cap = std::max<uint64_t>(1ull << 16, (64ull << 20) / (size_t)m);
        if ((rc = p->d_mix_mat.alloc(p->mix_mat_cap * (size_t)m))) return rc;
        // the table of records seen (mix_list_kernel): 2^24 indices, 64 MB -- room for eight million distinct matrices; beyond, raw
        p->mix_seen_mask = (1ull << 24) - 1;
        if ((rc = p->d_mix_seen.alloc((p->mix_seen_mask + 1) * sizeof(unsigned)))) return rc;
    }
    if (!propose) HIP_TRY(hipMemsetAsync(p->d_mix_seen.p, 0, p->d_mix_seen.bytes, st));
    const bool debug = getenv("THETA_BNB_DEBUG") != nullptr;
    const uint64_t max_tested = p->opt_mix_max_boxes;
    int parity = 0;
    uint64_t syncs = 0;
    // Walk the leaves collected so far (and empty the list).  The list of matrices is redone with a larger buffer when it overflows
    // (the walk is deterministic; the counter is put back first).
    // The leaves are walked MIX_LEAF_LAUNCH at a time: a walk over the intervals is a serial affair of up to mix_max_steps steps per
    // (leaf, corner), and a launch over a million leaves of a flat likelihood would hold the device for minutes (round 6 lost a box
    // to one).  A launch of 4096 leaves is a second or two at the very worst; the host looks at the counters and the clock in between.
    const unsigned long long MIX_LEAF_LAUNCH = 4096;
    auto walk_leaves = [&](unsigned long long n_leaves) -> int {
        if (!n_leaves) return THETA_OK;
        for (unsigned long long first = 0; first < n_leaves; first += MIX_LEAF_LAUNCH) {
            const unsigned long long part = std::min<unsigned long long>(MIX_LEAF_LAUNCH, n_leaves - first);
            const unsigned long long before = h_ctr[MIX_LISTED];
            for (;;) {
                mix_launch_list(A, d_leaves + first, part, (unsigned char *)p->d_mix_mat.p, p->mix_mat_cap, ~0ull, p->opt_mix_max_steps, d_ctr,
                                (unsigned *)p->d_mix_seen.p, p->mix_seen_mask, st);
                unsigned long long got[2];
                HIP_TRY(hipMemcpyAsync(got, d_ctr + MIX_LISTED, sizeof(got), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                HIP_TRY(hipGetLastError());
                syncs++;
                if (got[1] > 0) {
                    theta_set_error("theta_mix_search: %llu walks over the intervals of a leaf did not finish within their %llu steps (%llu matrices listed so "
                                    "far): the likelihood is too flat there (option mix_max_steps)", got[1], (unsigned long long)p->opt_mix_max_steps, got[0]);
                    return THETA_ERR_CAPACITY;
                }
                if (got[0] <= p->mix_mat_cap) {
                    h_ctr[MIX_LISTED] = got[0];
                    break;
                }
                // grow: what was listed before this launch is kept
                size_t free_b = 0, total_b = 0;
                (void)hipMemGetInfo(&free_b, &total_b);
                const uint64_t want = std::max<uint64_t>(got[0] + got[0] / 4, 2 * p->mix_mat_cap);
                if ((size_t)want * m > free_b / 2 + p->mix_mat_cap * (size_t)m) {
                    theta_set_error("theta_mix_search: %llu matrices listed within the threshold (%.1f GB of records): it is too far above the minimum",
                                    (unsigned long long)got[0], (double)got[0] * m / 1e9);
                    return THETA_ERR_CAPACITY;
                }
                DevBuf bigger;
                if ((rc = bigger.alloc((size_t)want * m))) return rc;
                if (before) HIP_TRY(hipMemcpyAsync(bigger.p, p->d_mix_mat.p, (size_t)before * m, hipMemcpyDeviceToDevice, st));
                HIP_TRY(hipMemcpyAsync(d_ctr + MIX_LISTED, &before, sizeof(before), hipMemcpyHostToDevice, st));
                // (the table of seen records points into the list: the entries of the launch that overflowed would point at records about
                // to be rewritten -- the table starts afresh; matrices listed before come once more at worst, the host's sort takes them out)
                if (p->d_mix_seen.p) HIP_TRY(hipMemsetAsync(p->d_mix_seen.p, 0, p->d_mix_seen.bytes, st));
                HIP_TRY(hipStreamSynchronize(st));
                std::swap(p->d_mix_mat.p, bigger.p);
                std::swap(p->d_mix_mat.bytes, bigger.bytes);
                p->mix_mat_cap = want;
                if (debug) fprintf(stderr, "mix: list buffer grown to %llu matrices\n", (unsigned long long)want);
            }
            const double spent = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
            if (p->opt_mix_max_ms > 0.0 && spent > p->opt_mix_max_ms && first + part < n_leaves) {
                theta_set_error("theta_mix_search: %.0f ms spent (option mix_max_ms) with %llu leaves still to walk, %llu matrices listed: the "
                                "threshold is too far above the minimum", spent, n_leaves - first - part, (unsigned long long)h_ctr[MIX_LISTED]);
                return THETA_ERR_CAPACITY;
            }
        }
        const unsigned long long zero = 0;
        HIP_TRY(hipMemcpyAsync(d_ctr + MIX_LEAVES, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
        h_ctr[MIX_LEAVES] = 0;
        return THETA_OK;
    };
    // batches of iterations: as many as the leaf list has room for (an iteration adds at most 2 x chunk leaves)
    for (;;) {
        const unsigned long long room = p->mix_leaf_cap - h_ctr[MIX_LEAVES];
        int batch = (int)std::min<unsigned long long>(propose ? 24 : room / (2ull * A.chunk), 24);
        if (batch < 1) {
            if ((rc = walk_leaves(h_ctr[MIX_LEAVES]))) return rc;
            continue;
        }
        if (syncs == 0 && !dive) batch = std::min(batch, 12);         // (the first cuts: a handful of boxes each)
        if (dive) batch = 40;                                          // (a beam's worth of boxes per level: the launches are the cost)
        const unsigned long long top0 = h_ctr[MIX_TOP0 + parity];
        for (int it = 0; it < batch; it++) {
            unsigned long long n_max = it < 40 ? top0 << it : ~0ull;     // (an iteration at most doubles the stack)
            if (dive) n_max = std::min<unsigned long long>(n_max, beam);
            mix_launch_iteration(A, d_stack, p->mix_stack_cap, d_work, d_leaves, p->mix_leaf_cap, d_ctr, parity, propose ? 1 : 0, n_max, beam, st);
            parity ^= 1;
        }
        HIP_TRY(hipMemcpyAsync(h_ctr, d_ctr, sizeof(h_ctr), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipGetLastError());
        syncs++;
        const unsigned long long top = h_ctr[MIX_TOP0 + parity];
        if (debug)
            fprintf(stderr, "mix: %llu iterations, %llu boxes tested, %llu on the stack (deepest %llu), %llu leaves (%llu of lines)\n", h_ctr[MIX_ITERS], h_ctr[MIX_TESTED],
                    top, h_ctr[MIX_MAXTOP], h_ctr[MIX_LEAVES_ALL], h_ctr[MIX_LEAVES_LINE]);
        auto fill = [&]() {
            if (!stats) return;
            stats->boxes_tested = h_ctr[MIX_TESTED];
            stats->levels = h_ctr[MIX_ITERS];
            stats->max_boxes = h_ctr[MIX_MAXTOP];
            stats->leaves = h_ctr[MIX_LEAVES_ALL];
            stats->line_leaves = h_ctr[MIX_LEAVES_LINE];
            stats->lines = with_lines ? p->mix_lines.size() : 0;
            stats->syncs = syncs;
            stats->min_bound = h_ctr[MIX_MINB] == ~0ull ? INFINITY : mix_unord(h_ctr[MIX_MINB]);
            stats->min_bound_lines = h_ctr[MIX_MINB_LINE] == ~0ull ? INFINITY : mix_unord(h_ctr[MIX_MINB_LINE]);
        };
        if ((h_ctr[MIX_OVERFLOW] & 2ull) && !propose) {       // (cannot happen: a batch is sized for the room the list has)
            fill();
            theta_set_error("theta_mix_search: leaves were dropped");
            return THETA_ERR_HIP;
        }
        if (h_ctr[MIX_OVERFLOW] & 1ull) {
            fill();
            theta_set_error("theta_mix_search: more than %llu boxes on the stack (%llu tested): the threshold is too far above the minimum",
                            (unsigned long long)p->mix_stack_cap, (unsigned long long)h_ctr[MIX_TESTED]);
            return THETA_ERR_CAPACITY;
        }
        if (max_tested && h_ctr[MIX_LEAVES_ALL] > std::max<uint64_t>(max_tested / 128, 4096)) {
            // (a bounded walk: its leaves would be walked over the intervals next, and with a threshold this far up each holds matrices
            // by the thousand)
            fill();
            theta_set_error("theta_mix_search: %llu leaves within the threshold after %llu boxes (option mix_max_boxes): it is too far above the minimum",
                            (unsigned long long)h_ctr[MIX_LEAVES_ALL], (unsigned long long)h_ctr[MIX_TESTED]);
            return THETA_ERR_CAPACITY;
        }
        if (p->opt_mix_max_ms > 0.0 && top > 0 &&
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count() > p->opt_mix_max_ms) {
            fill();
            theta_set_error("theta_mix_search: %.0f ms spent (option mix_max_ms) with %llu boxes still on the stack, %llu tested: the threshold is too far "
                            "above the minimum", p->opt_mix_max_ms, top, (unsigned long long)h_ctr[MIX_TESTED]);
            return THETA_ERR_CAPACITY;
        }
        if (max_tested && h_ctr[MIX_TESTED] > max_tested && top > 0) {
            fill();
            theta_set_error("theta_mix_search: %llu boxes tested and %llu still on the stack (option mix_max_boxes): the threshold is too far above the minimum",
                            (unsigned long long)h_ctr[MIX_TESTED], top);
            return THETA_ERR_CAPACITY;
        }
        if (top == 0) {
            fill();
            break;
        }
    }
    const uint64_t n_leaves_all = h_ctr[MIX_LEAVES_ALL];
    if (propose) {
        // PROPOSALS instead of the exhaustive list: for the `cap` leaves of smallest bound, the matrix that fits the leaf's centre
        // best -- per interval the row that minimises phi_i(c.v) --; valued by the caller, the best of them lowers the threshold
        // of the next, finer, call
        const uint64_t take_max = std::min<uint64_t>(std::min<uint64_t>(h_ctr[MIX_LEAVES], p->mix_leaf_cap), 1ull << 20);
        std::vector<MixCell> hl(take_max);
        if (take_max) HIP_TRY(hipMemcpy(hl.data(), d_leaves, take_max * sizeof(MixCell), hipMemcpyDeviceToHost));
        const uint64_t want = std::min<uint64_t>(cap, take_max);
        std::partial_sort(hl.begin(), hl.begin() + want, hl.end(), [](const MixCell &a, const MixCell &b) { return a.lb < b.lb; });
        std::vector<std::vector<unsigned char>> seen;
        const N3Host &H = p->n3h;
        uint64_t nw = 0;
        for (uint64_t k = 0; k < want; k++) {
            std::vector<unsigned char> mat((size_t)m * 2);
            const double v0 = 0.5 * (hl[k].lo[0] + hl[k].hi[0]), v1 = 0.5 * (hl[k].lo[1] + hl[k].hi[1]), v2 = 0.5 * (hl[k].lo[2] + hl[k].hi[2]);
            for (int i = 0; i < m; i++) {
                // phi_i is convex in t = c.v with its minimum at r_i / rN_i: the best row is the nearest one on either side of it
                // (two logarithms per interval instead of one per row: 256 leaves x 200 intervals x 64 rows of them were half of a
                // pass's time at m = 200)
                const double tstar = p->h_r[i] / p->h_rN[i];
                double tb = -INFINITY, ta = INFINITY;
                int ab[2][2] = {{-1, -1}, {-1, -1}};
                for (int sidx = 0; sidx < Q_; sidx++) {
                    const int a = H.rowtab[sidx] & 15, b = H.rowtab[sidx] >> 4;
                    if (a < H.lb[i] || a > H.ub[i] || b < H.lb[i] || b > H.ub[i] || (p->tau - a) * (p->tau - b) < 0) continue;
                    const double t = p->tau * v0 + a * v1 + b * v2;
                    if (!(t > 0.0)) continue;
                    if (t <= tstar) {
                        if (t > tb) {
                            tb = t;
                            ab[0][0] = a;
                            ab[0][1] = b;
                        }
                    } else if (t < ta) {
                        ta = t;
                        ab[1][0] = a;
                        ab[1][1] = b;
                    }
                }
                auto phi = [&](double t) { return p->h_r[i] > 0 ? p->h_rN[i] * t - p->h_r[i] * log(p->h_rN[i] * t) : p->h_rN[i] * t; };
                int pick = ab[0][0] >= 0 ? 0 : 1;
                if (ab[0][0] >= 0 && ab[1][0] >= 0 && phi(ta) < phi(tb)) pick = 1;
                mat[2 * i] = (unsigned char)std::max(0, ab[pick][0]);
                mat[2 * i + 1] = (unsigned char)std::max(0, ab[pick][1]);
            }
            if (std::find(seen.begin(), seen.end(), mat) != seen.end()) continue;
            seen.push_back(mat);
            memcpy(C + nw * (size_t)m * 2, mat.data(), (size_t)m * 2);
            nw++;
        }
        *n_out = nw;
        if (stats) {
            stats->matrices = nw;
            stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        }
        return THETA_OK;
    }
    // the matrices of the leaves not walked yet
    if ((rc = walk_leaves(h_ctr[MIX_LEAVES]))) return rc;
    HIP_TRY(hipEventRecord(p->ctx->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, p->ctx->ev0, p->ctx->ev1));
    const uint64_t listed = h_ctr[MIX_LISTED];
    if (stats) {
        stats->leaves = n_leaves_all;
        stats->listed = listed;
        stats->kernel_ms = ms;
        stats->syncs = syncs;
    }
    std::vector<unsigned char> hm((size_t)listed * m);
    if (listed) HIP_TRY(hipMemcpy(hm.data(), p->d_mix_mat.p, hm.size(), hipMemcpyDeviceToHost));
    // without duplicates (neighbouring leaves and corners list the same matrix), in the reference's order: lexicographic in the slots
    std::vector<size_t> ix(listed);
    for (size_t i = 0; i < listed; i++) ix[i] = i;
    std::sort(ix.begin(), ix.end(), [&](size_t x, size_t y) { return memcmp(&hm[x * m], &hm[y * m], m) < 0; });
    ix.erase(std::unique(ix.begin(), ix.end(), [&](size_t x, size_t y) { return memcmp(&hm[x * m], &hm[y * m], m) == 0; }), ix.end());
    *n_out = ix.size();
    if (stats) {
        stats->matrices = ix.size();
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    }
    if (ix.size() > cap) {
        theta_set_error("theta_mix_search: %zu matrices but capacity is %llu", ix.size(), (unsigned long long)cap);
        return THETA_ERR_CAPACITY;        // (*n_out holds the size needed: the caller may come again)
    }
    // slots -> rows (a, b)
    for (size_t k = 0; k < ix.size(); k++)
        for (int i = 0; i < m; i++) {
            const unsigned rw = p->n3h.rowtab[hm[ix[k] * m + i]];
            C[(k * m + i) * 2] = (uint8_t)(rw & 15u);
            C[(k * m + i) * 2 + 1] = (uint8_t)(rw >> 4);
        }
    return THETA_OK;
}

extern "C" int theta_problem_hint(theta_problem *p, double nll_upper_bound) {
    if (!p) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    p->hint = (nll_upper_bound == nll_upper_bound) ? nll_upper_bound : INFINITY;
    return THETA_OK;
}

// ranks (+ optional bounds) and materialised matrices of one of the side lists of the last search
static int side_list_out(theta_problem *p, const std::vector<TieRecord> &sv, uint64_t dropped, const char *what, int cap,
                         uint64_t *rank, double *lbound, uint8_t *C, int *n_out) {
    if (!p || !n_out) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    *n_out = (int)sv.size();
    if (cap < 0) {   // query: number of entries that did not fit the device list
        *n_out = (int)std::min<uint64_t>(dropped, 0x7fffffff);
        return THETA_OK;
    }
    if (sv.empty()) return THETA_OK;
    if ((int)sv.size() > cap || !rank || !C) {
        theta_set_error("%zu %s but capacity is %d", sv.size(), what, cap);
        return THETA_ERR_CAPACITY;
    }
    HIP_ENTER(p->ctx->device);
    hipStream_t st = p->ctx->stream;
    size_t cb = (size_t)p->m * (p->n - 1);
    DevBuf d_rec, d_C;
    int rc;
    if ((rc = d_rec.alloc(sv.size() * sizeof(TieRecord)))) return rc;
    if ((rc = d_C.alloc(sv.size() * cb))) return rc;
    HIP_TRY(hipMemcpyAsync(d_rec.p, sv.data(), sv.size() * sizeof(TieRecord), hipMemcpyHostToDevice, st));
    if (p->n == 2) n2_launch_unrank_list(p->n2, (const TieRecord *)d_rec.p, (int)sv.size(), (unsigned char *)d_C.p, st);
    else n3_launch_unrank_list(p->n3, (const TieRecord *)d_rec.p, (int)sv.size(), (unsigned char *)d_C.p, st);
    HIP_TRY(hipMemcpyAsync(C, d_C.p, sv.size() * cb, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    for (size_t i = 0; i < sv.size(); i++) {
        rank[2 * i] = sv[i].rank_lo;
        rank[2 * i + 1] = sv[i].rank_hi;
        if (lbound) lbound[i] = sv[i].nll;
    }
    return THETA_OK;
}

extern "C" int theta_search_suspects(theta_problem *p, int cap, uint64_t *rank, double *lbound, uint8_t *C, int *n_out) {
    if (!p) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    if (cap > 0 && !lbound) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    return side_list_out(p, p->suspects, p->suspects_dropped, "suspects", cap, rank, lbound, C, n_out);
}

extern "C" int theta_search_degenerate(theta_problem *p, int cap, uint64_t *rank, uint8_t *C, int *n_out) {
    if (!p) {
        theta_set_error("null argument");
        return THETA_ERR_ARG;
    }
    return side_list_out(p, p->degenerate, p->degenerate_dropped, "degenerate candidates", cap, rank, nullptr, C, n_out);
}

extern "C" int theta_boundary_min(theta_ctx *ctx, int m, int tau, const int64_t *r, const int64_t *rN, int B,
                                  const uint8_t *C, double *bound) {
    if (!ctx || !r || !rN || !C || !bound || B < 0 || m < 1 || m > 4096) {
        theta_set_error("theta_boundary_min: bad argument");
        return THETA_ERR_ARG;
    }
    if (B == 0) return THETA_OK;
    HIP_ENTER(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<double> rd(m), rnd(m);
    for (int i = 0; i < m; i++) {
        rd[i] = (double)r[i];
        rnd[i] = (double)rN[i];
    }
    DevBuf d_r, d_rN, d_C, d_b;
    int rc;
    if ((rc = upload(d_r, rd.data(), m * sizeof(double), st))) return rc;
    if ((rc = upload(d_rN, rnd.data(), m * sizeof(double), st))) return rc;
    if ((rc = upload(d_C, C, (size_t)B * m * 2, st))) return rc;
    if ((rc = d_b.alloc((size_t)B * sizeof(double)))) return rc;
    batch_launch_boundary_min(m, tau, (const double *)d_r.p, (const double *)d_rN.p, B, (const unsigned char *)d_C.p,
                              (double *)d_b.p, st);
    HIP_TRY(hipMemcpyAsync(bound, d_b.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return THETA_OK;
}

// candidates [b, b + count) into DEVICE memory; kernel_ms (optional): HIP-event time of the kernels on the stream
static int enumerate_device(theta_problem *p, u128 b, uint64_t count, unsigned char *d_out, double *kernel_ms) {
    theta_ctx *ctx = p->ctx;
    hipStream_t st = ctx->stream;
    const size_t cb = (size_t)p->m * (p->n - 1);
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    if (p->n == 2) {
        n2_launch_enumerate(p->n2, (unsigned long long)b, count, d_out, st);
    } else {
        // the burst generator (n3_enum.hip: one contiguous output stream per wave) cuts its tasks at its own depth
        N3Dev PE = p->n3;
        // (THETA_ENUM_LEGACY, the lane-private generator kept as a second implementation, holds N3_MAX_M intervals: wider
        // problems -- also the internal callers, list_deficient and the NaN sweep -- always take the burst generator)
        const int burst_levels = (getenv("THETA_ENUM_LEGACY") && p->m <= N3_MAX_M) ? 0 : n3_enumerate_burst_levels(PE);
        if (burst_levels > 0) PE.L = burst_levels;
        // wave tasks of 8192 candidates (0.8 MB of output at m=50): a 2^28 request is 32768 tasks, 6.4 x the 5120 resident
        // waves, so the last round of blocks leaves little of the chip idle (measured: 16384 is as good at K=6, worse at
        // K=4; 65536 loses 15-40 %).  The lane-private generator unranks 64 chunk starts per prefix: fewer, longer tasks.
        uint64_t per_task = 8192;
        if (burst_levels == 0)
            while (per_task < 65536 && per_task * 8192 < count) per_task <<= 1;
        if (const char *e = getenv("THETA_ENUM_PER_TASK")) {
            const long v = atol(e);
            if (v >= 64 && v <= (1l << 24)) per_task = (uint64_t)v;
        }
        const uint64_t piece = (uint64_t)N3_MAX_TASKS * per_task;
        for (uint64_t off = 0; off < count; off += piece) {
            const uint64_t c = std::min<uint64_t>(piece, count - off);
            const int ntasks = (int)((c + per_task - 1) / per_task);
            n3_launch_tasks(PE, b + off, b + off + c, per_task, ntasks, (N3Task *)p->d_tasks.p, (unsigned *)p->d_stbuf.p, st);
            if (burst_levels > 0)
                n3_launch_enumerate_burst(PE, (const N3Task *)p->d_tasks.p, (const unsigned *)p->d_stbuf.p, ntasks, per_task,
                                          d_out + (size_t)off * cb, st);
            else
                n3_launch_enumerate(PE, (const N3Task *)p->d_tasks.p, (const unsigned *)p->d_stbuf.p, ntasks, per_task,
                                    d_out + (size_t)off * cb, st);
        }
    }
    HIP_TRY(hipEventRecord(ctx->ev2, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (kernel_ms) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev1, ctx->ev2));
        *kernel_ms = ms;
    }
    return THETA_OK;
}

static int enumerate_args(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, u128 &b) {
    if (p && p->n == 3 && p->m > N3_MAX_M && getenv("THETA_ENUM_LEGACY")) {
        theta_set_error("theta_enumerate: the lane-private n=3 generator (THETA_ENUM_LEGACY) holds at most %d intervals (m = %d)", N3_MAX_M, p->m);
        return THETA_ERR_ARG;
    }
    uint64_t re[2];
    u128 b0 = rank_begin ? mk128(rank_begin) : 0;
    u128 e0 = b0 + count;
    re[0] = (uint64_t)e0;
    re[1] = (uint64_t)(e0 >> 64);
    u128 e;
    return check_range(p, rank_begin, re, b, e);
}

extern "C" int theta_enumerate(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, uint8_t *out) {
    u128 b;
    int rc = enumerate_args(p, rank_begin, count, b);
    if (rc) return rc;
    if (count == 0) return THETA_OK;
    if (!out) {
        theta_set_error("null output");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(p->ctx->device);
    size_t cb = (size_t)p->m * (p->n - 1);
    DevBuf d_C;
    rc = d_C.alloc(count * cb);
    if (rc) return rc;
    rc = enumerate_device(p, b, count, (unsigned char *)d_C.p, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, d_C.p, count * cb, hipMemcpyDeviceToHost, p->ctx->stream));
    HIP_TRY(hipStreamSynchronize(p->ctx->stream));
    return THETA_OK;
}

extern "C" int theta_enumerate_device(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, void *d_out,
                                      double *kernel_ms) {
    u128 b;
    int rc = enumerate_args(p, rank_begin, count, b);
    if (rc) return rc;
    if (kernel_ms) *kernel_ms = 0.0;
    if (count == 0) return THETA_OK;
    if (!d_out) {
        theta_set_error("null output");
        return THETA_ERR_ARG;
    }
    if (((uintptr_t)d_out & 3u) != 0) {      // the generators store 32-bit (even m) / 16-bit (odd m) row units
        theta_set_error("theta_enumerate_device: the output pointer must be 4-byte aligned");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(p->ctx->device);
    return enumerate_device(p, b, count, (unsigned char *)d_out, kernel_ms);
}

// ---- materialised operators -----------------------------------------------------------------------
extern "C" int theta_solve_batch(theta_ctx *ctx, int n, int m, int tau, const int64_t *r, const int64_t *rN,
                                 double max_normal, int B, const uint8_t *C, uint8_t *ok, double *mu, double *nll,
                                 double *vals) {
    if (!ctx || !r || !rN || !C || !ok || !mu || !nll || B < 0) {
        theta_set_error("theta_solve_batch: null argument");
        return THETA_ERR_ARG;
    }
    if ((n != 2 && n != 3) || m < 1 || m > 4096) {
        theta_set_error("theta_solve_batch: bad shape n=%d m=%d", n, m);
        return THETA_ERR_ARG;
    }
    if (B == 0) return THETA_OK;
    HIP_ENTER(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<double> rd(m), rnd(m);
    for (int i = 0; i < m; i++) {
        rd[i] = (double)r[i];
        rnd[i] = (double)rN[i];
    }
    size_t cb = (size_t)m * (n - 1);
    DevBuf d_r, d_rN, d_C, d_ok, d_mu, d_nll, d_vals;
    int rc;
    if ((rc = upload(d_r, rd.data(), m * sizeof(double), st))) return rc;
    if ((rc = upload(d_rN, rnd.data(), m * sizeof(double), st))) return rc;
    if ((rc = upload(d_C, C, (size_t)B * cb, st))) return rc;
    if ((rc = d_ok.alloc(B))) return rc;
    if ((rc = d_mu.alloc((size_t)B * n * sizeof(double)))) return rc;
    if ((rc = d_nll.alloc((size_t)B * sizeof(double)))) return rc;
    if (vals && (rc = d_vals.alloc((size_t)B * m * sizeof(double)))) return rc;
    batch_launch_solve(n, m, tau, (const double *)d_r.p, (const double *)d_rN.p, max_normal, B, (const unsigned char *)d_C.p,
                       (unsigned char *)d_ok.p, (double *)d_mu.p, (double *)d_nll.p, vals ? (double *)d_vals.p : nullptr, st);
    HIP_TRY(hipMemcpyAsync(ok, d_ok.p, B, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(mu, d_mu.p, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(nll, d_nll.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    if (vals) HIP_TRY(hipMemcpyAsync(vals, d_vals.p, (size_t)B * m * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return THETA_OK;
}

static int score_batch_impl(theta_ctx *ctx, int n, int m, int B, const double *Cw, const double *mu, const double *r,
                            int r_per_matrix, double *nll, double *vals, uint8_t *valid) {
    if (!ctx || !Cw || !mu || !r || !nll || B < 0) {
        theta_set_error("theta_score_batch: null argument");
        return THETA_ERR_ARG;
    }
    if ((n != 2 && n != 3) || m < 1) {
        theta_set_error("theta_score_batch: bad shape n=%d m=%d", n, m);
        return THETA_ERR_ARG;
    }
    if (B == 0) return THETA_OK;
    HIP_ENTER(ctx->device);
    hipStream_t st = ctx->stream;
    DevBuf d_C, d_mu, d_r, d_nll, d_vals, d_valid;
    int rc;
    if ((rc = upload(d_C, Cw, (size_t)B * m * n * sizeof(double), st))) return rc;
    if ((rc = upload(d_mu, mu, (size_t)B * n * sizeof(double), st))) return rc;
    if ((rc = upload(d_r, r, (size_t)m * (r_per_matrix ? (size_t)B : 1) * sizeof(double), st))) return rc;
    if ((rc = d_nll.alloc((size_t)B * sizeof(double)))) return rc;
    if (vals && (rc = d_vals.alloc((size_t)B * m * sizeof(double)))) return rc;
    if (valid && (rc = d_valid.alloc((size_t)B * m))) return rc;
    batch_launch_score(n, m, B, (const double *)d_C.p, (const double *)d_mu.p, (const double *)d_r.p, r_per_matrix ? m : 0,
                       (double *)d_nll.p, vals ? (double *)d_vals.p : nullptr, valid ? (unsigned char *)d_valid.p : nullptr, st);
    HIP_TRY(hipMemcpyAsync(nll, d_nll.p, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, st));
    if (vals) HIP_TRY(hipMemcpyAsync(vals, d_vals.p, (size_t)B * m * sizeof(double), hipMemcpyDeviceToHost, st));
    if (valid) HIP_TRY(hipMemcpyAsync(valid, d_valid.p, (size_t)B * m, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    return THETA_OK;
}

extern "C" int theta_score_batch(theta_ctx *ctx, int n, int m, int B, const double *Cw, const double *mu,
                                 const double *r, double *nll, double *vals, uint8_t *valid) {
    return score_batch_impl(ctx, n, m, B, Cw, mu, r, 0, nll, vals, valid);
}

extern "C" int theta_score_batch_rows(theta_ctx *ctx, int n, int m, int B, const double *Cw, const double *mu,
                                      const double *r, double *nll, double *vals, uint8_t *valid) {
    return score_batch_impl(ctx, n, m, B, Cw, mu, r, 1, nll, vals, valid);
}

static double host_sum(const double *v, int m) {
    double s = 0.0;
    for (int i = 0; i < m; i++) s += v[i];
    return s;
}
// smallest / largest weight (NaN if any is): the plain scorer's per-candidate test that every row term is a positive normal number
static double host_min(const double *v, int m) {
    double s = INFINITY;
    for (int i = 0; i < m; i++) s = (v[i] < s || v[i] != v[i]) ? v[i] : s;
    return s;
}
static double host_max(const double *v, int m) {
    double s = -INFINITY;
    for (int i = 0; i < m; i++) s = (v[i] > s || v[i] != v[i]) ? v[i] : s;
    return s;
}
// sum r_i ln w_i for the n = 2 table-driven scorer (batch.hip): valid iff every weight is a positive finite number
static bool host_rlogw(const double *w, const double *r, int m, double &out) {
    double s = 0.0;
    for (int i = 0; i < m; i++) {
        if (!(w[i] > 0.0) || !(w[i] < INFINITY)) return false;
        s += r[i] * log(w[i]);
    }
    out = s;
    return true;
}

// ---- device memory the caller owns, for chains of operators that stay on the GPU ----------------------------------
extern "C" int theta_device_alloc(theta_ctx *ctx, size_t bytes, void **out) {
    if (!ctx || !out) {
        theta_set_error("theta_device_alloc: null argument");
        return THETA_ERR_ARG;
    }
    HIP_ENTER(ctx->device);
    HIP_TRY(hipMalloc(out, bytes ? bytes : 8));
    return THETA_OK;
}

extern "C" int theta_device_free(theta_ctx *ctx, void *p) {
    if (!ctx) {
        theta_set_error("null context");
        return THETA_ERR_ARG;
    }
    if (!p) return THETA_OK;
    HIP_ENTER(ctx->device);
    HIP_TRY(hipFree(p));
    return THETA_OK;
}

extern "C" int theta_device_copy(theta_ctx *ctx, void *dst, const void *src, size_t bytes, int to_device) {
    if (!ctx || (bytes && (!dst || !src))) {
        theta_set_error("theta_device_copy: null argument");
        return THETA_ERR_ARG;
    }
    if (!bytes) return THETA_OK;
    HIP_ENTER(ctx->device);
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return THETA_OK;
}

extern "C" int theta_solve_batch_device(theta_ctx *ctx, int n, int m, int tau, const int64_t *r, const int64_t *rN,
                                        double max_normal, int B, const void *d_C, void *d_ok, void *d_mu, void *d_nll,
                                        void *d_vals, double *kernel_ms) {
    if (!ctx || !r || !rN || !d_C || !d_ok || !d_mu || !d_nll || B < 0) {
        theta_set_error("theta_solve_batch_device: null argument");
        return THETA_ERR_ARG;
    }
    if ((n != 2 && n != 3) || m < 1 || m > 4096) {
        theta_set_error("theta_solve_batch_device: bad shape n=%d m=%d", n, m);
        return THETA_ERR_ARG;
    }
    if (kernel_ms) *kernel_ms = 0.0;
    if (B == 0) return THETA_OK;
    HIP_ENTER(ctx->device);
    hipStream_t st = ctx->stream;
    std::vector<double> rd(m), rnd(m);
    for (int i = 0; i < m; i++) {
        rd[i] = (double)r[i];
        rnd[i] = (double)rN[i];
    }
    DevBuf d_r, d_rN;
    int rc;
    if ((rc = upload(d_r, rd.data(), m * sizeof(double), st))) return rc;
    if ((rc = upload(d_rN, rnd.data(), m * sizeof(double), st))) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    batch_launch_solve(n, m, tau, (const double *)d_r.p, (const double *)d_rN.p, max_normal, B, (const unsigned char *)d_C,
                       (unsigned char *)d_ok, (double *)d_mu, (double *)d_nll, (double *)d_vals, st);
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (kernel_ms) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *kernel_ms = ms;
    }
    return THETA_OK;
}

extern "C" int theta_score_masked_device(theta_ctx *ctx, int n, int m, int tau, int B, int S, const void *d_C, const double *w,
                                         const double *r, const void *d_mu, const uint64_t *mask, void *d_nll,
                                         double *kernel_ms) {
    if (!ctx || !d_C || !w || !r || !d_mu || !d_nll || B < 0 || S < 1) {
        theta_set_error("theta_score_masked_device: bad argument");
        return THETA_ERR_ARG;
    }
    if ((n != 2 && n != 3) || m < 1 || m > 256) {
        theta_set_error("theta_score_masked_device: need n in {2,3}, 1 <= m <= 256");
        return THETA_ERR_ARG;
    }
    if (kernel_ms) *kernel_ms = 0.0;
    if (B == 0) return THETA_OK;
    HIP_ENTER(ctx->device);
    hipStream_t st = ctx->stream;
    int words = (m + 63) / 64;
    DevBuf d_w, d_r, d_mask, d_rsum;
    int rc;
    if ((rc = upload(d_w, w, (size_t)m * sizeof(double), st))) return rc;
    if ((rc = upload(d_r, r, (size_t)m * sizeof(double), st))) return rc;
    if (mask && (rc = upload(d_mask, mask, (size_t)S * words * sizeof(uint64_t), st))) return rc;
    if ((rc = d_rsum.alloc((size_t)S * sizeof(double)))) return rc;
    double rlogw = 0.0;
    const bool rlogw_ok = host_rlogw(w, r, m, rlogw);
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    batch_launch_score_masked(n, m, tau, B, S, (const unsigned char *)d_C, (const double *)d_w.p, (const double *)d_r.p,
                              (const double *)d_mu, mask ? (const unsigned long long *)d_mask.p : nullptr, (double *)d_nll,
                              (double *)d_rsum.p, st, host_sum(r, m), true, rlogw, rlogw_ok, host_sum(w, m), host_min(w, m), host_max(w, m));
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (kernel_ms) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *kernel_ms = ms;
    }
    return THETA_OK;
}

extern "C" int theta_score_masked(theta_ctx *ctx, int n, int m, int tau, int B, int S, const uint8_t *C, const double *w,
                                  const double *r, const double *mu, const uint64_t *mask, double *nll,
                                  double *kernel_ms) {
    if (!ctx || !C || !w || !r || !mu || !nll || B < 0 || S < 1) {
        theta_set_error("theta_score_masked: bad argument");
        return THETA_ERR_ARG;
    }
    if ((n != 2 && n != 3) || m < 1 || m > 256) {
        theta_set_error("theta_score_masked: need n in {2,3}, 1 <= m <= 256");
        return THETA_ERR_ARG;
    }
    if (B == 0) return THETA_OK;
    HIP_ENTER(ctx->device);
    hipStream_t st = ctx->stream;
    int words = (m + 63) / 64;
    DevBuf d_C, d_w, d_r, d_mu, d_mask, d_nll, d_rsum;
    int rc;
    if ((rc = upload(d_C, C, (size_t)B * m * (n - 1), st))) return rc;
    if ((rc = upload(d_w, w, (size_t)m * sizeof(double), st))) return rc;
    if ((rc = upload(d_r, r, (size_t)m * sizeof(double), st))) return rc;
    if ((rc = upload(d_mu, mu, (size_t)B * n * sizeof(double), st))) return rc;
    if (mask && (rc = upload(d_mask, mask, (size_t)S * words * sizeof(uint64_t), st))) return rc;
    if ((rc = d_nll.alloc((size_t)B * S * sizeof(double)))) return rc;
    if ((rc = d_rsum.alloc((size_t)S * sizeof(double)))) return rc;
    double rlogw = 0.0;
    const bool rlogw_ok = host_rlogw(w, r, m, rlogw);
    HIP_TRY(hipEventRecord(ctx->ev0, st));
    batch_launch_score_masked(n, m, tau, B, S, (const unsigned char *)d_C.p, (const double *)d_w.p, (const double *)d_r.p,
                              (const double *)d_mu.p, mask ? (const unsigned long long *)d_mask.p : nullptr,
                              (double *)d_nll.p, (double *)d_rsum.p, st, host_sum(r, m), true, rlogw, rlogw_ok, host_sum(w, m), host_min(w, m), host_max(w, m));
    HIP_TRY(hipEventRecord(ctx->ev1, st));
    HIP_TRY(hipMemcpyAsync(nll, d_nll.p, (size_t)B * S * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    if (kernel_ms) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
        *kernel_ms = ms;
    }
    return THETA_OK;
}
