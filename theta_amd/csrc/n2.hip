// n = 2 fused search for gfx950: rank -> candidate unranking, colex successor on run
// break-points, group-aggregated root solve of dL/dnu, NLL, running minimum + tie list.
//
// Reference operators replaced (file:line into the reference's python/):
//   Enumerator._generate_next_C_2 / _C_to_array      Enumerator.py:119-160
//   Optimizer._solve_n2, dL_dMu, M2, M2_Rev, L2       Optimizer.py:90-126, 187-231
//   the running minimum of do_optimization_single     RunTHetA.py:191-208
//
// A candidate is a non-decreasing column c_0..c_{m-1}; intervals with the same copy number v form
// one contiguous run [s_v, s_{v+1}).  Every likelihood sum of the reference factorises over these
// runs: with R_v = sum r_i and N_v = sum rN_i over the run (exact integers from prefix sums),
//   dL/dnu(x) = -sum_v R_v (sigma - v) / (v + x (sigma - v)),     sigma = sum_v v N_v / sum rN
//   NLL       = K0 - sum_v R_v ln(tau mu + v (1-mu)) + Rtot ln(tau mu + sigma (1-mu))
// so one candidate costs O(k) instead of O(m) and touches no HBM.
#include <stdlib.h>

#include "n2.hpp"

// ------------------------------------------------------------------------------------------------
// host: order-adjusted bounds + cumulative counts
// ------------------------------------------------------------------------------------------------
int n2_build_host(int m, const int32_t *lb_in, const int32_t *ub_in, N2Host &h) {
    h.m = m;
    h.lb.assign(lb_in, lb_in + m);
    h.ub.assign(ub_in, ub_in + m);
    for (int i = 1; i < m; i++)
        if (h.lb[i] < h.lb[i - 1]) h.lb[i] = h.lb[i - 1];
    for (int i = m - 2; i >= 0; i--)
        if (h.ub[i] > h.ub[i + 1]) h.ub[i] = h.ub[i + 1];
    int top = 0;
    for (int i = 0; i < m; i++) {
        if (h.lb[i] < 0 || h.ub[i] > THETA_MAX_COPY) {
            theta_set_error("copy-number bounds must lie in [0, %d] (interval %d: [%d, %d])", THETA_MAX_COPY, i,
                            h.lb[i], h.ub[i]);
            return THETA_ERR_ARG;
        }
        if (h.ub[i] > top) top = h.ub[i];
    }
    h.kv = top + 1;
    h.P.assign((size_t)m * N2_KVS, 0);
    // W = number of admissible prefixes ending in value v; P = its running sum over v.
    unsigned long long W[N2_KVS], Wn[N2_KVS];
    for (int v = 0; v < N2_KVS; v++) W[v] = (v >= h.lb[0] && v <= h.ub[0]) ? 1 : 0;
    for (int i = 0; i < m; i++) {
        if (i > 0) {
            unsigned long long run = 0;
            for (int v = 0; v < N2_KVS; v++) {
                unsigned long long nr = run + W[v];
                if (nr < run) {
                    theta_set_error("n=2 candidate count exceeds 64 bits");
                    return THETA_ERR_OVERFLOW;
                }
                run = nr;
                Wn[v] = (v >= h.lb[i] && v <= h.ub[i]) ? run : 0;
            }
            memcpy(W, Wn, sizeof(W));
        }
        unsigned long long cum = 0;
        for (int v = 0; v < N2_KVS; v++) {
            unsigned long long nc = cum + W[v];
            if (nc < cum) {
                theta_set_error("n=2 candidate count exceeds 64 bits");
                return THETA_ERR_OVERFLOW;
            }
            cum = nc;
            h.P[(size_t)i * N2_KVS + v] = cum;
        }
    }
    h.total = h.P[(size_t)(m - 1) * N2_KVS + (N2_KVS - 1)];
    return THETA_OK;
}

// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
#include "n2_cand.hpp"

struct N2Result {
    bool ok, degenerate, exact;
    double mu, nll, x;
    int iters, terms;
};

// Group-aggregated restatement of Optimizer._solve_n2 (Optimizer.py:90-126).
//   warm      root of the previous accepted candidate of this thread (NaN: none) -- neighbours in the colex
//             order differ in a few rows only, so it is an excellent Newton start
//   screen    NLL above which the exact (FP64 log) value is not needed; the f32 value is returned then
template <int KV, bool EXACT>
__device__ N2Result n2_solve(const N2Dev &P, const double *PRl, const double *PNl, const N2Cand<KV> &c, double warm,
                             double screen) {
    N2Result out;
    out.ok = false;
    out.degenerate = false;
    out.exact = true;
    out.mu = 0;
    out.nll = 0;
    out.x = warm;
    out.iters = 0;
    double R[KV], w[KV];
    double S1 = 0;
    int ng = 0;
#pragma unroll
    for (int v = 0; v < KV; v++) {
        int a = c.s[v], b = c.s[v + 1];
        double Nv = PNl[b] - PNl[a];
        R[v] = PRl[b] - PRl[a];
        S1 += (double)v * Nv;
        ng += (b > a);
    }
    out.terms = ng;
    if (S1 == 0.0) {  // all-zero tumour column: the reference's Chat is NaN -> brenth raises -> None (quirk Q4)
        out.degenerate = true;
        return out;
    }
    // An interval with c_i = 0 AND r_i = 0 makes the reference's dL_dMu 0/0 = NaN at the left end nu = 0 (the term
    // r_i (a_i - 0) / (a_i nu), Optimizer.py:208-221): brenth raises on the NaN and the candidate is None, whatever the
    // other intervals of the run hold (inf + NaN = NaN).  The run of zeros is the prefix [0, s[1]).
    if (c.s[1] > P.first_zero_r) return out;
    const double sigma = S1 / P.N;
    const double tau = (double)P.tau;
#pragma unroll
    for (int v = 0; v < KV; v++) w[v] = sigma - (double)v;

    // bracket [lo, hi] in nu-space; hi = M2_Rev(max_normal) (Optimizer.py:107-110, 228-231)
    const double lo = 0.0;
    double hi = 1.0;
    if (P.max_normal != 1.0) hi = P.max_normal * tau / ((1.0 - P.max_normal) * sigma + P.max_normal * tau);

    // f = dL/dnu is increasing (the NLL is convex in nu); exact divisions at the ends so that a pole gives +-inf
    auto fend = [&](double x) {
        double f = 0.0;
#pragma unroll
        for (int v = 0; v < KV; v++)
            if (R[v] != 0.0) f = __builtin_fma(-R[v], w[v] / __builtin_fma(x, w[v], (double)v), f);
        return f;
    };
    const double flo = (R[0] != 0.0) ? -__builtin_inf() : fend(lo);   // a run of zeros puts a pole at nu = 0
    const double fhi = fend(hi);
    double x;
    if (flo == 0.0) {
        x = lo;  // brenth returns the left end when f(lo) == 0 (proportional columns, quirk Q3)
    } else if (fhi == 0.0) {
        x = hi;
    } else if (flo != flo || fhi != fhi || (flo < 0) == (fhi < 0)) {
        return out;  // no sign change on [lo, hi]: brenth raises -> None (quirk Q6)
    } else {
        // Damped Newton on the self-concordant NLL/Rtot, safeguarded by the bracket f(a) < 0 < f(b).
        double a = lo, b = hi;
        x = (warm > lo && warm < hi) ? warm : 0.5 * (a + b);
        const double inv_R = 1.0 / P.Rtot;
        for (int it = 0; it < 100; it++) {
            out.iters++;
            double f = 0.0, d = 0.0;
#pragma unroll
            for (int v = 0; v < KV; v++) {
                if (R[v] != 0.0) {
                    double t = w[v] * rcp_nr1(__builtin_fma(x, w[v], (double)v));
                    f = __builtin_fma(-R[v], t, f);
                    d = __builtin_fma(R[v] * t, t, d);
                }
            }
            if (f == 0.0) break;
            if (f < 0) a = x; else b = x;
            const double dx = f * rcp_nr2(d);
            const double l2 = f * dx * inv_R;             // squared Newton decrement
            double step = 1.0;
            if (l2 > 0.09) step = 1.0 / (1.0 + sqrt(l2));
            double xn = __builtin_fma(-step, dx, x);
            // (a step below one ulp of x leaves xn == x == a or b: that is the root to machine precision, not a reason to bisect --
            // with the strict test a start AT the root jumped to the middle of the bracket and, the decrement being tiny, stayed there)
            if (!(xn >= a && xn <= b)) xn = 0.5 * (a + b);
            x = xn;
            if (l2 < 1e-12 || b - a <= 1e-300) break;     // the step just taken leaves an error ~ l2
        }
    }
    // nu -> mu (M2, Optimizer.py:223-226) and the NLL (L2, Optimizer.py:187-196) in group form
    const double mu = x * sigma / ((1.0 - x) * tau + x * sigma);
    const double nu1 = 1.0 - mu;
    const double dall = __builtin_fma(sigma, nu1, tau * mu);
    if (!EXACT) {   // single-precision screen first: most candidates are far above the running minimum
        float acc = 0.0f;
        const float fnu1 = (float)nu1, ftm = (float)(tau * mu);
#pragma unroll
        for (int v = 0; v < KV; v++)
            if (R[v] != 0.0) acc = __builtin_fmaf((float)R[v], __logf(__builtin_fmaf((float)v, fnu1, ftm)), acc);
        double approx = P.K0 - (double)acc + P.Rtot * (double)__logf((float)dall);
        if (approx > screen) {
            out.nll = approx;
            out.exact = false;
            out.mu = mu;
            out.x = x;
            out.ok = true;
            return out;
        }
    }
    double acc = 0.0;
#pragma unroll
    for (int v = 0; v < KV; v++)
        if (R[v] != 0.0) acc = __builtin_fma(R[v], log(__builtin_fma((double)v, nu1, tau * mu)), acc);
    out.nll = P.K0 - acc + P.Rtot * log(dall);
    out.mu = mu;
    out.x = x;
    out.ok = true;
    return out;
}


// ---- dismissal by a lower bound (the search's fast path; the n=3 sieve's idea in one dimension) --------------------------
// In nu-space the candidate's likelihood is  NLL(x) = K0 + Rtot ln sigma - sum_v R_v ln q_v,  q_v = v + x (sigma - v): an R-weighted
// log barrier, self-concordant with parameter 2 / sqrt(Rmin).  ONE evaluation of value, slope f and curvature d at the thread's
// chain point x (the Newton-stepped point of its previous candidate) gives, with lambda^2 = f^2 / d and t = lambda / sqrt(Rmin) < 1/2,
//      min_x NLL >= NLL(x) - (lambda^2 / 2) (1 + t + 2 t^2)
// -- a bound on ANYTHING the reference can report for the candidate (its root lies in the same domain; a candidate it rejects
// reports nothing).  A candidate whose bound lies beyond the window of the running minimum is done: no bracket test (16 exact
// divisions), no iteration to 1e-12, no exact logarithms.  What is undecided -- near-ties, a chain point that lost its
// candidate at a carry of the successor -- goes to a per-wave queue in LDS and takes the full path (n2_solve, which alone decides
// acceptance and lists finalists) 64 at a time: the two paths never share a wave instruction stream half empty.  Branch-free
// over the KV runs (an empty run has R = 0 and contributes nothing).  The value is a single-precision screen behind the margin
// the full path's screen uses (2e-5 Rtot + 1).
template <int KV>
__device__ __forceinline__ bool n2_quick(const N2Dev &P, double inv_N, const double2 *PRNl, const N2Cand<KV> &c, double &warm, double thr) {
    double R[KV];
    double S1 = 0.0;
    float Rmin = __builtin_inff();
    {
        double2 pv[KV + 1];                                            // {sum r, sum rN} below every break-point: one 16-byte read each,
#pragma unroll
        for (int v = 0; v <= KV; v++) pv[v] = PRNl[c.s[v]];            // all in flight before the first is used
#pragma unroll
        for (int v = 0; v < KV; v++) {
            R[v] = pv[v + 1].x - pv[v].x;
            S1 = __builtin_fma((double)v, pv[v + 1].y - pv[v].y, S1);
            const float rf = (float)R[v];
            Rmin = fminf(Rmin, rf > 0.0f ? rf : __builtin_inff());
        }
    }
    if (!(S1 > 0.0) || !(Rmin < __builtin_inff())) return false;       // all-zero column / no reads at all: the full path's business
    const double sigma = S1 * inv_N;
    const double x = (warm > 1e-30 && warm < 1.0) ? warm : 0.5;
    double f = 0.0, d = 0.0;
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < KV; v++) {
        const double w = sigma - (double)v;
        const double q = __builtin_fma(x, w, (double)v);               // > 0 for x in (0, 1)
        const float qf = (float)q;
        double u = (double)__builtin_amdgcn_rcpf(qf);
        u = __builtin_fma(u, __builtin_fma(-q, u, 1.0), u);            // 1 / q to ~3e-14
        const double t = w * u;
        f = __builtin_fma(-R[v], t, f);
        d = __builtin_fma(R[v] * t, t, d);
        acc = __builtin_fmaf((float)R[v], __builtin_amdgcn_logf(qf), acc);
    }
    const double dx = f * rcp_nr1(d);
    const double l2 = f * dx;                                          // lambda^2 (0/0 = NaN for proportional columns: undecided)
    const double t = (double)(__builtin_amdgcn_sqrtf((float)l2) * __builtin_amdgcn_rsqf(Rmin)) * 1.0001;
    const double gap = 0.525 * l2 * __builtin_fma(t, __builtin_fma(2.0, t, 1.0), 1.0);
    const double lower = P.K0 + 0.6931471805599453 * (P.Rtot * (double)__builtin_amdgcn_logf((float)sigma) - (double)acc) - gap;
    // the Newton-stepped point (damped while the decrement is large), kept inside (0, 1): where the next candidate of this
    // neighbourhood is evaluated, and where the full path starts if this one is undecided
    double xn = x - dx * (t > 0.3 ? rcp_nr1(1.0 + t) : 1.0);
    if (!(xn > 0.0 && xn < 1.0)) xn = 0.5 * (x + (dx > 0.0 ? 0.0 : 1.0));
    warm = xn == xn ? xn : x;
    return t < 0.5 && lower > thr;
}

#define N2_QCAP 96        // undecided candidates a wave holds (it runs the full path on 64 of them as soon as it has that many)
// LDS of the dismissing search behind the tables every n=2 kernel stages: {sum r, sum rN} interleaved [m + 1] (one 16-byte read
// per break-point), the chain points per successor level [kv][256], the per-wave queues [4][N2_QCAP][4 + (KV + 2) / 2 words]
__host__ __device__ inline size_t n2_prn_offset(int m) {
    return (((size_t)(m + 1) * 16 + (size_t)m * N2_KVS * 8 + (N2_KVS + 1 + 3) * 2 + (size_t)m + 16) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t n2_xs_offset(int m) { return n2_prn_offset(m) + (size_t)(m + 1) * 16; }
__host__ __device__ inline size_t n2_queue_offset(int m, int kv) { return n2_xs_offset(m) + (size_t)kv * 256 * 8; }     // (chain points and their slopes: floats)

template <int KV, bool DUMP, bool QUICK = false>
__global__ __launch_bounds__(256, (QUICK && KV <= 8) ? 3 : 1) void n2_search_kernel(N2Dev P, SearchArgs A, unsigned long long begin,
                                                        unsigned long long end, int per_thread, unsigned long long sample_stride) {
    extern __shared__ unsigned char smem[];
    // LDS staging of everything the candidates share
    double *PRl = (double *)smem;
    double *PNl = PRl + (P.m + 1);
    unsigned long long *Pl = (unsigned long long *)(PNl + (P.m + 1));
    short *lbposl = (short *)(Pl + (size_t)P.m * N2_KVS);
    unsigned char *ubl = (unsigned char *)(lbposl + (N2_KVS + 1) + 3);
    for (int i = threadIdx.x; i <= P.m; i += blockDim.x) {
        PRl[i] = P.PR[i];
        PNl[i] = P.PN[i];
    }
    for (int i = threadIdx.x; i < P.m * N2_KVS; i += blockDim.x) Pl[i] = P.P[i];
    for (int i = threadIdx.x; i <= N2_KVS; i += blockDim.x) lbposl[i] = P.lbpos[i];
    for (int i = threadIdx.x; i < P.m; i += blockDim.x) ubl[i] = P.ub[i];
    if (QUICK)
        for (int i = threadIdx.x; i <= P.m; i += blockDim.x) ((double2 *)(smem + n2_prn_offset(P.m)))[i] = make_double2(P.PR[i], P.PN[i]);
    __syncthreads();

    unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (sample_stride) {
        // SAMPLE launch of a search that starts without a minimum: every thread solves ONE candidate (ranks `sample_stride` apart)
        // and the smallest NLL the reference's procedure reports among them becomes the running minimum -- nothing else is written.
        // Without it every thread's first candidate is "within the window" of +inf: a million appends to the tie list and as
        // many atomicMin on one address, 1 ms of a 3 ms search (PMC / timing in profiles/r4/NOTES.md).
        const unsigned long long rank = begin + tid * sample_stride;
        double v = __builtin_inf();
        if (rank < end) {
            N2Cand<KV> c;
            n2_unrank<KV>(P, Pl, rank, c);
            const N2Result rs = n2_solve<KV, true>(P, PRl, PNl, c, __builtin_nan(""), __builtin_inf());
            if (rs.ok) v = rs.nll;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
        if (lane_id() == 0 && v < __builtin_inf()) atomicMin(&A.ctr->best_bits, order_bits(v));
        return;
    }
    unsigned long long t0 = begin + tid * (unsigned long long)per_thread;
    unsigned long long n_eval = 0, n_acc = 0, n_deg = 0, n_it = 0, n_terms = 0, n_fin = 0, n_dis = 0;
    // (QUICK: an instantiation of its own -- the dismissing search does not carry the other loop's registers)
    const double inv_N = 1.0 / P.N;
    // f32 screen: |error| <= ~1e-6 * Rtot * ln(range) -- a margin of 2e-5 Rtot is far outside it
    const double margin = 2e-5 * P.Rtot + 1.0;
    double best = order_unbits(load_agent_u64(&A.ctr->best_bits));
    // one candidate through Optimizer._solve_n2 as restated above: acceptance, root, NLL, tie list, running minimum
    auto full = [&](const N2Cand<KV> &cc, unsigned long long rank, double &wm) {
        N2Result rs = n2_solve<KV, DUMP>(P, PRl, PNl, cc, wm, best + A.window + margin);
        if (rs.ok) wm = rs.x;
        n_it += rs.iters;
        n_terms += (unsigned long long)rs.iters * rs.terms;
        n_deg += rs.degenerate;
        if (rs.ok) {
            n_acc++;
            n_fin += rs.terms + 1;
            if (rs.exact && rs.nll <= best + A.window) {
                // refresh: another thread may have lowered the global minimum meanwhile
                best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
                if (rs.nll <= best + A.window) {
                    tie_append(A.ctr, A.list, A.list_cap, (u128)rank, rs.nll, rs.mu, 1.0 - rs.mu, 0.0);
                    if (rs.nll < best) atomicMin(&A.ctr->best_bits, order_bits(rs.nll));
                }
            }
            if (rs.nll < best) best = rs.nll;        // (the thread's own minimum is an attained value too)
        }
        if (DUMP) {
            A.dump_nll[rank - begin] = rs.ok ? rs.nll : __builtin_nan("");
            A.dump_mu[(rank - begin) * 2] = rs.ok ? rs.mu : __builtin_nan("");
            A.dump_mu[(rank - begin) * 2 + 1] = rs.ok ? 1.0 - rs.mu : __builtin_nan("");
        }
    };
    if constexpr (!QUICK) {
        if (t0 < end) {
            unsigned long long t1 = t0 + (unsigned long long)per_thread;
            if (t1 > end) t1 = end;
            N2Cand<KV> c;
            n2_unrank<KV>(P, Pl, t0, c);
            double warm = __builtin_nan("");
            for (unsigned long long rank = t0; rank < t1; rank++) {
                full(c, rank, warm);
                n_eval++;
                if (rank + 1 < t1 && !n2_next<KV>(P, ubl, lbposl, c)) break;
            }
        }
    } else {
        // Every lane of the wave runs the same number of trips (`on` says whether it still has a candidate), so that all 64 lanes
        // are there when the queue of undecided candidates is drained.  Queue entry: rank, chain point, break-points (16 bits each).
        constexpr int EW = 4 + (KV + 2) / 2;
        unsigned *queue = (unsigned *)(smem + n2_queue_offset(P.m, KV)) + (threadIdx.x >> 6) * (N2_QCAP * EW);
        const double2 *PRNl = (const double2 *)(smem + n2_prn_offset(P.m));
        // Chain points: xs[w] = the stepped point of the candidate that followed the last successor step of level >= w.  A step of
        // level nv moves break-points 1 .. nv to one position and leaves the rest: the candidate it makes differs by ONE interval
        // from the one the previous step of that level made -- a far better start than the previous candidate, which a carry
        // (nv > 1) leaves behind by a whole run of intervals.
        float *xs = (float *)(smem + n2_xs_offset(P.m)) + threadIdx.x;             // xs[w * 256] (single precision: a start, nothing more)
        // ... and how far the point moved between the last two steps of that level (consecutive steps of one level raise
        // neighbouring positions by the same amount: the next optimum lies about as far on) -- xs + slope is where a step starts
        float *xsl = (float *)(smem + n2_xs_offset(P.m) + (size_t)KV * 256 * 4) + threadIdx.x;
        const int lane = threadIdx.x & 63;
        int qcount = 0;                                                  // (wave-uniform)
        auto drain = [&](int keep) {                                     // full path, 64 entries at a time, until <= keep are left
            while (qcount > keep) {
                const int take = qcount < WAVE ? qcount : WAVE;
                const bool has = lane < take;
                const unsigned *e = queue + (qcount - take + (has ? lane : 0)) * EW;
                N2Cand<KV> cc;
#pragma unroll
                for (int v = 0; v <= KV; v++) cc.s[v] = (int)((e[4 + (v >> 1)] >> (16 * (v & 1))) & 0xffffu);
                double wm = __longlong_as_double(((long long)e[3] << 32) | e[2]);
                // most of the queue is candidates whose chain point was a step behind (a carry of the successor, a tiny run): two
                // more evaluations, each from the Newton-stepped point of the last, dismiss them; what is left -- the near-ties --
                // is what the full path is for
                bool und = has;
#pragma unroll 1
                for (int k = 0; k < ((P.quick & 8) ? 0 : 2); k++) {
                    if (!ballot64(und)) break;
                    const bool dis = n2_quick<KV>(P, inv_N, PRNl, cc, wm, best + A.window + margin);
                    if (und) {
                        n_it++;
                        n_terms += KV;
                        n_fin += KV + 1;
                        n_dis += dis;
                    }
                    und = und && !dis;
                }
                if (und) full(cc, ((unsigned long long)e[1] << 32) | e[0], wm);
                qcount -= take;
                wave_lds_sync();
            }
        };
        unsigned long long t1 = t0 + (unsigned long long)per_thread;
        if (t1 > end) t1 = end;
        bool on = t0 < end;
        N2Cand<KV> c;
        if (on) n2_unrank<KV>(P, Pl, t0, c);
        else {
#pragma unroll
            for (int v = 0; v <= KV; v++) c.s[v] = v ? P.m : 0;          // (a harmless candidate for idle lanes)
        }
        double warm = __builtin_nan("");
        {   // the thread's first candidate: a few Newton steps from the centre settle the chain point of every level
#pragma unroll 1
            for (int k = 0; k < 4; k++) (void)n2_quick<KV>(P, inv_N, PRNl, c, warm, __builtin_inf());
#pragma unroll
            for (int w = 1; w < KV; w++) {
                xs[w * 256] = (float)warm;
                xsl[w * 256] = 0.0f;
            }
            if (on) {
                n_it += 4;
                n_terms += 4 * KV;
            }
        }
        int nv = 1;
        unsigned n_q = 0;                                                // (32 bits: a thread walks <= 512 candidates)
        // bound tables of the successor, wave-uniform (scalar registers): lbp[w] / ubp[w] = first position whose lower / upper
        // bound is >= w
        int lbp[KV + 1], ubp[KV + 1];
#pragma unroll
        for (int w = 0; w <= KV; w++) {
            lbp[w] = (int)P.lbpos[w];
            int lo = 0, hi = P.m;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if ((int)P.ub[mid] >= w) hi = mid;
                else lo = mid + 1;
            }
            ubp[w] = lo;
        }
        for (int it = 0; it < per_thread; it++) {
            if (!ballot64(on)) break;
            const unsigned long long rank = t0 + (unsigned long long)it;
            // what the other threads found meanwhile: often at the start (a search without a hint begins with no minimum at all)
            if (!(P.quick & 2) && ((it & 31) == 31 || (it & (it - 1)) == 0)) best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
            const double xprev = (double)xs[nv * 256];
            warm = xprev + ((P.quick & 4) ? 0.0 : (double)xsl[nv * 256]);
            if (!(warm > 1e-30 && warm < 1.0)) warm = xprev;
            const bool dis = n2_quick<KV>(P, inv_N, PRNl, c, warm, best + A.window + margin);
            for (int w = 1; w < KV; w++) {                               // (wave-uniform trip count: the largest level of the wave)
                if (!ballot64(w <= nv)) break;
                if (w <= nv) {
                    xs[w * 256] = (float)warm;
                    xsl[w * 256] = w == nv ? (float)(warm - xprev) : 0.0f;
                }
            }
            if (on) {
                n_q++;
                n_dis += dis;
            }
            const bool push = on && !dis;
            const unsigned long long pm = ballot64(push);
            if (pm) {
                if (qcount + __builtin_popcountll(pm) > N2_QCAP) drain(0);
                if (push) {
                    unsigned *e = queue + (qcount + mbcnt(pm)) * EW;
                    e[0] = (unsigned)rank;
                    e[1] = (unsigned)(rank >> 32);
                    const unsigned long long xb = (unsigned long long)__double_as_longlong(warm);
                    e[2] = (unsigned)xb;
                    e[3] = (unsigned)(xb >> 32);
#pragma unroll
                    for (int v = 0; v <= KV; v += 2) e[4 + (v >> 1)] = (unsigned)c.s[v] | (v + 1 <= KV ? (unsigned)c.s[v + 1] << 16 : 0u);
                }
                qcount += __builtin_popcountll(pm);
                wave_lds_sync();
                if (qcount >= WAVE) drain(WAVE - 1);
            }
            if (on) {
                nv = rank + 1 < t1 ? n2_next_tab<KV>(lbp, ubp, c) : 0;
                if (nv == 0) {
                    on = false;
                    nv = 1;
                }
            }
        }
        drain(0);
        n_eval += n_q;                                                   // every candidate had its one shared-point evaluation: KV terms each
        n_it += n_q;
        n_terms += (unsigned long long)n_q * KV;
        n_fin += (unsigned long long)n_q * (KV + 1);
    }
    // one atomic per wave per counter
    n_eval = wave_sum_u64(n_eval);
    n_acc = wave_sum_u64(n_acc);
    n_deg = wave_sum_u64(n_deg);
    n_it = wave_sum_u64(n_it);
    n_terms = wave_sum_u64(n_terms);
    n_fin = wave_sum_u64(n_fin);
    n_dis = wave_sum_u64(n_dis);
    if (lane_id() == 0 && n_eval) {
        SearchCounters *sc = stat_slot(A);
        if (n_dis) atomicAdd(&sc->dismissed, n_dis);
        atomicAdd(&sc->evaluated, n_eval);
        if (n_acc) atomicAdd(&sc->accepted, n_acc);
        if (n_deg) atomicAdd(&sc->degenerate, n_deg);
        atomicAdd(&sc->iterations, n_it);
        atomicAdd(&sc->terms, n_terms);                 // (terms64 = terms for n=2: set by the host)
        atomicAdd(&sc->final_terms, n_fin);
    }
}

typedef unsigned n2_u4v __attribute__((ext_vector_type(4)));
typedef n2_u4v N2U4 __attribute__((aligned(4)));   // a 16-byte store that is only 4-byte aligned

// Materialise candidates: thread t writes `per_thread` consecutive candidates starting at begin+t*per_thread.
template <int KV>
__global__ __launch_bounds__(256) void n2_enumerate_kernel(N2Dev P, unsigned long long begin, unsigned long long count,
                                                           int per_thread, unsigned char *out) {
    extern __shared__ unsigned char smem[];
    unsigned long long *Pl = (unsigned long long *)smem;
    short *lbposl = (short *)(Pl + (size_t)P.m * N2_KVS);
    unsigned char *ubl = (unsigned char *)(lbposl + (N2_KVS + 1) + 3);
    for (int i = threadIdx.x; i < P.m * N2_KVS; i += blockDim.x) Pl[i] = P.P[i];
    for (int i = threadIdx.x; i <= N2_KVS; i += blockDim.x) lbposl[i] = P.lbpos[i];
    for (int i = threadIdx.x; i < P.m; i += blockDim.x) ubl[i] = P.ub[i];
    __syncthreads();
    unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long k0 = tid * (unsigned long long)per_thread;
    if (k0 >= count) return;
    unsigned long long k1 = k0 + per_thread;
    if (k1 > count) k1 = count;
    N2Cand<KV> c;
    n2_unrank<KV>(P, Pl, begin + k0, c);
    const bool words = (P.m & 3) == 0 && (((unsigned long long)out) & 3ull) == 0ull;
    const bool stream = !words && (((unsigned long long)(out + k0 * (unsigned long long)P.m)) & 3ull) == 0ull;
    unsigned *s32 = (unsigned *)(out + k0 * (unsigned long long)P.m);
    unsigned carry = 0;
    int phase = 0;
    for (unsigned long long k = k0; k < k1; k++) {
        unsigned char *dst = out + k * (unsigned long long)P.m;
        // value at position i = number of v >= 1 with s[v] <= i
        if (words) {
            // four positions per 32-bit word: break-point v adds 1 to every byte at or after s[v]; 16-byte stores (a record
            // is only 4-byte aligned: the hardware takes unaligned multi-dword stores)
            unsigned *d32 = (unsigned *)dst;
            const int nw = P.m >> 2;
            auto word = [&](int w) -> unsigned {
                unsigned val = 0;
#pragma unroll
                for (int v = 1; v < KV; v++) {
                    const int d = c.s[v] - 4 * w;
                    val += d <= 0 ? 0x01010101u : (d >= 4 ? 0u : (0x01010101u << (8 * d)));
                }
                return val;
            };
            int w = 0;
            for (; w + 4 <= nw; w += 4) *(N2U4 *)(d32 + w) = N2U4{word(w), word(w + 1), word(w + 2), word(w + 3)};
            for (; w < nw; w++) d32[w] = word(w);
        } else if (stream) {
            // m not a multiple of 4: the thread's candidates are one contiguous, 4-byte-aligned run of bytes -- emit it word
            // by word, a word that straddles two records being finished by the next one (`carry` holds its first `phase` bytes)
            int pos = -phase;                                   // record position of the first byte of the current word
            while (pos < P.m) {
                unsigned val = 0;
#pragma unroll
                for (int v = 1; v < KV; v++) {
                    const int d = c.s[v] - pos;
                    val += d <= 0 ? 0x01010101u : (d >= 4 ? 0u : (0x01010101u << (8 * d)));
                }
                if (pos < 0) val = (val & (0xffffffffu << (8 * (-pos)))) | carry;
                const int over = pos + 4 - P.m;
                if (over > 0) {
                    carry = val & (0xffffffffu >> (8 * over));
                    phase = 4 - over;
                    break;
                }
                *s32++ = val;
                carry = 0;
                phase = 0;
                pos += 4;
            }
        } else {
            for (int i = 0; i < P.m; i++) {
                int val = 0;
#pragma unroll
                for (int v = 1; v < KV; v++) val += (c.s[v] <= i);
                dst[i] = (unsigned char)val;
            }
        }
        if (k + 1 < k1 && !n2_next<KV>(P, ubl, lbposl, c)) {
            k1 = k + 1;
            break;
        }
    }
    if (stream && phase > 0) {                                   // the last, partial word of the run
        unsigned char *tail = (unsigned char *)s32;
        for (int i = 0; i < phase; i++) tail[i] = (unsigned char)(carry >> (8 * i));
    }
}

// ------------------------------------------------------------------------------------------------
// The same generator, written through an LDS transposition so that the memory system sees WHOLE CACHE LINES.
// In the kernel above every lane owns one output stream: a store instruction touches 64 different 128-byte lines, 4e5
// streams are open at once, and partial lines leave the L2 before they are complete (0.9-1.5 TB/s).  Here a thread's run is
// T candidates with T m a multiple of 128 bytes (so every run starts on a line boundary); per line each lane builds its
// eight 16-byte chunks into its own row of a per-wave LDS tile, and the wave then stores the tile line by line: one store
// instruction = 8 complete lines (8 lanes x 16 bytes each).  Control flow is wave-uniform (see `pos` below); the bytes of a word
// are the SWAR sum over the break-points, as in the kernel above.
// ------------------------------------------------------------------------------------------------
#define N2L_STRIDE 36      // dwords per LDS row (128 bytes of payload + 16 of padding: spreads the rows over the banks)
template <int KV>
__global__ __launch_bounds__(256) void n2_enumerate_lines_kernel(N2Dev P, unsigned long long begin, unsigned long long count, int T,
                                                                 unsigned char *out) {
    extern __shared__ unsigned char smem[];
    unsigned long long *Pl = (unsigned long long *)smem;
    short *lbposl = (short *)(Pl + (size_t)P.m * N2_KVS);
    unsigned char *ubl = (unsigned char *)(lbposl + (N2_KVS + 1) + 3);
    unsigned *tile = (unsigned *)(smem + (((size_t)P.m * N2_KVS * 8 + (N2_KVS + 1 + 3) * 2 + P.m + 15) & ~(size_t)15)) +
                     (threadIdx.x >> 6) * (WAVE * N2L_STRIDE);
    for (int i = threadIdx.x; i < P.m * N2_KVS; i += blockDim.x) Pl[i] = P.P[i];
    for (int i = threadIdx.x; i <= N2_KVS; i += blockDim.x) lbposl[i] = P.lbpos[i];
    for (int i = threadIdx.x; i < P.m; i += blockDim.x) ubl[i] = P.ub[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, m = P.m;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long wave_first = tid - lane;
    const unsigned long long k0 = tid * (unsigned long long)T;
    const unsigned long long RB = (unsigned long long)T * m;                 // bytes per run, a multiple of 128
    const int lines = (int)(RB >> 7);
    unsigned long long mine = k0 < count ? (count - k0 < (unsigned long long)T ? count - k0 : (unsigned long long)T) : 0;   // candidates of this lane
    N2Cand<KV> c;
    if (mine) n2_unrank<KV>(P, Pl, begin + k0, c);
    else {
#pragma unroll
        for (int v = 0; v <= KV; v++) c.s[v] = m;
    }
    auto word_at = [&](int pos) -> unsigned {                                // bytes pos .. pos+3 of the current record
        unsigned val = 0;
#pragma unroll
        for (int v = 1; v < KV; v++) {
            const int d = c.s[v] - pos;
            val += d <= 0 ? 0x01010101u : (d >= 4 ? 0u : (0x01010101u << (8 * d)));
        }
        return val;
    };
    // Every lane's run starts on a record boundary AND on a line boundary, and all runs have the same length: the position
    // inside the current record is the same for all 64 lanes at every word -- `pos` is wave-uniform, so is every branch on it
    // (only the break-points differ from lane to lane).  A lane whose run is shorter (the last one) emits zeros, never stored.
    int pos = 0;
    unsigned long long left = mine;                                          // records still to finish, the current one included
    auto advance = [&]() {                                                   // next record of this lane's run
        if (left > 0) {
            left--;
            if (left > 0 && !n2_next<KV>(P, ubl, lbposl, c)) left = 0;
        }
    };
    auto next_word = [&]() -> unsigned {
        unsigned val = left > 0 ? word_at(pos) : 0u;
        const int over = pos + 4 - m;                                        // (uniform)
        if (over > 0) {                                                      // the word straddles two records
            val &= 0xffffffffu >> (8 * over);
            advance();
            if (left > 0) val |= word_at(0) << (8 * (4 - over));
            pos = over;
        } else {
            pos += 4;
            if (pos == m) {
                advance();
                pos = 0;
            }
        }
        return val;
    };
    unsigned *row = tile + lane * N2L_STRIDE;
    for (int line = 0; line < lines; line++) {
#pragma unroll 2
        for (int ch = 0; ch < 8; ch++) {
            n2_u4v v;
            v.x = next_word();
            v.y = next_word();
            v.z = next_word();
            v.w = next_word();
            *(n2_u4v *)(row + 4 * ch) = v;
        }
        wave_lds_sync();
#pragma unroll
        for (int sidx = 0; sidx < 8; sidx++) {
            const int r = (lane >> 3) + 8 * sidx, ch = lane & 7;
            const unsigned long long tt = wave_first + (unsigned long long)r;
            const unsigned long long kk = tt * (unsigned long long)T;
            if (kk < count) {
                const unsigned long long nv = (count - kk < (unsigned long long)T ? count - kk : (unsigned long long)T) * (unsigned long long)m;
                const unsigned long long off = ((unsigned long long)line << 7) + (unsigned long long)ch * 16;
                if (off < nv) {
                    const n2_u4v v = *(const n2_u4v *)(tile + r * N2L_STRIDE + 4 * ch);
                    unsigned char *dst = out + tt * RB + off;
                    if (off + 16 <= nv) {
                        *(n2_u4v *)dst = v;
                    } else {                                                 // the tail of the very last record
                        const unsigned w4[4] = {v.x, v.y, v.z, v.w};
                        for (int bidx = 0; bidx < (int)(nv - off); bidx++) dst[bidx] = (unsigned char)(w4[bidx >> 2] >> (8 * (bidx & 3)));
                    }
                }
            }
        }
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// The whole-line generator with the records RENDERED by scatter + prefix sum (n2_render.hpp) instead of summed break-point by
// break-point into every word.  Same run / tile / store scheme as n2_enumerate_lines_kernel.  The default since round 3
// (THETA_N2_ENUM_RENDER=0 selects the summing writer): 4.2 / 4.7 TB/s at m = 50 / 100 against 1.5 / 2.1 (word-major LDS tile, n2_render.hpp).  Its per-lane logic is
// also verified on the CPU -- tools/n2_render_emul.hip runs this very code lane by lane against the oracle's enumeration
// (tests/test_n2_render_cpu.py) -- and on the device against the other two generators (tests/test_gpu_zzz_render.py).
// ------------------------------------------------------------------------------------------------
#include "n2_render.hpp"
template <int KV>
__global__ __launch_bounds__(256) void n2_enumerate_render_kernel(N2Dev P, unsigned long long begin, unsigned long long count, int T,
                                                                  unsigned char *out) {
    extern __shared__ unsigned char smem[];
    unsigned long long *Pl = (unsigned long long *)smem;
    short *lbposl = (short *)(Pl + (size_t)P.m * N2_KVS);
    unsigned char *ubl = (unsigned char *)(lbposl + (N2_KVS + 1) + 3);
    unsigned *tile = (unsigned *)(smem + (((size_t)P.m * N2_KVS * 8 + (N2_KVS + 1 + 3) * 2 + P.m + 15) & ~(size_t)15)) +
                     (threadIdx.x >> 6) * N2R_TILE_DWORDS;
    for (int i = threadIdx.x; i < P.m * N2_KVS; i += blockDim.x) Pl[i] = P.P[i];
    for (int i = threadIdx.x; i <= N2_KVS; i += blockDim.x) lbposl[i] = P.lbpos[i];
    for (int i = threadIdx.x; i < P.m; i += blockDim.x) ubl[i] = P.ub[i];
    __syncthreads();
    short *ubposl = (short *)(tile - (threadIdx.x >> 6) * N2R_TILE_DWORDS + 4 * N2R_TILE_DWORDS);   // behind the four tiles
    if (threadIdx.x <= KV) ubposl[threadIdx.x] = n2r_ubpos(ubl, P.m, (int)threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, m = P.m;
    const unsigned long long tid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long wave_first = tid - lane;
    const unsigned long long k0 = tid * (unsigned long long)T;
    const int lines = (int)(((unsigned long long)T * m) >> 7);
    const unsigned long long mine = k0 < count ? (count - k0 < (unsigned long long)T ? count - k0 : (unsigned long long)T) : 0;
    N2Run<KV> R;
    n2r_begin<KV>(R, mine);
    if (mine) n2_unrank<KV>(P, Pl, begin + k0, R.c);
    else {
#pragma unroll
        for (int v = 0; v <= KV; v++) R.c.s[v] = m;
    }
    N2RStore S;
    n2r_store_prepare(lane, wave_first, T, m, count, out, S);
    for (int line = 0; line < lines; line++) {
        n2r_scatter_line<KV>(m, lbposl, ubposl, R, tile, lane);
        n2r_prefix_line(tile, lane);
        wave_lds_sync();
        n2r_store_line(lane, line, S, tile);
        wave_lds_sync();
    }
}

// Candidates for an explicit list of ranks (tie-list materialisation).
template <int KV>
__global__ __launch_bounds__(64) void n2_unrank_list_kernel(N2Dev P, const TieRecord *recs, int count, unsigned char *out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    N2Cand<KV> c;
    n2_unrank<KV>(P, P.P, recs[k].rank_lo, c);
    unsigned char *dst = out + (size_t)k * P.m;
    for (int i = 0; i < P.m; i++) {
        int val = 0;
#pragma unroll
        for (int v = 1; v < KV; v++) val += (c.s[v] <= i);
        dst[i] = (unsigned char)val;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static size_t n2_smem_bytes(const N2Dev &P) {
    return (size_t)(P.m + 1) * 16 + (size_t)P.m * N2_KVS * 8 + (N2_KVS + 1 + 3) * 2 + P.m + 16;
}

void n2_launch_search(const N2Dev &P, const SearchArgs &A, unsigned long long begin, unsigned long long end,
                      int per_thread, hipStream_t st, unsigned long long sample_stride) {
    unsigned long long n = end - begin;
    unsigned long long threads = sample_stride ? (n + sample_stride - 1) / sample_stride : (n + per_thread - 1) / per_thread;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    const bool dump = A.dump_nll != nullptr;
    bool quick = !dump && P.quick != 0;             // the dismissing search: an instantiation of its own (QUICK), with its per-wave queues,
    const int kvt = P.kv <= 8 ? 8 : 16;             // chain points and interleaved prefix sums in LDS behind the common tables
    size_t sm = quick ? n2_queue_offset(P.m, kvt) + (size_t)4 * N2_QCAP * (4 + (kvt + 2) / 2) * 4 : n2_smem_bytes(P);
    if (quick && sm > 48 * 1024) {
        // The dismissing kernel's LDS grows with m and the copy numbers (161 m + ... bytes: beyond 64 KB for m > ~218 at KV = 8, m > ~78
        // at KV = 16).  gfx950 grants 160 KB per workgroup; a device that refuses the size runs the kernel that solves every candidate
        // (37 KB at m = 256) instead -- same finalists -- rather than failing the launch (round-4 advice).
        const hipError_t e = P.kv <= 8 ? hipFuncSetAttribute((const void *)n2_search_kernel<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)
                                       : hipFuncSetAttribute((const void *)n2_search_kernel<16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            quick = false;
            sm = n2_smem_bytes(P);
        }
    }
#define N2_LAUNCH(KVV, DD, QQ)                                                                                                                    \
    do {                                                                                                                                          \
        if (sm > 48 * 1024) (void)hipFuncSetAttribute((const void *)n2_search_kernel<KVV, DD, QQ>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); \
        hipLaunchKernelGGL((n2_search_kernel<KVV, DD, QQ>), dim3(blocks), dim3(256), sm, st, P, A, begin, end, per_thread, sample_stride);       \
    } while (0)
    if (P.kv <= 8) {
        if (dump) N2_LAUNCH(8, true, false);
        else if (quick) N2_LAUNCH(8, false, true);
        else N2_LAUNCH(8, false, false);
    } else {
        if (dump) N2_LAUNCH(16, true, false);
        else if (quick) N2_LAUNCH(16, false, true);
        else N2_LAUNCH(16, false, false);
    }
#undef N2_LAUNCH
}

void n2_launch_enumerate(const N2Dev &P, unsigned long long begin, unsigned long long count, unsigned char *out,
                         hipStream_t st) {
    // whole-line writer: runs of T candidates with T m a multiple of 128 bytes, output 128-byte aligned, enough work to fill
    // the chip (THETA_N2_ENUM_LEGACY=1 keeps the one-stream-per-lane kernel: second implementation for the tests)
    {
        int g = 128, a = P.m;
        while (a) { const int t = g % a; g = a; a = t; }      // gcd(m, 128)
        int T = 128 / g;
        while (T < 32) T *= 2;
        if (!getenv("THETA_N2_ENUM_LEGACY") && P.m >= 4 && (((unsigned long long)out) & 127ull) == 0ull && T <= 256 && count >= (unsigned long long)T * 4096ull) {
            const unsigned long long threads = (count + T - 1) / T;
            const unsigned blocks = (unsigned)((threads + 255) / 256);
            const size_t base = (((size_t)P.m * N2_KVS * 8 + (N2_KVS + 1 + 3) * 2 + P.m + 15) & ~(size_t)15);
            const size_t sm2 = base + (size_t)4 * WAVE * N2L_STRIDE * 4;
            const size_t smr = base + (size_t)4 * N2R_TILE_DWORDS * 4 + 64;            // render kernel: word-major tiles + ubpos
            if (const char *e = getenv("THETA_N2_ENUM_RENDER"); !e || atoi(e) > 0) {      // the default since round 3 (0: the summing whole-line writer)
                if (P.kv <= 8) {
                    (void)hipFuncSetAttribute((const void *)n2_enumerate_render_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smr);
                    hipLaunchKernelGGL(n2_enumerate_render_kernel<8>, dim3(blocks), dim3(256), smr, st, P, begin, count, T, out);
                } else {
                    (void)hipFuncSetAttribute((const void *)n2_enumerate_render_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smr);
                    hipLaunchKernelGGL(n2_enumerate_render_kernel<16>, dim3(blocks), dim3(256), smr, st, P, begin, count, T, out);
                }
                return;
            }
            if (P.kv <= 8) {
                (void)hipFuncSetAttribute((const void *)n2_enumerate_lines_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2);
                hipLaunchKernelGGL(n2_enumerate_lines_kernel<8>, dim3(blocks), dim3(256), sm2, st, P, begin, count, T, out);
            } else {
                (void)hipFuncSetAttribute((const void *)n2_enumerate_lines_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2);
                hipLaunchKernelGGL(n2_enumerate_lines_kernel<16>, dim3(blocks), dim3(256), sm2, st, P, begin, count, T, out);
            }
            return;
        }
    }
    const int per_thread = 16;     // (64 measured no better: the limit is the write pattern, one stream per lane)
    unsigned long long threads = (count + per_thread - 1) / per_thread;
    unsigned blocks = (unsigned)((threads + 255) / 256);
    size_t sm = n2_smem_bytes(P);
    if (P.kv <= 8)
        hipLaunchKernelGGL(n2_enumerate_kernel<8>, dim3(blocks), dim3(256), sm, st, P, begin, count, per_thread, out);
    else
        hipLaunchKernelGGL(n2_enumerate_kernel<16>, dim3(blocks), dim3(256), sm, st, P, begin, count, per_thread, out);
}

void n2_launch_unrank_list(const N2Dev &P, const TieRecord *recs, int count, unsigned char *out, hipStream_t st) {
    unsigned blocks = (unsigned)((count + 63) / 64);
    if (P.kv <= 8)
        hipLaunchKernelGGL(n2_unrank_list_kernel<8>, dim3(blocks), dim3(64), 0, st, P, recs, count, out);
    else
        hipLaunchKernelGGL(n2_unrank_list_kernel<16>, dim3(blocks), dim3(64), 0, st, P, recs, count, out);
}
