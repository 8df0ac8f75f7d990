// Materialised-candidate operators for gfx950 (compiled with -ffp-contract=off so that the
// per-interval arithmetic below rounds exactly like the reference's Python/numpy expressions):
//
//   solve_batch   Optimizer.solve on B given candidates          Optimizer.py:68-165
//   score_batch   CalcAllC.L2 / CalcAllC.L3 on literal matrices  CalcAllC.py:44-76
//   score_masked  the same likelihood on byte candidates x row masks (interval-subset resampling)
#include "n3_core.hpp"
#include "n3_refbfgs.hpp"

// ------------------------------------------------------------------------------------------------
// n = 2: faithful per-interval dL/dnu (Optimizer.py:208-221) + the Bus-Dekker/Brent hyperbolic
// root finder the reference calls (scipy.optimize.brenth, default xtol=2e-12, rtol=4*eps,
// maxiter=100), restated from its published algorithm.  One thread per candidate.
// ------------------------------------------------------------------------------------------------
struct N2Exact {
    int m;
    const double *w0, *rr, *rn;  // rN_i*tau, r_i, rN_i  (LDS)
    const unsigned char *c;      // the candidate's tumour column (global)
    double S0, S1;
    __device__ double f(double x) const {
        double acc = 0.0;
        double omx = 1.0 - x;
        for (int i = 0; i < m; i++) {
            double a = w0[i] / S0;
            double b = (rn[i] * (double)c[i]) / S1;
            double num = rr[i] * (a - b);
            acc = acc + num / ((a * x) + (b * omx));
        }
        return -acc;
    }
};

// returns 0 = converged, 1 = sign error, 2 = no convergence, 3 = NaN function value
__device__ int brenth_root(const N2Exact &F, double xa, double xb, double &root) {
    const double xtol = 2e-12, rtol = 8.881784197001252e-16;
    double xpre = xa, xcur = xb, xblk = 0.0, fblk = 0.0, spre = 0.0, scur = 0.0;
    double fpre = F.f(xpre), fcur = F.f(xcur);
    if (fpre != fpre || fcur != fcur) return 3;
    if (fpre == 0.0) { root = xpre; return 0; }
    if (fcur == 0.0) { root = xcur; return 0; }
    if (signbit(fpre) == signbit(fcur)) return 1;
    for (int it = 0; it < 100; it++) {
        if (fpre != 0.0 && fcur != 0.0 && signbit(fpre) != signbit(fcur)) {
            xblk = xpre;
            fblk = fpre;
            spre = scur = xcur - xpre;
        }
        if (fabs(fblk) < fabs(fcur)) {
            xpre = xcur; xcur = xblk; xblk = xpre;
            fpre = fcur; fcur = fblk; fblk = fpre;
        }
        double delta = (xtol + rtol * fabs(xcur)) / 2.0;
        double sbis = (xblk - xcur) / 2.0;
        if (fcur == 0.0 || fabs(sbis) < delta) { root = xcur; return 0; }
        if (fabs(spre) > delta && fabs(fcur) < fabs(fpre)) {
            double stry;
            if (xpre == xblk) {
                stry = -fcur * (xcur - xpre) / (fcur - fpre);            // secant
            } else {
                double dpre = (fpre - fcur) / (xpre - xcur);              // hyperbolic extrapolation
                double dblk = (fblk - fcur) / (xblk - xcur);
                stry = -fcur * (fblk - fpre) / (fblk * dpre - fpre * dblk);
            }
            double lim = fmin(fabs(spre), 3.0 * fabs(sbis) - delta);
            if (2.0 * fabs(stry) < lim) { spre = scur; scur = stry; }
            else { spre = sbis; scur = sbis; }
        } else {
            spre = sbis;
            scur = sbis;
        }
        xpre = xcur;
        fpre = fcur;
        if (fabs(scur) > delta) xcur += scur;
        else xcur += (sbis > 0.0 ? delta : -delta);
        fcur = F.f(xcur);
        if (fcur != fcur) return 3;
    }
    return 2;
}

__global__ __launch_bounds__(64) void solve_batch_n2_kernel(int m, int tau, const double *r, const double *rN,
                                                            double max_normal, int B, const unsigned char *C,
                                                            unsigned char *ok, double *mu, double *nll, double *vals) {
    extern __shared__ double sm[];
    double *w0 = sm, *rr = sm + m, *rn = sm + 2 * m;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        w0[i] = rN[i] * (double)tau;   // weighted_C column 0 (Optimizer.py:176-182)
        rr[i] = r[i];
        rn[i] = rN[i];
    }
    __syncthreads();
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    N2Exact F;
    F.m = m; F.w0 = w0; F.rr = rr; F.rn = rn;
    F.c = C + (size_t)b * m;
    double S0 = 0.0, S1 = 0.0;
    for (int i = 0; i < m; i++) {      // normalize_C column sums (Optimizer.py:169)
        S0 = S0 + w0[i];
        S1 = S1 + rn[i] * (double)F.c[i];
    }
    F.S0 = S0; F.S1 = S1;
    double lo = 0.0, hi = max_normal;
    if (hi != 1.0) {                   // M2_Rev (Optimizer.py:228-231)
        double num = -hi * S0;
        double den = (hi - 1.0) * S1 + num;
        hi = num / den;
    }
    double root = 0.0;
    int st = brenth_root(F, lo, hi, root);
    if (st != 0) {
        ok[b] = 0;
        mu[2 * b] = mu[2 * b + 1] = nll[b] = __builtin_nan("");
        if (vals) for (int i = 0; i < m; i++) vals[(size_t)b * m + i] = __builtin_nan("");
        return;
    }
    double num = -root * S1;           // M2 (Optimizer.py:223-226)
    double den = (root - 1.0) * S0 + num;
    double muv = num / den;
    double mu1 = 1.0 - muv;
    double dsum = 0.0;                 // L2 (Optimizer.py:187-196)
    for (int j = 0; j < m; j++) dsum = dsum + (w0[j] * muv + (rn[j] * (double)F.c[j]) * mu1);
    double tot = 0.0;
    for (int i = 0; i < m; i++) {
        double nm = w0[i] * muv + (rn[i] * (double)F.c[i]) * mu1;
        double p = nm / dsum;
        tot = tot + rr[i] * log(p);
        if (vals) vals[(size_t)b * m + i] = p;
    }
    ok[b] = 1;
    mu[2 * b] = muv;
    mu[2 * b + 1] = mu1;
    nll[b] = -tot;
}

// ------------------------------------------------------------------------------------------------
// n = 3: per-interval Newton (same core as the fused kernel, one likelihood term per interval),
// admissibility as Optimizer._solve_n3plus (Optimizer.py:150-160), then nu -> mu (closed form of M3)
// and Optimizer.L3 (Optimizer.py:236-244) in the reference's summation order.
// ------------------------------------------------------------------------------------------------
// What Optimizer._solve_n3plus reports for a candidate (Optimizer.py:128-165), decided the way the reference decides it:
//   1. fsolve (MINPACK hybrj, restated in hybrj4.hpp) on the Lagrangian system in the reference's operation order
//      (n3_refsys.hpp) from (1/3,1/3,1/3,1).  Whatever it returns is taken -- converged or not -- if every nu_j is in
//      [0,1] (NaN passes, Misc.py:49-57): ok = 1.
//   2. Otherwise fmin_bfgs is started from nu = (1/3,1/3) with dL3_hat as gradient, which points uphill
//      (Optimizer.py:255-265 against :246-252).  Its line searches fail and it hands back its start -- (1/3,1/3,1/3) is in
//      range and is reported: ok = 2 -- unless the first trial point lies outside the domain of the logarithms: then the
//      search accepts a step on the derivative alone, BFGS stops at a point with NaN likelihood, out of range, and the
//      reference returns None: ok = 0.  The decision sequence of scipy's search is restated in n3_refbfgs.hpp.
// On the reference's own m=6, K=3 table (21 050 entries) this reproduces the outcome class of every entry: 16 286 own
// optima, 4 467 fallbacks, 284 None (tools/hybrj_check.py; the 13 NaN entries are all-zero columns, ok = 0 above).
// nu -> mu is the closed form of M3 (Optimizer.py:318-330), the NLL Optimizer.L3's sums (Optimizer.py:236-244).
__global__ __launch_bounds__(64) void solve_batch_n3_kernel(int m, int tau, const double *r, const double *rN, int B,
                                                            const unsigned char *C, unsigned char *ok, double *mu,
                                                            double *nll, double *vals) {
    extern __shared__ double sm[];
    double *rr = sm, *rn = sm + m;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        rr[i] = r[i];
        rn[i] = rN[i];
    }
    __syncthreads();
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned char *c = C + (size_t)b * m * 2;
    N3RefSystem sys;
    sys.m = m;
    sys.tau = (double)tau;
    sys.r = rr;
    sys.rN = rn;
    sys.c = c;
    sys.init();
    // the reference's whole per-candidate procedure, restated (n3_refbfgs.hpp: n3_ref_solve) -- hybrj, the BFGS decision, M3's
    // hybrd call, L3's sums; all-zero columns and NaN likelihoods come out the way the reference reports them
    double mv[3], value = 0.0;
    const int outcome = n3_ref_solve(sys, mv, value, vals ? vals + (size_t)b * m : nullptr);
    if (outcome == 0) {
        ok[b] = 0;
        mu[3 * b] = mu[3 * b + 1] = mu[3 * b + 2] = nll[b] = __builtin_nan("");
        if (vals) for (int i = 0; i < m; i++) vals[(size_t)b * m + i] = __builtin_nan("");
        return;
    }
    // A NaN likelihood is still a solution tuple in the reference (Optimizer.py:162-165 has no check), and its driver
    // appends it to `best` through isClose(NaN) (Misc.py:44-46): ok stays 1 / 2, the host replays that.
    ok[b] = (unsigned char)outcome;                  // 2: the reference's nu = (1/3, 1/3, 1/3) fallback
    mu[3 * b] = mv[0];
    mu[3 * b + 1] = mv[1];
    mu[3 * b + 2] = mv[2];
    nll[b] = value;
}

// ------------------------------------------------------------------------------------------------
// The same procedure, ONE WAVE PER CANDIDATE (round 6): for the few dozen matrices a whole-space search hands over, the lane-per-candidate
// kernel above is all latency -- every likelihood term costs 6 IEEE divisions in `f`, 12 in `jac`, one lane walks the m terms of every
// evaluation, and one candidate in a hundred takes hundreds of evaluations (6 ms for 140 matrices of 200 intervals).  Here all 64 lanes
// run the procedure's control flow on the same numbers, and an evaluation's TERMS are spread over them: lane l computes terms l, l + 64,
// ... into LDS, then every lane adds the m values up in the reference's order, top to bottom -- the same operations on the same operands
// in the same order as N3RefSystem's loops, so the same bits (tests/test_gpu_parity.py compares the two kernels entry by entry).
// ------------------------------------------------------------------------------------------------
struct N3RefSystemW {
    int m, mp;
    double tau;
    const double *r, *rN;          // [m] (LDS)
    const unsigned char *c;        // [m][2]
    double *sc;                    // LDS scratch [9][mp]
    double S[3];

    __device__ void init() {
        for (int i = threadIdx.x; i < m; i += 64) {
            sc[i] = rN[i] * tau;
            sc[mp + i] = rN[i] * (double)c[2 * i];
            sc[2 * mp + i] = rN[i] * (double)c[2 * i + 1];
        }
        __syncthreads();
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < m; i++) {
            s0 = s0 + sc[i];
            s1 = s1 + sc[mp + i];
            s2 = s2 + sc[2 * mp + i];
        }
        __syncthreads();
        S[0] = s0;
        S[1] = s1;
        S[2] = s2;
    }
    __device__ void chat(int i, double &h0, double &h1, double &h2) const {
        h0 = (rN[i] * tau) / S[0];
        h1 = (rN[i] * (double)c[2 * i]) / S[1];
        h2 = (rN[i] * (double)c[2 * i + 1]) / S[2];
    }
    __device__ void f(const double *x, double *fv) const {
        for (int i = threadIdx.x; i < m; i += 64) {
            double h0, h1, h2;
            chat(i, h0, h1, h2);
            const double p = (h0 * x[1] + h1 * x[2]) + h2 * x[3];
            sc[i] = (r[i] * h0) / p;
            sc[mp + i] = (r[i] * h1) / p;
            sc[2 * mp + i] = (r[i] * h2) / p;
        }
        __syncthreads();
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int i = 0; i < m; i++) {
            a0 = a0 + sc[i];
            a1 = a1 + sc[mp + i];
            a2 = a2 + sc[2 * mp + i];
        }
        __syncthreads();
        fv[1] = (-a0) - x[4];
        fv[2] = (-a1) - x[4];
        fv[3] = (-a2) - x[4];
        fv[4] = 1.0 - ((x[1] + x[2]) + x[3]);
    }
    __device__ void jac(const double *x, double fj[hybrj4::N + 1][hybrj4::N + 1]) const {
        for (int i = threadIdx.x; i < m; i += 64) {
            double h[3];
            chat(i, h[0], h[1], h[2]);
            const double p = (h[0] * x[1] + h[1] * x[2]) + h[2] * x[3];
            const double den = refpow::square(p);
            for (int k = 0; k < 3; k++)
                for (int q = 0; q < 3; q++) sc[(3 * k + q) * mp + i] = ((r[i] * h[k]) * h[q]) / den;
        }
        __syncthreads();
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int i = 0; i < m; i++)
            for (int k = 0; k < 3; k++)
                for (int q = 0; q < 3; q++) J[k][q] = J[k][q] + sc[(3 * k + q) * mp + i];
        __syncthreads();
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < 3; q++) fj[k + 1][q + 1] = J[k][q];
        for (int k = 1; k <= 3; k++) {
            fj[4][k] = -1.0;
            fj[k][4] = -1.0;
        }
        fj[4][4] = 0.0;
    }
    __device__ double fhat(double v0, double v1) const {
        const double v2 = 1.0 - (v0 + v1);
        bool bad = false;
        for (int i = threadIdx.x; i < m; i += 64) {
            double h0, h1, h2;
            chat(i, h0, h1, h2);
            const double p = (h0 * v0 + h1 * v1) + h2 * v2;
            if (p < 0.0 || p != p) bad = true;
            else sc[i] = r[i] * log(p);
        }
        const bool any_bad = __syncthreads_or(bad) != 0;         // (the serial loop returns NaN at its first such term: the same value)
        double acc = 0.0;
        if (!any_bad)
            for (int i = 0; i < m; i++) acc = acc + sc[i];
        __syncthreads();
        return any_bad ? NAN : -acc;
    }
    __device__ void ghat(double v0, double v1, double &g0, double &g1) const {
        for (int i = threadIdx.x; i < m; i += 64) {
            double h0, h1, h2;
            chat(i, h0, h1, h2);
            const double n0 = h0 - h2, n1 = h1 - h2;
            const double den = (n0 * v0 + n1 * v1) + h2;
            sc[i] = r[i] * (n0 / den);
            sc[mp + i] = r[i] * (n1 / den);
        }
        __syncthreads();
        g0 = 0.0;
        g1 = 0.0;
        for (int i = 0; i < m; i++) {
            g0 = g0 + sc[i];
            g1 = g1 + sc[mp + i];
        }
        __syncthreads();
    }
    __device__ double l3(const double mu[3], double *vals) const {
        const double m0 = mu[0], m1 = mu[1], m2 = mu[2];
        for (int h = threadIdx.x; h < m; h += 64) {
            sc[h] = (rN[h] * tau) * m0;
            sc[mp + h] = (rN[h] * (double)c[2 * h]) * m1;
            sc[2 * mp + h] = (rN[h] * (double)c[2 * h + 1]) * m2;
        }
        __syncthreads();
        double den = 0.0;
        for (int h = 0; h < m; h++) den = den + sc[h];
        for (int h = 0; h < m; h++) den = den + sc[mp + h];
        for (int h = 0; h < m; h++) den = den + sc[2 * mp + h];
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += 64) {
            const double nm = ((rN[i] * tau) * m0 + (rN[i] * (double)c[2 * i]) * m1) + (rN[i] * (double)c[2 * i + 1]) * m2;
            const double p = nm / den;
            sc[i] = r[i] * log(p);
            if (vals) vals[i] = p;
        }
        __syncthreads();
        double tot = 0.0;
        for (int i = 0; i < m; i++) tot = tot + sc[i];
        __syncthreads();
        return -tot;
    }
};

__global__ __launch_bounds__(64) void solve_wave_n3_kernel(int m, int tau, const double *r, const double *rN, int B, const unsigned char *C,
                                                           unsigned char *ok, double *mu, double *nll, double *vals) {
    extern __shared__ double sm[];
    const int mp = (m + 1) & ~1;
    double *rr = sm, *rn = sm + mp;
    for (int i = threadIdx.x; i < m; i += 64) {
        rr[i] = r[i];
        rn[i] = rN[i];
    }
    __syncthreads();
    const int b = blockIdx.x;
    N3RefSystemW sys;
    sys.m = m;
    sys.mp = mp;
    sys.tau = (double)tau;
    sys.r = rr;
    sys.rN = rn;
    sys.c = C + (size_t)b * m * 2;
    sys.sc = sm + 2 * mp;
    sys.init();
    double mv[3], value = 0.0;
    const int outcome = n3_ref_solve(sys, mv, value, vals ? vals + (size_t)b * m : nullptr);
    if (outcome == 0) {
        if (threadIdx.x == 0) {
            ok[b] = 0;
            mu[3 * b] = mu[3 * b + 1] = mu[3 * b + 2] = nll[b] = __builtin_nan("");
        }
        if (vals) for (int i = threadIdx.x; i < m; i += 64) vals[(size_t)b * m + i] = __builtin_nan("");
        return;
    }
    if (threadIdx.x == 0) {
        ok[b] = (unsigned char)outcome;
        mu[3 * b] = mv[0];
        mu[3 * b + 1] = mv[1];
        mu[3 * b + 2] = mv[2];
        nll[b] = value;
    }
}

// ------------------------------------------------------------------------------------------------
// Smallest NLL a candidate can take anywhere on the BOUNDARY of the simplex (some nu_j = 0), per-interval
// sums.  For a candidate whose optimum lies outside the simplex this is the lowest value the reference's solver
// could ever report for it (its iterates stay inside [0,1]^3 or are rejected, Optimizer.py:150-160), so
// "boundary minimum > winner + tie margin" certifies that the candidate cannot change the result.
// Each face nu_j = 0 is a 1-D convex problem in t: p_i = Chat_ia t + Chat_ib (1 - t).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void boundary_min_n3_kernel(int m, int tau, const double *r, const double *rN, int B,
                                                             const unsigned char *C, double *bound) {
    extern __shared__ double sm[];
    double *rr = sm, *rn = sm + m;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        rr[i] = r[i];
        rn[i] = rN[i];
    }
    __syncthreads();
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const unsigned char *c = C + (size_t)b * m * 2;
    double S[3] = {0.0, 0.0, 0.0};
    for (int i = 0; i < m; i++) {
        S[0] += rn[i] * (double)tau;
        S[1] += rn[i] * (double)c[2 * i];
        S[2] += rn[i] * (double)c[2 * i + 1];
    }
    auto col = [&](int j, int i) -> double {   // normalised column entry Chat_ij (Optimizer.py:167-174)
        double cij = (j == 0) ? (double)tau : (double)c[2 * i + (j - 1)];
        return (rn[i] * cij) / S[j];
    };
    double best = __builtin_inf();
    for (int ja = 0; ja < 3; ja++)
        for (int jb = ja + 1; jb < 3; jb++) {
            if (S[ja] == 0.0 || S[jb] == 0.0) continue;   // an all-zero column cannot carry weight
            auto fval = [&](double t, double &g, double &h) {
                double f = 0.0;
                g = 0.0;
                h = 0.0;
                for (int i = 0; i < m; i++) {
                    double a = col(ja, i), bb = col(jb, i);
                    double p = a * t + bb * (1.0 - t);
                    double d = a - bb;
                    f -= rr[i] * log(p);
                    g -= rr[i] * d / p;
                    h += rr[i] * d * d / (p * p);
                }
                return f;
            };
            // minimise over t in [0,1]: bisection on the sign of the derivative, then Newton polish
            double lo = 0.0, hi = 1.0, g, h;
            double f0 = fval(0.0, g, h);
            double g0 = g;
            double f1 = fval(1.0, g, h);
            double g1 = g;
            double fmin_face;
            if (!(g0 < 0.0)) fmin_face = f0;                // increasing from t = 0 (or NaN/inf): vertex
            else if (!(g1 > 0.0)) fmin_face = f1;
            else {
                double t = 0.5;
                for (int it = 0; it < 200; it++) {
                    fval(t, g, h);
                    if (g > 0.0) hi = t; else lo = t;
                    double tn = t - g / h;
                    if (!(tn > lo && tn < hi)) tn = 0.5 * (lo + hi);
                    if (fabs(tn - t) <= 1e-15) { t = tn; break; }
                    t = tn;
                }
                fmin_face = fval(t, g, h);
            }
            if (fmin_face == fmin_face && fmin_face < best) best = fmin_face;
            if (f0 == f0 && f0 < best) best = f0;
            if (f1 == f1 && f1 < best) best = f1;
        }
    bound[b] = best;
}

// ------------------------------------------------------------------------------------------------
// CalcAllC.L2 / L3 on literal float matrices: one wave per matrix, lanes over rows.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

__global__ __launch_bounds__(64) void score_batch_kernel(int n, int m, int B, const double *Cw, const double *mu,
                                                         const double *r_all, int r_stride, double *nll, double *vals,
                                                         unsigned char *valid) {
    int b = blockIdx.x;
    int lane = threadIdx.x;
    const double *r = r_all + (size_t)b * r_stride;     // r_stride = 0: one r for all matrices; m: one per matrix
    const double *M = Cw + (size_t)b * m * n;
    const double *mv = mu + (size_t)b * n;
    double m0 = mv[0];
    double den = 0.0;
    // pass 1: C.mu per row and the masked denominator
    for (int i = lane; i < m; i += WAVE) {
        double s;
        bool v;
        if (n == 2) {
            v = (m0 != 0.0) ? (M[i * 2] != 0.0) : (M[i * 2 + 1] != 0.0);          // CalcAllC.py:49-52
            s = M[i * 2] * m0 + M[i * 2 + 1] * (1.0 - m0);                        // CalcAllC.py:54-56
        } else {
            v = M[i * 3] != 0.0;                                                  // CalcAllC.py:70
            s = (M[i * 3] * m0 + M[i * 3 + 1] * mv[1]) + M[i * 3 + 2] * mv[2];    // CalcAllC.py:71
        }
        den += s * (v ? 1.0 : 0.0);
    }
    den = wave_sum_f64(den);
    double tot = 0.0;
    for (int i = lane; i < m; i += WAVE) {
        double s;
        bool v;
        if (n == 2) {
            v = (m0 != 0.0) ? (M[i * 2] != 0.0) : (M[i * 2 + 1] != 0.0);
            s = M[i * 2] * m0 + M[i * 2 + 1] * (1.0 - m0);
        } else {
            v = M[i * 3] != 0.0;
            s = (M[i * 3] * m0 + M[i * 3 + 1] * mv[1]) + M[i * 3 + 2] * mv[2];
        }
        double p = s / den;
        tot += (log(p) * (v ? 1.0 : 0.0)) * r[i];   // log(0)*0 = NaN is part of the contract (quirk Q10)
        if (vals) vals[(size_t)b * m + i] = p;
        if (valid) valid[(size_t)b * m + i] = v ? 1 : 0;
    }
    tot = wave_sum_f64(tot);
    if (lane == 0) nll[b] = -tot;
}

// ------------------------------------------------------------------------------------------------
// Interval-subset resampling: B byte candidates x S row masks.  One wave per candidate: the row
// terms C.mu and r ln(C.mu) are computed once and kept in registers (4 rows per lane cover
// m <= 256); every mask then costs one 8-byte word load per lane-row group and three wave sums.
// ------------------------------------------------------------------------------------------------
#define SM_ROWS 4
__global__ __launch_bounds__(256) void score_masked_kernel(int n, int m, int tau, int B, int S, const unsigned char *C,
                                                           const double *w, const double *r, const double *mu,
                                                           const unsigned long long *mask, double *nll) {
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wv;
    if (b >= B) return;
    const int nc = n - 1;
    const unsigned char *c = C + (size_t)b * m * nc;
    const double *mv = mu + (size_t)b * n;
    const double m0 = mv[0], m1 = (n == 2) ? 1.0 - mv[0] : mv[1], m2 = (n == 3) ? mv[2] : 0.0;
    const int words = (m + 63) / 64;
    double cm[SM_ROWS], tl[SM_ROWS], rr[SM_ROWS], off[SM_ROWS];
    // row k of this lane is interval k*64 + lane: each mask word is then one coalesced 8-byte load
#pragma unroll
    for (int k = 0; k < SM_ROWS; k++) {
        int i = k * 64 + lane;
        cm[k] = tl[k] = rr[k] = off[k] = 0.0;
        if (i < m) {
            double x = (double)c[i * nc], y = (nc == 2) ? (double)c[i * nc + 1] : 0.0;
            double tum = x * m1 + y * m2;
            cm[k] = w[i] * ((double)tau * m0 + tum);
            off[k] = w[i] * tum;             // value of the row once its column 0 is zeroed
            rr[k] = r[i];
            tl[k] = rr[k] * log(cm[k]);
        }
    }
    for (int s = 0; s < S; s++) {
        double den = 0.0, tot = 0.0, rs = 0.0;
        bool poison = false;
#pragma unroll
        for (int k = 0; k < SM_ROWS; k++) {
            if (k < words) {
                unsigned long long wd = mask ? mask[(size_t)s * words + k] : ~0ull;
                int i = k * 64 + lane;
                bool on = (wd >> lane) & 1ull;
                if (i < m) {
                    if (on) { den += cm[k]; tot += tl[k]; rs += rr[k]; }
                    else if (!(off[k] > 0.0)) poison = true;   // ln(0) * 0 = NaN in the reference (quirk Q10)
                }
            }
        }
        den = wave_sum_f64(den);
        tot = wave_sum_f64(tot);
        rs = wave_sum_f64(rs);
        bool anyp = ballot64(poison) != 0ull;
        if (lane == 0) nll[(size_t)b * S + s] = anyp ? __builtin_nan("") : -(tot - rs * log(den));
    }
}

// ------------------------------------------------------------------------------------------------
// The same operator as a dense FP64 GEMM on the matrix cores: for a block of 16 candidates the masked sums
// over intervals are   D[S x 32] = MASK[S x m] . X[m x 32],   X[:, 2c] = C.mu of candidate c, X[:, 2c+1] =
// r ln(C.mu): the 0/1 mask matrix is shared by all candidates -- a genuine GEMM (K = m), so it runs on
// v_mfma_f64_16x16x4_f64.  The X tile is built once per block (all the logarithms) and stays in LDS; each
// wave walks 16-mask chunks, 4 rows of X per MFMA step.  Fragment maps (f64 form): A[i][k]: i = lane&15,
// k = lane>>4; B[k][j]: j = lane&15, k = lane>>4; D[row][col]: col = lane&15, row = (lane>>4) + 4*reg.
// ------------------------------------------------------------------------------------------------
typedef double mfma_d4 __attribute__((ext_vector_type(4)));

#include "smx_log.hpp"
#define SMX_CAND 16
#define SMX_WAVES 8
#define SMX_MAXM 256

__global__ __launch_bounds__(64) void mask_rsum_kernel(int m, int S, const double *r, const unsigned long long *mask,
                                                       double *rsum) {
    const int s = blockIdx.x, lane = threadIdx.x;     // one wave per mask
    const int words = (m + 63) / 64;
    double acc = 0.0;
    for (int i = lane; i < m; i += WAVE)
        if ((mask[(size_t)s * words + (i >> 6)] >> (i & 63)) & 1ull) acc += r[i];
    acc = wave_sum_f64(acc);
    if (lane == 0) rsum[s] = acc;
}

// Block = 8 waves and 16 candidates.  X (dynamic LDS) is [row][candidate]{C.mu, r ln C.mu}: one ds_read_b128 hands a lane
// the B operands of BOTH accumulators (fragment 0 = the sums of C.mu, fragment 1 = the sums of r ln C.mu of the same 16
// candidates), so D[mask][candidate] of the two fragments meet in the same lane and register: the epilogue needs no
// exchange.  On gfx950 the f64 MFMA does not overlap with vector instructions of other waves (tools/micro/mfma_f64_rate.hip),
// so what counts is the number of vector instructions per MFMA; the A operand -- the mask bits -- is built with ONE:
//   * K index: MFMA step `st` of 16-row group g takes matrix row g*16 + 4 lk + st for K-lane lk -- a lane's four bits of a
//     group are adjacent in the mask word (one shift pair per group puts them on bits 27..30);
//   * A = the double whose high word is the single exponent bit 27+st, i.e. 2^-895, 2^-767, 2^-511, 2 for st = 0..3 -- one
//     v_and_b32 --, and phase 1 stores rows with (row & 3) == st multiplied by the inverse power of two (v_ldexp_f64):
//     the products are exactly the unscaled terms.
// LDS: m rows of 256 B, then the logarithm table (2 KB), which doubles as the rows m.. of the last 16-row group (finite
// numbers under A = 0: mask bits >= m are cleared when the words are loaded), zero fill up to the group boundary if any
// is left, then the Q10 bitmaps -- 53 760 B at m = 200: three blocks = 24 waves per CU.
extern __shared__ double smx_lds[];
__device__ __forceinline__ size_t smx_x_bytes(int m) {
    const size_t a = (size_t)m * 256 + SMX_TAB_BYTES, b = (size_t)((m + 15) & ~15) * 256;
    return a > b ? a : b;
}
template <int NG>                                                             // NG = 16-row groups = ceil(m / 16)
__global__ __launch_bounds__(64 * SMX_WAVES) void score_masked_mfma_kernel(int n, int m, int tau, int B, int S,
                                                                           const unsigned char *C, const double *w, const double *r,
                                                                           const double *mu, const unsigned long long *mask,
                                                                           const double *rsum, double *nll) {
    constexpr int words = (NG + 3) / 4;
    const int nc = n - 1;
    double2 *const X = (double2 *)smx_lds;                                    // [row][16]: {C.mu, r ln C.mu} of candidate c
    double2 *const tab = X + (size_t)m * 16;
    const size_t xb = smx_x_bytes(m);
    unsigned long long(*const zrow)[4] = (unsigned long long(*)[4])((char *)smx_lds + xb);   // [SMX_CAND][4]: rows that
    const int b0 = blockIdx.x * SMX_CAND;                                     // become 0 once masked (NaN poison, quirk Q10)
    smx_log_stage(tab);
    for (size_t i = (size_t)m * 256 + SMX_TAB_BYTES + threadIdx.x * 8; i < xb; i += blockDim.x * 8) *(double *)((char *)smx_lds + i) = 0.0;
    for (int i = threadIdx.x; i < SMX_CAND * 4; i += blockDim.x) (&zrow[0][0])[i] = 0ull;
    __syncthreads();
    // ---- phase 1: the per-row terms of the 16 candidates; a wave covers 4 rows x 16 candidates per pass (256 B runs of LDS)
    {
        const int c = threadIdx.x & 15, b = b0 + c;
        const bool live = b < B;
        double m0 = 0, m1 = 0, m2 = 0;
        if (live) {
            const double *mv = mu + (size_t)b * n;
            m0 = (double)tau * mv[0];
            m1 = (n == 2) ? 1.0 - mv[0] : mv[1];
            m2 = (n == 3) ? mv[2] : 0.0;
        }
        const int st = (threadIdx.x >> 4) & 3;                                // = row & 3: the MFMA step that consumes the row
        const int ex = st == 0 ? 895 : st == 1 ? 767 : st == 2 ? 511 : -1;
        // the candidate bytes of ALL of this thread's rows first (2-byte loads 400 bytes apart between lanes: uncoalesced, a round
        // trip to HBM each) -- in flight together; issued one per pass of the loop below they were seven dependent round trips
        // before the block's first MFMA
        constexpr int NIT = (16 * NG + 31) / 32;
        unsigned short cby[NIT];
        double wv_[NIT], rv_[NIT];
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = (threadIdx.x >> 4) + 32 * it;
            cby[it] = 0;
            wv_[it] = rv_[it] = 0.0;
            if (live && i < m) {
                const unsigned char *cc = C + ((size_t)b * m + i) * nc;
                cby[it] = (unsigned short)(cc[0] | (nc == 2 ? (unsigned)cc[1] << 8 : 0u));
                wv_[it] = w[i];
                rv_[it] = r[i];
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int i = (threadIdx.x >> 4) + 32 * it;
            if (i < m) {
                double cm = 0.0, tl = 0.0;
                if (live) {
                    const double x = (double)(cby[it] & 0xffu), y = (double)(cby[it] >> 8);
                    const double tum = x * m1 + y * m2;
                    cm = wv_[it] * (m0 + tum);
                    tl = rv_[it] * smx_log(cm, tab);
                    if (!(wv_[it] * tum > 0.0)) atomicOr(&zrow[c][i >> 6], 1ull << (i & 63));
                }
                X[i * 16 + c] = double2{ldexp(cm, ex), ldexp(tl, ex)};
            }
        }
    }
    __syncthreads();
    // (quirk Q10 needs a look at the mask only if some row of these candidates becomes 0 once its column 0 is masked)
    bool any_zrow = false;
    for (int i = 0; i < SMX_CAND * 4; i++) any_zrow |= (&zrow[0][0])[i] != 0ull;
    // ---- phase 2: masked sums on the matrix cores
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const unsigned sh0 = lk * 4, sh1 = lk * 4 + 16;
    const double2 *const xl = X + lk * 64 + li;                               // row 4 lk (+ st), candidate li
    for (int s0 = wv * 16; s0 < S; s0 += 16 * SMX_WAVES) {
        const int sA = s0 + li;                             // mask row this lane feeds into A
        unsigned long long mw[4] = {0, 0, 0, 0};
        if (sA < S) {
            const unsigned long long *mp = mask + (size_t)sA * words;
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (q < words) mw[q] = (q == words - 1 && (m & 63)) ? (mp[q] & ((1ull << (m & 63)) - 1ull)) : mp[q];
        }
        mfma_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        // 16 rows of X per group, the NG groups unrolled; the four 16-byte LDS reads of group g+1 are
        // issued before the eight MFMAs of group g (the scheduling barrier keeps the compiler from sinking them again)
        double2 bv[4];
#pragma unroll
        for (int st = 0; st < 4; st++) bv[st] = xl[st * 16];
#pragma unroll
        for (int g = 0; g < NG; g++) {
            {
                double2 nx[4] = {bv[0], bv[1], bv[2], bv[3]};
                if (g + 1 < NG) {
#pragma unroll
                    for (int st = 0; st < 4; st++) nx[st] = xl[((g + 1) * 16 + st) * 16];
                }
                const unsigned long long wq = mw[g / 4];
                const unsigned half = (g & 2) ? (unsigned)(wq >> 32) : (unsigned)wq;
                const unsigned al = (half >> ((g & 1) ? sh1 : sh0)) << 27;    // this lane's bits of the group on bits 27..30
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 4; st++) {
                    const double a = __hiloint2double((int)(al & (1u << (27 + st))), 0);
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[st].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[st].y, acc1, 0, 0, 0);
                }
#pragma unroll
                for (int st = 0; st < 4; st++) bv[st] = nx[st];
            }
        }
        // D[row = lk + 4 reg][col = li]: acc0 = sum of C.mu (den), acc1 = sum of r ln(C.mu) (tot) of mask s0 + row, candidate li
        const int b = b0 + li;
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int s = s0 + lk + 4 * reg;
            if (s < S && b < B) {
                bool poison = false;
                if (any_zrow) {
                    const unsigned long long *mp = mask + (size_t)s * words;
                    for (int q = 0; q < words; q++) {
                        unsigned long long live_bits = (q == words - 1 && (m & 63)) ? ((1ull << (m & 63)) - 1ull) : ~0ull;
                        if (~mp[q] & zrow[li][q] & live_bits) poison = true;
                    }
                }
                nll[(size_t)b * S + s] = poison ? __builtin_nan("") : -(acc1[reg] - rsum[s] * smx_log(acc0[reg], tab));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// No masks: the plain CalcAllC.L2 / L3 score of B byte candidates, one THREAD per candidate.  The wave-per-candidate kernel above
// pays three wave reductions for every candidate; here a block stages its 256 candidates in LDS with coalesced 4-byte loads
// (m (n-1) bytes each -- the algorithmic HBM traffic, plus 8 n bytes of mu in and 8 bytes out) and every thread walks its own
// candidate's rows: two byte reads, the row's C.mu, one logarithm (smx_log) and two FMAs per interval.  Bound: the FP64
// logarithm (~15 instructions per interval), not HBM.  NLL = -(sum r ln(C.mu) - sum r ln(sum C.mu)).
// ------------------------------------------------------------------------------------------------
extern __shared__ unsigned int spc_lds[];
#define SPT_VALUES 8       // n = 2: logarithms tabulated per candidate (copy numbers 0..7; larger ones take the direct logarithm)
// A tile = TS consecutive candidates (cw dwords each, contiguous in global memory) staged into LDS rows of odd stride pw
// (lanes then read their own rows conflict-free): 16-byte loads, eight per thread in flight before the first LDS store.
// (Round 2 staged with dword loads and a store after each: a chain of dependent round trips to HBM.  A persistent,
// double-buffered form -- tile k+1 in registers while tile k is scored -- was tried in round 3 and lost: the 64 registers of
// the prefetch cost the occupancy the scoring phase needs.)
__device__ __forceinline__ void spc_stage(const unsigned char *src_bytes, int nb, int cw, int pw) {
    const int total = nb * cw;                                  // dwords
    const int nq = total >> 2;                                  // whole 16-byte chunks (a tile starts 16-byte aligned)
    const uint4 *src4 = (const uint4 *)src_bytes;
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)cw - 1) / (unsigned)cw);
    if (pw == cw) {                                             // odd record length: the LDS image IS the global one -- a straight copy
        uint4 *const dst4 = (uint4 *)spc_lds;
        for (int base = 0; base < nq; base += 256 * 8) {
            uint4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int q = base + u * 256 + (int)threadIdx.x;
                v[u] = src4[q < nq ? q : nq - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; u++)                         // (all eight loads in flight before the first store: the optimizer would sink each load to its store)
                asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int q = base + u * 256 + (int)threadIdx.x;
                if (q < nq) dst4[q] = v[u];
            }
        }
        const unsigned int *src = (const unsigned int *)src_bytes;
        for (int idx = (nq << 2) + (int)threadIdx.x; idx < total; idx += 256) spc_lds[idx] = src[idx];
        return;
    }
    for (int base = 0; base < nq; base += 256 * 8) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = base + u * 256 + (int)threadIdx.x;
            v[u] = src4[q < nq ? q : nq - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));   // (as above)
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = base + u * 256 + (int)threadIdx.x;
            if (q < nq) {
                const unsigned vals[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const unsigned g = 4u * (unsigned)q + (unsigned)k;
                    const unsigned cand = __umulhi(g, magic);      // g / cw (exact: g (cw - 1) < 2^32)
                    spc_lds[cand * (unsigned)pw + (g - cand * (unsigned)cw)] = vals[k];
                }
            }
        }
    }
    const unsigned int *src = (const unsigned int *)src_bytes;
    for (int idx = (nq << 2) + (int)threadIdx.x; idx < total; idx += 256) {          // (a last tile's tail of < 4 dwords)
        const unsigned cand = __umulhi((unsigned)idx, magic);
        spc_lds[cand * (unsigned)pw + ((unsigned)idx - cand * (unsigned)cw)] = src[idx];
    }
}

// TABLE (n = 2 only): with one tumour column the row term is C.mu = w_i (tau mu0 + x mu1); its logarithm is
// ln w_i + ln(tau mu0 + x mu1) -- the first part a constant of the problem (sum r_i ln w_i, summed once on the host), the
// second one of a handful of values per candidate (x is a copy number).  A thread computes SPT_VALUES logarithms once, keeps
// them in its column of an LDS table [value][thread] (the bank depends on the thread only: conflict-free) and walks its m rows
// with one 8-byte table read and two FMAs each (r_i ln(..) into the total, w_i x into sum w_i x_i: the sum of C.mu is
// tau mu0 sum w_i + mu1 sum w_i x_i, sum w_i from the host): ~6 vector instructions per interval instead of 24.  Requires every
// w_i > 0 (else the direct form, whose per-interval order of operations then decides between inf and NaN).
template <int NC, bool TABLE>
__global__ __launch_bounds__(256) void score_plain_kernel(int m, int tau, int B, const unsigned char *C, const double *__restrict__ w,
                                                          const double *__restrict__ r, const double *mu, double rsum, double rlogw, double wsum, double wmin, double wmax, double *nll) {
    static_assert(!TABLE || NC == 1, "the table form is for one tumour column");
    const int cb = m * NC;                       // bytes per candidate, a multiple of 4 (checked by the launcher)
    const int cw = cb >> 2, pw = cw | 1;         // words per candidate; odd LDS stride: lanes fall on distinct banks
    const int TS = cw <= 64 ? 256 : 128;         // candidates per tile (records beyond 256 bytes: half tiles, the LDS rows are what limits them)
    const int tab_off = (TS * pw + 3) & ~3;
    const double2 *const tab = (const double2 *)(spc_lds + tab_off);                       // smx_log's table (16-byte aligned)
    double2 *const wr = (double2 *)(spc_lds + tab_off + SMX_TAB_BYTES / 4);                // {w_i, r_i}: every lane reads the same pair (a broadcast)
    double *const T = (double *)(wr + ((m + 3) & ~3));                                     // TABLE: [SPT_VALUES][256]
    const long long b0 = (long long)blockIdx.x * TS;
    const int nb = (int)((long long)B - b0 < TS ? (long long)B - b0 : TS);
    double mu_a, mu_b, mu_c = 0.0;
    const int nq = (nb * cw) >> 2;                                                         // whole 16-byte chunks of the tile
    if (pw == cw && nq >= 1 && nq <= 8 * 256 && m <= 512) {
        // The common shape (odd record length in words, a tile of at most 32 KB): EVERYTHING the block reads from global memory
        // -- its tile, the logarithm table, {w, r}, this thread's mu -- is requested before the first wait: one exposed round
        // trip per block instead of four in a row (table, {w, r}, tile, and mu behind the barrier).  A block lives ~15 us.
        const int tid = threadIdx.x;
        const uint4 *src4 = (const uint4 *)(C + (size_t)b0 * cb);
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = u * 256 + tid;
            v[u] = src4[q < nq ? q : nq - 1];
        }
        double2 tv = ((const double2 *)smx_log_table)[tid & 127];
        const int i0 = tid < m ? tid : m - 1, i1 = tid + 256 < m ? tid + 256 : m - 1;
        double w0 = w[i0], r0 = r[i0], w1 = w[i1], r1 = r[i1];
        const double *mv = mu + (size_t)(b0 + (tid < nb ? tid : 0)) * (NC + 1);
        mu_a = mv[0];
        mu_b = mv[1];
        if (NC == 2) mu_c = mv[2];
#pragma unroll
        for (int u = 0; u < 8; u++) asm volatile("" : "+v"(v[u].x), "+v"(v[u].y), "+v"(v[u].z), "+v"(v[u].w));
        asm volatile("" : "+v"(tv.x), "+v"(tv.y), "+v"(w0), "+v"(r0), "+v"(w1), "+v"(r1), "+v"(mu_a), "+v"(mu_b), "+v"(mu_c));
        if (tid < 128) ((double2 *)tab)[tid] = double2{2.0 * tv.x, tv.y};                  // (smx_log_stage)
        if (tid < m) wr[tid] = double2{w0, r0};
        if (tid + 256 < m) wr[tid + 256] = double2{w1, r1};
        uint4 *const dst4 = (uint4 *)spc_lds;
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int q = u * 256 + tid;
            if (q < nq) dst4[q] = v[u];
        }
        const unsigned int *src = (const unsigned int *)src4;
        for (int idx = (nq << 2) + tid; idx < nb * cw; idx += 256) spc_lds[idx] = src[idx];
        __syncthreads();
        if (tid >= nb) return;
    } else {
        smx_log_stage((double2 *)tab);
        for (int i = threadIdx.x; i < m; i += 256) wr[i] = double2{w[i], r[i]};
        spc_stage(C + (size_t)b0 * cb, nb, cw, pw);
        __syncthreads();
        if ((int)threadIdx.x >= nb) return;
        const double *mv = mu + (size_t)(b0 + threadIdx.x) * (NC + 1);
        mu_a = mv[0];
        mu_b = mv[1];
        if (NC == 2) mu_c = mv[2];
    }
    const long long b = b0 + threadIdx.x;
    const double m0 = (double)tau * mu_a, m1 = (NC == 1) ? 1.0 - mu_a : mu_b, m2 = mu_c;
    double den = 0.0, tot = 0.0;
    if constexpr (TABLE) {
        double *const mine = T + threadIdx.x;
#pragma unroll
        for (int v = 0; v < SPT_VALUES; v++) mine[v * 256] = smx_log(__builtin_fma((double)v, m1, m0), tab);
        const unsigned int *row = spc_lds + threadIdx.x * pw;
        // Branch-free walk: the table index is the value's low three bits, and values beyond the table are noticed word by
        // word (one and + compare per four rows); a candidate that has one -- copy numbers 8..15, rare -- is redone below
        // with the direct logarithm.  (A test and branch per row cost more than the row's arithmetic.)
        bool big = false;
        double wx = 0.0;                                         // sum w_i x_i: sum C.mu = tau mu0 sum w_i + mu1 sum w_i x_i
        for (int q = 0; q < cw; q++) {
            const unsigned int word = row[q];
            big |= (word & ~(0x01010101u * (SPT_VALUES - 1))) != 0u;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const double2 wi = double2{w[4 * q + k], r[4 * q + k]};       // (wave-uniform: scalar loads, SGPR operands)
                wx = __builtin_fma(wi.x, (double)((word >> (8 * k)) & 0xffu), wx);
                tot = __builtin_fma(wi.y, mine[__builtin_amdgcn_ubfe(word, 8 * k, 3) * 256], tot);
            }
        }
        den = __builtin_fma(m1, wx, m0 * wsum);
        if (big) {
            tot = 0.0;
            for (int q = 0; q < cw; q++) {
                const unsigned int word = row[q];
                for (int k = 0; k < 4; k++) {
                    const unsigned x = (word >> (8 * k)) & 0xffu;
                    const double lg = x < SPT_VALUES ? mine[x * 256] : smx_log(__builtin_fma((double)x, m1, m0), tab);
                    tot = __builtin_fma(wr[4 * q + k].y, lg, tot);
                }
            }
        }
        nll[b] = -((tot + rlogw) - rsum * smx_log(den, tab));
    } else {
        const unsigned char *row = (const unsigned char *)(spc_lds + threadIdx.x * pw);
        // Branch-free walk with the unguarded logarithm.  That needs every row term w_i (m0 + x m1 + y m2) to be a positive normal
        // number, which is decided per candidate from the extremes -- weights in [wmin, wmax] (from the host), copy numbers in
        // 0..255, m1, m2 >= 0: the terms lie between wmin m0 and wmax (m0 + 255 (m1 + m2)), and rounding is monotone -- instead
        // of a test per row.  A candidate that fails (w_i = 0, mu outside the simplex, ...) is done with the guarded logarithm
        // below: same operations in the same order, so the same bits.
        const bool odd = !(m1 >= 0.0 && m2 >= 0.0 && smx_log_fast_ok(wmin * m0) &&
                           smx_log_fast_ok(wmax * __builtin_fma(255.0, m1, __builtin_fma(255.0, m2, m0))));
        auto term = [&](int i) {
            const double2 wi = double2{w[i], r[i]};                   // (wave-uniform: scalar loads, SGPR operands)
            const double x = (double)row[i * NC], y = (NC == 2) ? (double)row[i * NC + 1] : 0.0;
            const double cm = wi.x * __builtin_fma(x, m1, __builtin_fma(y, m2, m0));
            den += cm;
            tot = __builtin_fma(wi.y, smx_log_fast(cm, tab), tot);
        };
        int i = 0;
        for (; i + 4 <= m; i += 4) {                             // (unrolled by hand: smx_log's asm statements keep the optimizer from it)
            term(i);
            term(i + 1);
            term(i + 2);
            term(i + 3);
        }
        for (; i < m; i++) term(i);
        if (odd) {
            den = 0.0;
            tot = 0.0;
            for (i = 0; i < m; i++) {
                const double2 wi = wr[i];
                const double x = (double)row[i * NC], y = (NC == 2) ? (double)row[i * NC + 1] : 0.0;
                const double cm = wi.x * __builtin_fma(x, m1, __builtin_fma(y, m2, m0));
                den += cm;
                tot = __builtin_fma(wi.y, smx_log(cm, tab), tot);
            }
        }
        nll[b] = -(tot - rsum * smx_log(den, tab));
    }
}


// Records beyond 256 bytes (m = 200, n = 3: 400 bytes): the kernel above holds whole records in LDS, which caps a block at 128
// candidates -- half of its threads idle, six waves per CU, 1.2 TB/s.  Here a block scores 256 candidates SLICE BY SLICE: equal
// parts of at most 28 words of each record at a time, in LDS rows of odd stride; every thread walks its own row and carries
// den / tot across the slices in registers.  All 256 threads work, < 32 KB of LDS per block (five blocks per CU), and the global
// reads of a slice are 16-byte loads of consecutive lanes.  Same per-candidate operations in the same order as
// score_plain_kernel: the same bits.  (m = 200, n = 3: 1.22 -> 1.97 TB/s of the algorithmic bytes; the row rate is that of the
// m = 50 shape less the slices' barriers -- the kernel is bound by the logarithm's 15 vector instructions per interval, so the
// byte rate of long records cannot exceed ~2.4 TB/s: their candidates carry fewer bytes of mu / NLL per interval.)
#define SPS_MAXW 28        // most words of a slice: slices are equal parts of a record, multiples of 4 words (16-byte staging)
__host__ __device__ inline int sps_slice_words(int cw, int maxw = SPS_MAXW) {
    const int ns = (cw + maxw - 1) / maxw;
    return (((cw + ns - 1) / ns) + 3) & ~3;
}
template <int NC>
__global__ __launch_bounds__(256) void score_plain_sliced_kernel(int m, int tau, int B, const unsigned char *C, const double *__restrict__ w,
                                                                 const double *__restrict__ r, const double *mu, double rsum, double wmin, double wmax, double *nll, int maxw) {
    const int cb = m * NC, cw = cb >> 2;         // bytes / words per candidate (a multiple of 4 bytes: the launcher checks)
    const int SW = sps_slice_words(cw, maxw), STR = SW | 1;                                      // words per slice; odd LDS stride: lanes on distinct banks
    unsigned int *const tile = spc_lds;                                                    // [256][STR]
    const int tid = threadIdx.x;
    const long long b0 = (long long)blockIdx.x * 256;
    const int nb = (int)((long long)B - b0 < 256 ? (long long)B - b0 : 256);
    double2 *const tabw = (double2 *)(spc_lds + ((256 * STR + 3) & ~3));
    smx_log_stage(tabw);
    const double2 *const tb = tabw;
    double mu_a = 0.0, mu_b = 0.0, mu_c = 0.0;
    if (tid < nb) {
        const double *mv = mu + (size_t)(b0 + tid) * (NC + 1);
        mu_a = mv[0];
        mu_b = mv[1];
        if (NC == 2) mu_c = mv[2];
    }
    const double m0 = (double)tau * mu_a, m1 = (NC == 1) ? 1.0 - mu_a : mu_b, m2 = mu_c;
    // (the unguarded logarithm needs positive normal row terms: decided per candidate from the extremes, see score_plain_kernel; a wave
    // that holds one odd candidate takes the guarded logarithm for all of its lanes -- the same bits wherever both apply)
    const bool odd = !(m1 >= 0.0 && m2 >= 0.0 && smx_log_fast_ok(wmin * m0) &&
                       smx_log_fast_ok(wmax * __builtin_fma(255.0, m1, __builtin_fma(255.0, m2, m0))));
    const bool guarded = __ballot(odd && tid < nb) != 0ull;
    const bool vec = (cb & 15) == 0 && (((uintptr_t)C) & 15) == 0;      // records start 16-byte aligned: 16-byte loads
    const unsigned int *const gsrc = (const unsigned int *)(C + (size_t)b0 * cb);
    const unsigned int *const row = tile + tid * STR;
    double den = 0.0, tot = 0.0;
    for (int s0 = 0; s0 < cw; s0 += SW) {
        const int sw = cw - s0 < SW ? cw - s0 : SW;                     // words of this slice
        __syncthreads();                                                // (the previous slice has been walked)
        if (vec && (sw & 3) == 0) {
            const int per = sw >> 2, total = nb * per;                  // 16-byte chunks per record slice / of the tile slice
            const unsigned magic = (unsigned)((0x100000000ull + (unsigned)per - 1) / (unsigned)per);
            for (int q = tid; q < total; q += 256) {
                const int cand = (int)__umulhi((unsigned)q, magic), ch = q - cand * per;       // q / per (exact: q (per - 1) < 2^32)
                const uint4 v = *(const uint4 *)(gsrc + (size_t)cand * cw + s0 + 4 * ch);
                unsigned int *d = tile + cand * STR + 4 * ch;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            const int total = nb * sw;
            const unsigned magic = (unsigned)((0x100000000ull + (unsigned)sw - 1) / (unsigned)sw);
            for (int q = tid; q < total; q += 256) {
                const int cand = (int)__umulhi((unsigned)q, magic), wd = q - cand * sw;
                tile[cand * STR + wd] = gsrc[(size_t)cand * cw + s0 + wd];
            }
        }
        __syncthreads();
        if (tid < nb) {
            auto term = [&](int i, unsigned x8, unsigned y8) {
                const double2 wi = double2{w[i], r[i]};                   // (wave-uniform: scalar loads, SGPR operands)
                const double x = (double)x8, y = (double)y8;
                const double cm = wi.x * __builtin_fma(x, m1, __builtin_fma(y, m2, m0));
                den += cm;
                tot = __builtin_fma(wi.y, guarded ? smx_log(cm, tb) : smx_log_fast(cm, tb), tot);
            };
            if (guarded) {
                for (int q = 0; q < sw; q++) {
                    const unsigned word = row[q];
                    const int i = ((s0 + q) * 4) / NC;
                    if (NC == 2) {
                        term(i, word & 0xffu, (word >> 8) & 0xffu);
                        term(i + 1, (word >> 16) & 0xffu, word >> 24);
                    } else {
                        for (int k = 0; k < 4; k++) term(i + k, (word >> (8 * k)) & 0xffu, 0u);
                    }
                }
            } else {
                for (int q = 0; q < sw; q++) {
                    const unsigned word = row[q];
                    const int i = ((s0 + q) * 4) / NC;
                    if (NC == 2) {
                        term(i, word & 0xffu, (word >> 8) & 0xffu);
                        term(i + 1, (word >> 16) & 0xffu, word >> 24);
                    } else {
                        term(i, word & 0xffu, 0u);
                        term(i + 1, (word >> 8) & 0xffu, 0u);
                        term(i + 2, (word >> 16) & 0xffu, 0u);
                        term(i + 3, word >> 24, 0u);
                    }
                }
            }
        }
    }
    if (tid < nb) nll[b0 + tid] = -(tot - rsum * smx_log(den, tb));
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define SOLVE_WAVE_MAX_B 2048      // (the chip holds 4 096 waves of this kernel at once: beyond that a lane per candidate has the better rate)
void batch_launch_solve(int n, int m, int tau, const double *r, const double *rN, double max_normal, int B,
                        const unsigned char *C, unsigned char *ok, double *mu, double *nll, double *vals,
                        hipStream_t st) {
    unsigned blocks = (unsigned)((B + 63) / 64);
    if (n == 2)
        hipLaunchKernelGGL(solve_batch_n2_kernel, dim3(blocks), dim3(64), (size_t)m * 3 * sizeof(double), st, m, tau, r, rN,
                           max_normal, B, C, ok, mu, nll, vals);
    else if (B <= SOLVE_WAVE_MAX_B && m <= 512 && !getenv("THETA_SOLVE_NO_WAVE"))
        // a small batch (the records of a whole-space search): one wave per candidate -- the same bits, a thirtieth of the latency
        hipLaunchKernelGGL(solve_wave_n3_kernel, dim3((unsigned)B), dim3(64), (size_t)((m + 1) & ~1) * 11 * sizeof(double), st, m, tau, r, rN,
                           B, C, ok, mu, nll, vals);
    else
        hipLaunchKernelGGL(solve_batch_n3_kernel, dim3(blocks), dim3(64), (size_t)m * 2 * sizeof(double), st, m, tau, r, rN,
                           B, C, ok, mu, nll, vals);
}

void batch_launch_boundary_min(int m, int tau, const double *r, const double *rN, int B, const unsigned char *C,
                               double *bound, hipStream_t st) {
    hipLaunchKernelGGL(boundary_min_n3_kernel, dim3((B + 63) / 64), dim3(64), (size_t)m * 2 * sizeof(double), st, m, tau, r, rN, B,
                       C, bound);
}

void batch_launch_score(int n, int m, int B, const double *Cw, const double *mu, const double *r, int r_stride, double *nll,
                        double *vals, unsigned char *valid, hipStream_t st) {
    hipLaunchKernelGGL(score_batch_kernel, dim3(B), dim3(64), 0, st, n, m, B, Cw, mu, r, r_stride, nll, vals, valid);
}

void batch_launch_score_masked(int n, int m, int tau, int B, int S, const unsigned char *C, const double *w,
                               const double *r, const double *mu, const unsigned long long *mask, double *nll,
                               double *rsum_scratch, hipStream_t st, double rsum_host, bool rsum_host_valid, double rlogw_host,
                               bool rlogw_valid, double wsum_host, double wmin_host, double wmax_host) {
    if (mask != nullptr && S >= 16 && rsum_scratch != nullptr) {   // enough masks to fill the 16-row MFMA tiles
        hipLaunchKernelGGL(mask_rsum_kernel, dim3(S), dim3(64), 0, st, m, S, r, mask, rsum_scratch);
        const size_t xa = (size_t)m * 256 + SMX_TAB_BYTES, xb = (size_t)((m + 15) & ~15) * 256;
        const size_t lds = (xa > xb ? xa : xb) + SMX_CAND * 4 * sizeof(unsigned long long);
        // (set on every launch: the attribute belongs to the current device's copy of the kernel, and costs nothing)
        const dim3 grid((B + SMX_CAND - 1) / SMX_CAND), block(64 * SMX_WAVES);
        // (the attribute is set on every launch: it belongs to the current device's copy of the kernel, and costs nothing)
#define SMX_LAUNCH(NG)                                                                                                            \
    case NG:                                                                                                                      \
        (void)hipFuncSetAttribute((const void *)score_masked_mfma_kernel<NG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(score_masked_mfma_kernel<NG>, grid, block, lds, st, n, m, tau, B, S, C, w, r, mu, mask, rsum_scratch, nll); \
        break;
        switch ((m + 15) / 16) {
            SMX_LAUNCH(1) SMX_LAUNCH(2) SMX_LAUNCH(3) SMX_LAUNCH(4) SMX_LAUNCH(5) SMX_LAUNCH(6) SMX_LAUNCH(7) SMX_LAUNCH(8)
            SMX_LAUNCH(9) SMX_LAUNCH(10) SMX_LAUNCH(11) SMX_LAUNCH(12) SMX_LAUNCH(13) SMX_LAUNCH(14) SMX_LAUNCH(15) SMX_LAUNCH(16)
        }
#undef SMX_LAUNCH
    } else if (mask == nullptr && S == 1 && ((m * (n - 1)) & 3) == 0 && m * (n - 1) > 256 && (((uintptr_t)C) & 3) == 0 && rsum_scratch != nullptr && rsum_host_valid &&
               !getenv("THETA_SCORE_NO_SLICES")) {
        // records beyond 256 bytes: 256 candidates per block, slice by slice (score_plain_sliced_kernel)
        int maxw = SPS_MAXW;
        // (>= 8: with 4-word slices a chunk is one candidate and the kernel's division-by-multiplication constant wraps -- round-4 advice)
        if (const char *e = getenv("THETA_SPS_MAXW")) maxw = atoi(e) >= 8 && atoi(e) <= 64 ? atoi(e) & ~3 : maxw;
        const size_t lds = (((size_t)256 * (sps_slice_words((m * (n - 1)) >> 2, maxw) | 1) + 3) & ~(size_t)3) * 4 + SMX_TAB_BYTES;
        const unsigned blocks = (unsigned)(((long long)B + 255) / 256);
        if (n == 2) hipLaunchKernelGGL((score_plain_sliced_kernel<1>), dim3(blocks), dim3(256), lds, st, m, tau, B, C, w, r, mu, rsum_host, wmin_host, wmax_host, nll, maxw);
        else hipLaunchKernelGGL((score_plain_sliced_kernel<2>), dim3(blocks), dim3(256), lds, st, m, tau, B, C, w, r, mu, rsum_host, wmin_host, wmax_host, nll, maxw);
    } else if (mask == nullptr && S == 1 && ((m * (n - 1)) & 3) == 0 && m * (n - 1) <= 512 && (((uintptr_t)C) & 15) == 0 && rsum_scratch != nullptr && rsum_host_valid) {
        const int ts = (m * (n - 1)) / 4 <= 64 ? 256 : 128;           // candidates per tile (batch.hip: score_plain_kernel)
        const size_t lds = (((size_t)ts * (((m * (n - 1)) >> 2) | 1) + 3) & ~(size_t)3) * 4 + SMX_TAB_BYTES + (size_t)((m + 3) & ~3) * 16;
        const long long tiles = ((long long)B + ts - 1) / ts;
        const unsigned blocks = (unsigned)tiles;
        if (n == 2 && rlogw_valid && !getenv("THETA_SCORE_NO_TABLE")) {      // every w_i > 0: the table-driven form
            const size_t ldt = lds + (size_t)SPT_VALUES * 256 * sizeof(double);
            (void)hipFuncSetAttribute((const void *)score_plain_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldt);
            hipLaunchKernelGGL((score_plain_kernel<1, true>), dim3(blocks), dim3(256), ldt, st, m, tau, B, C, w, r, mu, rsum_host, rlogw_host, wsum_host, wmin_host, wmax_host, nll);
        } else if (n == 2) {
            (void)hipFuncSetAttribute((const void *)score_plain_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((score_plain_kernel<1, false>), dim3(blocks), dim3(256), lds, st, m, tau, B, C, w, r, mu, rsum_host, 0.0, 0.0, wmin_host, wmax_host, nll);
        } else {
            (void)hipFuncSetAttribute((const void *)score_plain_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL((score_plain_kernel<2, false>), dim3(blocks), dim3(256), lds, st, m, tau, B, C, w, r, mu, rsum_host, 0.0, 0.0, wmin_host, wmax_host, nll);
        }
    } else {
        // (a HIP grid holds fewer than 2^32 threads: 2^24 candidates -- one wave each -- per launch)
        const int chunk = 1 << 24;
        for (long long b0 = 0; b0 < B; b0 += chunk) {
            const int nb = (int)((long long)B - b0 < chunk ? (long long)B - b0 : chunk);
            hipLaunchKernelGGL(score_masked_kernel, dim3((nb + 3) / 4), dim3(256), 0, st, n, m, tau, nb, S,
                               C + (size_t)b0 * m * (n - 1), w, r, mu + (size_t)b0 * n, mask, nll + (size_t)b0 * S);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Does THIS machine's libm square like the restatement (refpow.hpp)?  The reference's outcome on a rank-deficient n=3 candidate
// hangs on the last bit of numpy's `x ** 2`, i.e. of libm's pow(x, 2.0) under its interpreter: the
// kernels restate glibc >= 2.28 on x86-64 with FMA.  On another libm the reference itself reports other values for those
// candidates; the Python driver says so in its report (search.last_report.libm_pow_matches).  Returns the number of arguments
// (out of `n` seeded ones, including the ~1 in 1 300 where x * x differs from pow) on which the two disagree.
// ------------------------------------------------------------------------------------------------------------------
extern "C" int theta_refpow_check(int n, int *mismatches) {
    if (!mismatches || n < 0) {
        theta_set_error("theta_refpow_check: bad argument");
        return THETA_ERR_ARG;
    }
    volatile double two = 2.0;                     // (keeps the call a call: the compiler must not fold pow(x, 2) into x * x)
    unsigned long long s = 0x9e3779b97f4a7c15ull;
    int bad = 0;
    for (int i = 0; i < n; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const int e = (int)((s >> 52) % 80) - 40;
        const double x = ldexp(1.0 + (double)(s & 0xfffffffffffffull) / 4503599627370496.0, e);
        const double a = refpow::square(x), b = pow(x, two);
        if (memcmp(&a, &b, 8) != 0) bad++;
    }
    *mismatches = bad;
    return THETA_OK;
}
