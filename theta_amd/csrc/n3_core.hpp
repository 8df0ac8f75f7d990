// n = 3: pieces shared by the fused search kernel (n3.hip) and the per-candidate batch solver
// (batch.hip): the row alphabet / edge rules of the reference's enumerator and the Newton core of
// the mixture solve.
#pragma once
#include "common.hpp"

// The row alphabet of an n=3 search holds at most 64 rows (a, b): a child mask is one 64-bit word, an alphabet slot is one lane.
// Up to K = 7 the alphabet is the whole grid (K+1)^2 (slot = a + (K+1) b, rows the reference's _is_valid_row rejects included,
// masked out).  Beyond -- the reference's own bounds heuristic produces ub = max(k, y + 1) with y = round(tau ratio),
// DataTools.py:64-66: an interval at four times the normal ratio has ub = 9 -- the alphabet is COMPACT: the valid rows that lie
// within the bounds of at least one interval, in grid order (the order of Enumerator._create_graph, Enumerator.py:272-298, so
// that the enumeration order is unchanged).  Such instances have narrow per-interval windows (lb = max(tau, y - 1)): a search
// whose bounds leave more than 64 usable rows is refused.  Copy numbers themselves go up to 15 (4-bit fields of a packed node).
#define N3_MAX_COPY 15
#define N3_GRID_K 7              // up to this K the alphabet is the full grid
#define N3_MAX_Q 64
#define N3_MAX_M 64              // one interval per lane (fused kernel, generators)
#define N3_MAX_M_WIDE 256        // up to four prefix intervals per lane: the sieve path (n3_sieve.hip), the burst generator, task and unrank kernels
#define N3_MAX_TASKS (1 << 18)
#define N3_STB 256               // stride of the per-task prefix states (one packed DFS node per depth)
#define N3_RIDX_W (2 * N3_MAX_COPY + 1)

// Wave-uniform description of one n=3 search instance.
struct N3Dev {
    int m, K, Q, tau;            // Q alphabet slots; slot s <-> row (a, b) through rowtab (K <= 7: the grid, a = s % (K+1), b = s / (K+1))
    int NT;                      // number of distinct finite ratio values; lo in [0..NT], hi in [1..NT+1]
    int L;                       // levels enumerated by lanes (leaf levels); D = m - L prefix levels
    double N, Rtot, K0;
    const double *r, *rN;        // [m] as doubles (exact)
    const unsigned char *lb, *ub;   // [m] order-adjusted bounds
    const unsigned char *ridx;   // [N3_RIDX_W * N3_RIDX_W] rank of the ratio dy/(-dx) in the sorted table (1-based)
    const u128 *cnt;             // [m][Q][2][NT+1][NT+1] completions below a DFS node
    const unsigned char *rowtab; // [Q] slot -> a | b << 4
    const unsigned long long *smask; // [m][N3_MAX_Q] slots that may follow a parent row at depth d (static rules)
    const unsigned long long *dynmask; // [Q][NT+1][NT+1] slots that keep the ratio window [lo,hi] non-empty after a parent row
    unsigned long long swmask;       // slots with a <= b
    double warm_blend;           // weight of the previous optimum in the warm start (rest: simplex centre)
    int force64;                 // 1: iterate every candidate in FP64 (THETA_N3_FORCE_F64; the packed-f32 pass is the default)
    double conv_l2;              // convergence threshold on the squared Newton decrement
    double mu_tol;               // > 0: a candidate also counts as converged only where ONE more Newton step is certified to end within this of its
                                 // optimum IN MU (n3_sieve.hip: sv_mu_limit; option "n3_mu_tol"); 0: the decrement alone decides
    int no_dismiss;              // 1: never finish a candidate by its lower bound (THETA_N3_NO_DISMISS): every one is iterated to the coarse tolerance
    int prefix_bound;            // 1: the sieve finishes a whole prefix by the lower bound of its relaxed problem (search mode; n3_sieve.hip: sv_prefix_beyond)
    int no_second;               // 1: no second evaluation in place in the tight full-solve modes (n3_sieve.hip: sv_children; option "n3_second" = 0: A/B)
    unsigned long long total_lo, total_hi;
};

__host__ __device__ inline size_t n3_cnt_index(const N3Dev &P, int d, int slot, int sw, int lo, int hi) {
    // lo in [0..NT] (0 = -inf), hi in [1..NT+1] (NT+1 = +inf) stored as hi-1
    return ((((size_t)d * P.Q + slot) * 2 + sw) * (P.NT + 1) + lo) * (P.NT + 1) + (hi - 1);
}

struct N3Task {
    uint64_t base_lo, base_hi;   // rank of the first candidate of the task
    uint64_t count;              // candidates in the task
    uint64_t skip;               // leaves of the first prefix that precede base
};

// Are the points (x_i, y_i) of a candidate's rows collinear?  Then its three columns (tau, x, y) are linearly dependent, the
// bordered Jacobian of the reference's Lagrangian system is exactly singular at every iterate, and what MINPACK makes of it
// has nothing to do with the candidate's optimum: hybrj may stop, unconverged, at a nu inside [0,1]^3 whose components do not
// sum to one -- Optimizer._solve_n3plus takes it (Optimizer.py:150-153), M3 turns it into a mu with a negative entry, and L3
// reports NaN (or a finite value below the candidate's true minimum).  Such a candidate takes part in the reference's result
// wherever it stands (a NaN likelihood is "close" to anything, Misc.py:44-46), so the search hands EVERY rank-deficient
// candidate to the reference's own procedure (n3_ref_solve) instead of its own solver: the kernels list them next to the
// candidates with an all-zero tumour column (x = 0 or y = 0 for all rows: a special case of collinear).
// Incremental test in integers: kind 0 = no point yet, 1 = one distinct point, 2 = a line through (a0, b0) with direction
// (da, db), 3 = three points not on a line (full rank; final).
struct N3Line {
    int kind, a0, b0, da, db;
};
__host__ __device__ inline void n3_line_add(N3Line &s, int a, int b) {
    if (s.kind == 0) {
        s.a0 = a;
        s.b0 = b;
        s.kind = 1;
    } else if (s.kind == 1) {
        if (a != s.a0 || b != s.b0) {
            s.da = a - s.a0;
            s.db = b - s.b0;
            s.kind = 2;
        }
    } else if (s.kind == 2) {
        if (s.da * (b - s.b0) - s.db * (a - s.a0) != 0) s.kind = 3;
    }
}

struct N3Host {
    int m = 0, K = 0, Q = 0, NT = 0;
    bool mix_only = false;       // more than N3_MAX_Q rows within the bounds: no masks, no counting table -- only the searches that need no
                                 // ranks (theta_mix_search) and the batch operators serve such a problem
    std::vector<int> lb, ub;
    std::vector<unsigned char> ridx, rowtab;
    std::vector<unsigned long long> smask, dynmask;
    unsigned long long swmask = 0;
};

struct N3State {
    int slot;   // row at this depth, slot = a + (K+1) b
    int sw;     // 1 while every row so far had a == b (Enumerator.py:181-183,199-202)
    int lo, hi; // feasible interval of the ratio mu1/mu2 as ranks in the sorted ratio table
    int a, b;   // the row itself
};

// Enumerator._is_valid_row with allow_multi_event hard-wired True (Enumerator.py:55,262-264,286).
__host__ __device__ inline bool n3_valid_row(int a, int b, int tau) { return (tau - a) * (tau - b) >= 0; }

// First row of a matrix (Enumerator.py:175-186): in bounds, a <= b.  (a, b) given, no division.
__host__ __device__ inline bool n3_first_row_ab(const N3Dev &P, int a, int b, int slot, N3State &out) {
    if (!n3_valid_row(a, b, P.tau)) return false;
    int l = P.lb[0], u = P.ub[0];
    if (a < l || a > u || b < l || b > u) return false;
    if (a > b) return false;
    out.slot = slot;
    out.sw = (a == b);
    out.lo = 0;
    out.hi = P.NT + 1;
    out.a = a;
    out.b = b;
    return true;
}
__host__ __device__ inline bool n3_first_row(const N3Dev &P, int slot, N3State &out) {
    const unsigned rw = P.rowtab[slot];
    return n3_first_row_ab(P, (int)(rw & 15u), (int)(rw >> 4), slot, out);
}

// One DFS edge (Enumerator.py:192-212): may row (a, b) at depth d follow state `par`?
__host__ __device__ inline bool n3_edge_ab(const N3Dev &P, const N3State &par, int a, int b, int slot, int d, N3State &out) {
    if (!n3_valid_row(a, b, P.tau)) return false;
    int l = P.lb[d], u = P.ub[d];
    if (a < l || a > u || b < l || b > u) return false;            // _in_bounds, :241
    int pa = par.a, pb = par.b;
    if (!(slot == par.slot || a > pa || b > pb)) return false;     // _is_valid_edge, :258-260
    int sw = 0;
    if (par.sw) {                                                  // symmetry breaking, :199-202
        if (a > b) return false;
        sw = (a == b);
    }
    int lo = par.lo, hi = par.hi;
    int dx = a - pa, dy = b - pb;
    if (dx != 0 && dy != 0) {                                      // _get_mu_bounds, :225-239
        int t = P.ridx[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)];
        if (dx > 0) lo = (t > lo) ? t : lo; else hi = (t < hi) ? t : hi;
    }
    if (lo > hi) return false;                                     // :212
    out.slot = slot;
    out.sw = sw;
    out.lo = lo;
    out.hi = hi;
    out.a = a;
    out.b = b;
    return true;
}
__host__ __device__ inline bool n3_edge(const N3Dev &P, const N3State &par, int slot, int d, N3State &out) {
    const unsigned rw = P.rowtab[slot];
    return n3_edge_ab(P, par, (int)(rw & 15u), (int)(rw >> 4), slot, d, out);
}

#ifdef __HIPCC__
// 32-bit form of a DFS node: slot 7 bits | sw | lo 8 | hi 8 | a 4 | b 4
__device__ __forceinline__ unsigned n3_pack(const N3State &s) {
    return (unsigned)s.slot | ((unsigned)s.sw << 7) | ((unsigned)s.lo << 8) | ((unsigned)s.hi << 16) |
           ((unsigned)s.a << 24) | ((unsigned)s.b << 28);
}
__device__ __forceinline__ N3State n3_unpack(unsigned v) {
    N3State s;
    s.slot = v & 0x7f;
    s.sw = (v >> 7) & 1;
    s.lo = (v >> 8) & 0xff;
    s.hi = (v >> 16) & 0xff;
    s.a = (v >> 24) & 0xf;
    s.b = (v >> 28) & 0xf;
    return s;
}

// Dynamic part of the edge test for a child that already passed the static mask (valid row, bounds,
// edge rule) and the symmetry mask: only the ratio window remains (Enumerator.py:204-212).
__device__ __forceinline__ bool n3_child_dyn(const unsigned char *ridx, const unsigned char *rowtab, const N3State &par,
                                             int slot, N3State &out) {
    unsigned rw = rowtab[slot];
    int a = rw & 15, b = rw >> 4;
    int lo = par.lo, hi = par.hi;
    int dx = a - par.a, dy = b - par.b;
    if (dx != 0 && dy != 0) {
        int t = ridx[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)];
        if (dx > 0) lo = (t > lo) ? t : lo; else hi = (t < hi) ? t : hi;
    }
    out.slot = slot;
    out.sw = par.sw && (a == b);
    out.lo = lo;
    out.hi = hi;
    out.a = a;
    out.b = b;
    return lo <= hi;
}

// Next prefix in DFS order (wave-uniform): lane l holds the packed nodes of depths l, 64 + l, ... in st[0], st[1], ... (NS per lane:
// m <= 64 NS + ML).  Returns false at the end of the space.
template <int NS>
__device__ __forceinline__ unsigned n3_lane_state(const unsigned (&st)[NS], int d) {
    unsigned v = st[0];
#pragma unroll
    for (int j = 1; j < NS; j++) v = (d >> 6) == j ? st[j] : v;          // (d is wave-uniform)
    return (unsigned)__builtin_amdgcn_readlane((int)v, d & (WAVE - 1));
}
template <int NS>
__device__ __forceinline__ bool n3_next_prefix(const N3Dev &P, unsigned (&st)[NS], int D, int lane) {
    const int Q = P.Q;
    const unsigned myrow = lane < Q ? P.rowtab[lane] : 0u;          // Q <= 64: one alphabet slot per lane
    const int sa = (int)(myrow & 15u), sb = (int)(myrow >> 4);
    int d = D - 1;
    bool fresh = false;
    while (true) {
        const int cur_slot = (int)(n3_lane_state<NS>(st, d) & 0x7fu);
        const int start = fresh ? 0 : cur_slot + 1;
        const N3State pst = n3_unpack(n3_lane_state<NS>(st, d > 0 ? d - 1 : 0));
        N3State nx{0, 0, 0, 0, 0, 0};
        const bool ok = lane >= start && lane < Q &&
                        (d == 0 ? n3_first_row_ab(P, sa, sb, lane, nx) : n3_edge_ab(P, pst, sa, sb, lane, d, nx));
        const unsigned long long mk = ballot64(ok);
        if (mk) {
            const int first = __builtin_ctzll(mk);
            const unsigned mine = ok ? n3_pack(nx) : 0u;
            const unsigned packed = (unsigned)__builtin_amdgcn_readlane((int)mine, first);
#pragma unroll
            for (int j = 0; j < NS; j++)
                if ((d >> 6) == j && lane == (d & (WAVE - 1))) st[j] = packed;
            if (d == D - 1) return true;
            d++;
            fresh = true;
        } else {
            d--;
            fresh = false;
            if (d < 0) return false;
        }
    }
}

#endif

#ifdef __HIPCC__
// ---------------------------------------------------------------------------------------------
// Mixture solve for n = 3 in the scaled variables u_j = nu_j N / S_j (S_j = column sums of the
// weighted matrix, sigma_j = S_j / N):  with u_0 eliminated through sum_j sigma_j u_j = 1,
//      p_i = (rN_i / N) q_i,     q_i = 1 + (x_i - sigma_1) u_1 + (y_i - sigma_2) u_2,
//      NLL(u) = K0 - sum_i r_i ln q_i      (convex; R-weighted log barrier => self-concordant)
// One Newton step of that 2-D problem.  `terms(body)` calls body(x, y, R) for every likelihood
// term (interval or group of intervals sharing a row).  Restates the stationarity system of
// Optimizer.equations / jacobian (Optimizer.py:273-316) after eliminating the multiplier.
// ---------------------------------------------------------------------------------------------
struct N3Newton {
    double u1, u2;       // current iterate
    double p1, p2;       // previous feasible iterate (for the q <= 0 safeguard)
    int iters;
    int status;          // 0 running, 1 converged, 2 failed (diverged / iteration cap)
    bool singular;       // Hessian numerically rank-deficient at the last evaluation
};

// 2x2 Hessian of a converged point, only needed by n3_admissible for rank-deficient candidates
struct N3Hess {
    double u1, u2, h11, h12, h22;
};

#define N3_MAX_ITERS 60

template <class Terms>
__device__ __forceinline__ void n3_newton_step(Terms &&terms, double s1, double s2, double inv_Rtot, N3Newton &S,
                                               double conv_l2 = 1e-12) {
    double g1 = 0, g2 = 0, h11 = 0, h12 = 0, h22 = 0;
    bool bad = false;
    const double u1 = S.u1, u2 = S.u2;
    terms([&](double x, double y, double R) {
        double a = x - s1, b = y - s2;
        double q = __builtin_fma(a, u1, __builtin_fma(b, u2, 1.0));
        bad |= !(q > 0.0);
        double w = rcp_nr1(q);
        double t = R * w;
        g1 = __builtin_fma(t, a, g1);
        g2 = __builtin_fma(t, b, g2);
        double tw = t * w;
        double ta = tw * a, tb = tw * b;
        h11 = __builtin_fma(ta, a, h11);
        h12 = __builtin_fma(ta, b, h12);
        h22 = __builtin_fma(tb, b, h22);
    });
    S.iters++;
    if (bad) {  // outside the domain: halve the step; a START outside it falls back towards u = 0 (interior for every candidate)
        if (S.p1 == S.u1 && S.p2 == S.u2) S.p1 = S.p2 = 0.0;
        S.u1 = 0.5 * (S.u1 + S.p1);
        S.u2 = 0.5 * (S.u2 + S.p2);
        if (S.iters >= N3_MAX_ITERS) S.status = 2;
        return;
    }
    const double hh = h11 * h22;
    S.singular = (hh - h12 * h12) <= 1e-10 * hh;
    if (h11 + h22 == 0.0) {  // every row equals (sigma1, sigma2): the likelihood does not depend on u
        S.status = 1;
        return;
    }
    double reg = 1e-13 * (h11 + h22);              // Levenberg floor: rank-deficient candidates stay solvable
    double a11 = h11 + reg, a22 = h22 + reg;
    double det = a11 * a22 - h12 * h12;
    double idet = rcp_nr2(det);
    double d1 = (a22 * g1 - h12 * g2) * idet;      // H d = g   (g = -grad NLL)
    double d2 = (a11 * g2 - h12 * g1) * idet;
    double l2 = (g1 * d1 + g2 * d2) * inv_Rtot;    // squared Newton decrement of NLL / Rtot
    if (!(l2 == l2) || !(fabs(d1) + fabs(d2) < 1e30)) {  // NaN / overflow: give up on this candidate
        S.status = 2;
        return;
    }
    double step = 1.0;
    if (l2 > 0.09) step = 1.0 / (1.0 + sqrt(l2));          // damped phase keeps q > 0 (self-concordance)
    S.p1 = u1; S.p2 = u2;
    S.u1 = __builtin_fma(step, d1, u1);
    S.u2 = __builtin_fma(step, d2, u2);
    if (l2 < conv_l2) S.status = 1;             // quadratic phase: the step just taken leaves an error ~lam^2
    else if (S.iters >= N3_MAX_ITERS || fabs(S.u1) + fabs(S.u2) > 1e8) S.status = 2;
}

// ---- packed single-precision evaluation for the COARSE first pass of the fused search -------------------------
// Two likelihood terms per VALU instruction (v_pk_fma_f32 & co.).  Only the coarse pass uses it: the iterate it
// produces is screened in single precision anyway, and every contender is polished and evaluated in FP64.  With
// gradient noise dg ~ 1e-7 sum|t a| the optimum moves by H^-1 dg, an NLL error ~ dg^2 / H ~ 1e-14 sum(r).
// The 2x2 solve is single precision too.  Returns false -- without stepping -- when the Hessian is too
// ill-conditioned for single-precision sums (det < 1e-3 h11 h22, which includes the rank-deficient candidates):
// the caller then iterates that candidate with n3_newton_step in FP64.
typedef float v2f __attribute__((ext_vector_type(2)));
#define N3_COND_MIN 1e-3   // smallest det / (h11 h22) the single-precision pass accepts

// VAL: also return sum R log2 q and the decrement lambda^2 / sum(r), both at the point the evaluation was made at (the
// iterate BEFORE the step) -- what the caller needs for the self-concordance lower bound of the optimum.
template <bool VAL, class Pairs>
__device__ __forceinline__ bool n3_newton_step_pk(Pairs &&pairs, float s1, float s2, double inv_Rtot, N3Newton &S,
                                                  double conv_l2, float &val_log2, float &l2_out) {
    v2f g1 = {0.f, 0.f}, g2 = g1, h11 = g1, h12 = g1, h22 = g1, lg = g1;
    float qmin = __builtin_inff();
    l2_out = -1.0f;
    const float u1 = (float)S.u1, u2 = (float)S.u2;
    const v2f vs1 = {s1, s1}, vs2 = {s2, s2}, vu1 = {u1, u1}, vu2 = {u2, u2}, one = {1.f, 1.f};
    pairs([&](v2f x, v2f y, v2f R) {
        v2f a = x - vs1, b = y - vs2;
        v2f q = __builtin_elementwise_fma(a, vu1, __builtin_elementwise_fma(b, vu2, one));
        qmin = fminf(qmin, fminf(q.x, q.y));
        v2f w = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
        if (VAL) lg = __builtin_elementwise_fma(R, v2f{__builtin_amdgcn_logf(q.x), __builtin_amdgcn_logf(q.y)}, lg);
        v2f t = R * w;
        g1 = __builtin_elementwise_fma(t, a, g1);
        g2 = __builtin_elementwise_fma(t, b, g2);
        v2f tw = t * w;
        v2f ta = tw * a, tb = tw * b;
        h11 = __builtin_elementwise_fma(ta, a, h11);
        h12 = __builtin_elementwise_fma(ta, b, h12);
        h22 = __builtin_elementwise_fma(tb, b, h22);
    });
    S.iters++;
    if (!(qmin > 0.0f)) {   // outside the domain: halve the step; a START outside it falls back towards u = 0
        if (S.p1 == S.u1 && S.p2 == S.u2) S.p1 = S.p2 = 0.0;
        S.u1 = 0.5 * (S.u1 + S.p1);
        S.u2 = 0.5 * (S.u2 + S.p2);
        if (S.iters >= N3_MAX_ITERS) S.status = 2;
        return true;
    }
    // 2x2 solve in single precision as well: det > 1e-3 h11 h22 bounds the cancellation (relative error <= ~1e-4
    // in the step, irrelevant for a pass that only has to reach lambda^2 < conv_l2)
    const float G1 = g1.x + g1.y, G2 = g2.x + g2.y;
    const float H11 = h11.x + h11.y, H12 = h12.x + h12.y, H22 = h22.x + h22.y;
    const float hh = H11 * H22;
    const float det = __builtin_fmaf(-H12, H12, hh);
    if (!(det > (float)N3_COND_MIN * hh)) return false;
    S.singular = false;
    const float idet = __builtin_amdgcn_rcpf(det);
    const float d1 = (H22 * G1 - H12 * G2) * idet;
    const float d2 = (H11 * G2 - H12 * G1) * idet;
    const float l2 = (G1 * d1 + G2 * d2) * (float)inv_Rtot;
    if (VAL) {
        val_log2 = lg.x + lg.y;
        l2_out = l2;
    }
    if (!(l2 == l2) || !(fabsf(d1) + fabsf(d2) < 1e30f)) {
        S.status = 2;
        return true;
    }
    float step = 1.0f;
    if (l2 > 0.09f) step = __builtin_amdgcn_rcpf(1.0f + __builtin_sqrtf(l2));
    S.p1 = S.u1; S.p2 = S.u2;
    S.u1 = (double)__builtin_fmaf(step, d1, u1);
    S.u2 = (double)__builtin_fmaf(step, d2, u2);
    if (l2 < (float)conv_l2) S.status = 1;
    else if (S.iters >= N3_MAX_ITERS || fabs(S.u1) + fabs(S.u2) > 1e8) S.status = 2;
    return true;
}

// After convergence: decide admissibility the way Optimizer._solve_n3plus does (all nu_j in [0,1],
// Optimizer.py:150-160).  For rank-deficient candidates the minimiser is a line; the reference
// accepts when its root finder happens to land inside the simplex, so the line is intersected with
// the simplex and a point inside is taken when one exists.  Returns true if admissible.
__device__ __forceinline__ bool n3_admissible(N3Hess &S, double s1, double s2) {
    double n1 = s1 * S.u1, n2 = s2 * S.u2, n0 = 1.0 - n1 - n2;
    bool in = (n0 >= 0.0 && n0 <= 1.0 && n1 >= 0.0 && n1 <= 1.0 && n2 >= 0.0 && n2 <= 1.0);
    if (in) return true;
    double det0 = S.h11 * S.h22 - S.h12 * S.h12;
    if (!(det0 <= 1e-10 * S.h11 * S.h22)) return false;   // regular optimum outside the simplex: None
    // null direction of H
    double v1, v2;
    if (S.h11 >= S.h22) { v1 = -S.h12; v2 = S.h11; } else { v1 = S.h22; v2 = -S.h12; }
    double nrm = sqrt(v1 * v1 + v2 * v2);
    if (!(nrm > 0.0)) return false;
    v1 /= nrm; v2 /= nrm;
    const double d0 = -(s1 * v1 + s2 * v2), d1 = s1 * v1, d2 = s2 * v2;
    double slo = -1e300, shi = 1e300;
    bool dead = false;
    auto clip = [&](double nu, double dn) {   // 0 <= nu + s dn <= 1
        if (dn > 0) { slo = fmax(slo, -nu / dn); shi = fmin(shi, (1.0 - nu) / dn); }
        else if (dn < 0) { slo = fmax(slo, (1.0 - nu) / dn); shi = fmin(shi, -nu / dn); }
        else if (nu < 0.0 || nu > 1.0) dead = true;
    };
    clip(n0, d0);
    clip(n1, d1);
    clip(n2, d2);
    if (dead) return false;
    if (!(slo <= shi)) return false;
    double s = 0.5 * (slo + shi);
    S.u1 += s * v1;
    S.u2 += s * v2;
    return true;
}
#endif
