// n = 3 fused search for gfx950.
//
// Reference operators replaced (file:line into the reference's python/):
//   Enumerator._create_graph, _generate_next_C_3(_recurse), _in_bounds, _get_mu_bounds
//                                                         Enumerator.py:172-298
//   Optimizer._solve_n3plus + equations/jacobian/M3/L3    Optimizer.py:128-165, 236-330
//   the running minimum of do_optimization_single          RunTHetA.py:191-208
//
// Mapping to the hardware.  One wavefront owns one contiguous rank range of the reference's DFS
// order.  The first D = m - L rows of the matrix (the "prefix") are wave-uniform and live one
// interval per lane (lane i holds r_i, rN_i, bounds and the DFS state of depth i); the last L rows
// are enumerated by the 64 lanes.  Feasible leaves are compacted through an LDS queue so the solver
// always runs on full waves.  Because the prefix is shared, intervals with the same (a, b) row
// collapse into one likelihood term: the wave builds an LDS "group tile" {a, b, sum r} of the
// prefix once per prefix and every lane's Newton iteration streams that tile (LDS broadcast reads)
// plus its own L rows.  Nothing but the tie records ever goes to HBM.
#include <algorithm>

#include "n3_core.hpp"

// ------------------------------------------------------------------------------------------------
// host: bounds, ratio table
// ------------------------------------------------------------------------------------------------
int n3_build_host(int m, const int32_t *lb_in, const int32_t *ub_in, N3Host &h) {
    h.m = m;
    h.lb.assign(lb_in, lb_in + m);
    h.ub.assign(ub_in, ub_in + m);
    for (int i = 1; i < m; i++)
        if (h.lb[i] < h.lb[i - 1]) h.lb[i] = h.lb[i - 1];
    for (int i = m - 2; i >= 0; i--)
        if (h.ub[i] > h.ub[i + 1]) h.ub[i] = h.ub[i + 1];
    int top = 0;
    for (int i = 0; i < m; i++) {
        if (h.lb[i] < 0) {
            theta_set_error("negative lower bound at interval %d", i);
            return THETA_ERR_ARG;
        }
        top = std::max(top, h.ub[i]);
    }
    if (top > N3_MAX_K) {
        theta_set_error("n=3 search supports copy numbers up to %d (max upper bound is %d)", N3_MAX_K, top);
        return THETA_ERR_ARG;
    }
    h.K = top;                    // Enumerator.py:58: k = max(upper_bound)
    h.Q = (top + 1) * (top + 1);
    // distinct values of dy/(-dx), dx,dy in +-[1..K], sorted ascending; compared by cross-multiplication
    struct Fr { int n, d; };
    std::vector<Fr> fr;
    for (int dx = -top; dx <= top; dx++)
        for (int dy = -top; dy <= top; dy++) {
            if (dx == 0 || dy == 0) continue;
            int n = dy, d = -dx;
            if (d < 0) { n = -n; d = -d; }
            fr.push_back({n, d});
        }
    auto less = [](const Fr &a, const Fr &b) { return a.n * b.d < b.n * a.d; };
    auto eq = [](const Fr &a, const Fr &b) { return a.n * b.d == b.n * a.d; };
    std::sort(fr.begin(), fr.end(), less);
    fr.erase(std::unique(fr.begin(), fr.end(), eq), fr.end());
    h.NT = (int)fr.size();
    if (h.NT > 253) {
        theta_set_error("ratio table too large");
        return THETA_ERR_ARG;
    }
    h.ridx.assign(N3_RIDX_W * N3_RIDX_W, 0);
    for (int dx = -top; dx <= top; dx++)
        for (int dy = -top; dy <= top; dy++) {
            if (dx == 0 || dy == 0) continue;
            Fr v{dy, -dx};
            if (v.d < 0) { v.n = -v.n; v.d = -v.d; }
            int idx = (int)(std::lower_bound(fr.begin(), fr.end(), v, less) - fr.begin());
            h.ridx[(dy + N3_MAX_K) * N3_RIDX_W + (dx + N3_MAX_K)] = (unsigned char)(idx + 1);
        }
    return THETA_OK;
}

// ------------------------------------------------------------------------------------------------
// counting DP (exact number of matrices below every DFS node), one launch per depth
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void n3_dp_kernel(N3Dev P, u128 *cnt, int d, unsigned *overflow) {
    size_t per_level = (size_t)P.Q * 2 * (P.NT + 1) * (P.NT + 1);
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_level) return;
    int NT1 = P.NT + 1;
    int hi = (int)(idx % NT1) + 1;
    int lo = (int)((idx / NT1) % NT1);
    int sw = (int)((idx / ((size_t)NT1 * NT1)) % 2);
    int slot = (int)(idx / ((size_t)NT1 * NT1 * 2));
    u128 sum = 0;
    if (lo <= hi) {
        if (d == P.m - 1) {
            sum = 1;
        } else {
            N3State par{slot, sw, lo, hi}, ch;
            for (int c = 0; c < P.Q; c++) {
                if (n3_edge(P, par, c, d + 1, ch)) {
                    u128 v = cnt[n3_cnt_index(P, d + 1, ch.slot, ch.sw, ch.lo, ch.hi)];
                    u128 ns = sum + v;
                    if (ns < sum) atomicOr(overflow, 1u);
                    sum = ns;
                }
            }
        }
    }
    cnt[(size_t)d * per_level + idx] = sum;
}

__global__ void n3_total_kernel(N3Dev P, unsigned long long *total, unsigned *overflow) {
    u128 sum = 0;
    N3State s;
    for (int c = 0; c < P.Q; c++)
        if (n3_first_row(P, c, s)) {
            u128 v = P.cnt[n3_cnt_index(P, 0, s.slot, s.sw, s.lo, s.hi)];
            u128 ns = sum + v;
            if (ns < sum) atomicOr(overflow, 1u);
            sum = ns;
        }
    total[0] = (unsigned long long)sum;
    total[1] = (unsigned long long)(sum >> 64);
}

// ------------------------------------------------------------------------------------------------
// rank -> DFS path (serial; used for task starts, tie records and the materialised generator)
// ------------------------------------------------------------------------------------------------
__device__ bool n3_unrank(const N3Dev &P, u128 rho, int depth, N3State *st, u128 &rem) {
    bool found = false;
    N3State s;
    for (int c = 0; c < P.Q && !found; c++) {
        if (!n3_first_row(P, c, s)) continue;
        u128 v = P.cnt[n3_cnt_index(P, 0, s.slot, s.sw, s.lo, s.hi)];
        if (rho < v) { st[0] = s; found = true; } else rho -= v;
    }
    if (!found) return false;
    for (int d = 1; d < depth; d++) {
        found = false;
        for (int c = 0; c < P.Q && !found; c++) {
            if (!n3_edge(P, st[d - 1], c, d, s)) continue;
            u128 v = P.cnt[n3_cnt_index(P, d, s.slot, s.sw, s.lo, s.hi)];
            if (rho < v) { st[d] = s; found = true; } else rho -= v;
        }
        if (!found) return false;
    }
    rem = rho;
    return true;
}

__device__ __forceinline__ unsigned n3_pack(const N3State &s) {
    return (unsigned)s.slot | ((unsigned)s.sw << 8) | ((unsigned)s.lo << 16) | ((unsigned)s.hi << 24);
}
__device__ __forceinline__ N3State n3_unpack(unsigned v) {
    N3State s;
    s.slot = v & 0xff;
    s.sw = (v >> 8) & 1;
    s.lo = (v >> 16) & 0xff;
    s.hi = (v >> 24) & 0xff;
    return s;
}

// one thread per task: where does the task start?
__global__ __launch_bounds__(64) void n3_task_kernel(N3Dev P, uint64_t b_lo, uint64_t b_hi, uint64_t e_lo, uint64_t e_hi,
                                                     uint64_t per_task, int ntasks, N3Task *tasks, unsigned *stbuf) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntasks) return;
    const u128 begin = ((u128)b_hi << 64) | b_lo, end = ((u128)e_hi << 64) | e_lo;
    u128 base = begin + (u128)t * per_task;
    u128 left = end - base;
    uint64_t count = left < (u128)per_task ? (uint64_t)left : per_task;
    int D = P.m - P.L;
    N3State st[N3_MAX_M];
    u128 rem = 0;
    bool ok = n3_unrank(P, base, D, st, rem);
    N3Task tk;
    tk.base_lo = (uint64_t)base;
    tk.base_hi = (uint64_t)(base >> 64);
    tk.count = ok ? count : 0;
    tk.skip = (uint64_t)rem;
    tasks[t] = tk;
    for (int d = 0; d < D; d++) stbuf[(size_t)t * N3_MAX_M + d] = n3_pack(st[d]);
}

__global__ __launch_bounds__(64) void n3_unrank_list_kernel(N3Dev P, const TieRecord *recs, int count, unsigned char *out) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    N3State st[N3_MAX_M];
    u128 rho = ((u128)recs[k].rank_hi << 64) | recs[k].rank_lo, rem;
    bool ok = n3_unrank(P, rho, P.m, st, rem);
    unsigned char *dst = out + (size_t)k * P.m * 2;
    int K1 = P.K + 1;
    for (int i = 0; i < P.m; i++) {
        dst[2 * i] = ok ? (unsigned char)(st[i].slot % K1) : 255;
        dst[2 * i + 1] = ok ? (unsigned char)(st[i].slot / K1) : 255;
    }
}

__global__ __launch_bounds__(64) void n3_enumerate_kernel(N3Dev P, uint64_t b_lo, uint64_t b_hi, unsigned long long count,
                                                          unsigned char *out) {
    unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    const u128 begin = ((u128)b_hi << 64) | b_lo;
    N3State st[N3_MAX_M];
    u128 rem;
    bool ok = n3_unrank(P, begin + k, P.m, st, rem);
    unsigned char *dst = out + (size_t)k * P.m * 2;
    int K1 = P.K + 1;
    for (int i = 0; i < P.m; i++) {
        dst[2 * i] = ok ? (unsigned char)(st[i].slot % K1) : 255;
        dst[2 * i + 1] = ok ? (unsigned char)(st[i].slot / K1) : 255;
    }
}

// ------------------------------------------------------------------------------------------------
// the fused search kernel
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

#define N3_WAVES 4
#define N3_QCAP 128

struct N3Lds {
    double gX[N3_WAVES][N3_MAX_Q + 1], gY[N3_WAVES][N3_MAX_Q + 1], gR[N3_WAVES][N3_MAX_Q + 1];
    unsigned qCode[N3_WAVES][N3_QCAP], qOff[N3_WAVES][N3_QCAP];
    unsigned char lb[N3_MAX_M], ub[N3_MAX_M];
    unsigned char ridx[N3_RIDX_W * N3_RIDX_W + 3];
};

template <int L, bool DUMP>
__global__ __launch_bounds__(64 * N3_WAVES) void n3_search_kernel(N3Dev Pg, SearchArgs A, const N3Task *tasks,
                                                                const unsigned *stbuf, int ntasks, uint64_t per_task) {
    __shared__ N3Lds S;
    // stage what every wave of the block shares
    for (int i = threadIdx.x; i < Pg.m; i += blockDim.x) {
        S.lb[i] = Pg.lb[i];
        S.ub[i] = Pg.ub[i];
    }
    for (int i = threadIdx.x; i < N3_RIDX_W * N3_RIDX_W; i += blockDim.x) S.ridx[i] = Pg.ridx[i];
    __syncthreads();
    N3Dev P = Pg;
    P.lb = S.lb;
    P.ub = S.ub;
    P.ridx = S.ridx;

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * N3_WAVES + wv;
    if (task >= ntasks) return;  // whole wave leaves together; no block barrier below
    const int m = P.m, D = m - L, Q = P.Q, K1 = P.K + 1;
    const double tau = (double)P.tau;
    double *gX = S.gX[wv], *gY = S.gY[wv], *gR = S.gR[wv];
    unsigned *qCode = S.qCode[wv], *qOff = S.qOff[wv];

    // lane i holds interval i
    const double r_i = lane < m ? Pg.r[lane] : 0.0;
    const double rN_i = lane < m ? Pg.rN[lane] : 0.0;
    unsigned st = lane < D ? stbuf[(size_t)task * N3_MAX_M + lane] : 0u;

    const N3Task tk = tasks[task];
    const u128 base = ((u128)tk.base_hi << 64) | tk.base_lo;
    unsigned long long remaining = tk.count, skip = tk.skip, processed = 0;
    const unsigned long long dump_base = (unsigned long long)task * per_task;  // position of the task in the dump arrays

    // leaf rows' shared data (wave-uniform)
    double leafR[L], leafN[L];
#pragma unroll
    for (int l = 0; l < L; l++) {
        leafR[l] = readlane_f64(r_i, D + l);
        leafN[l] = readlane_f64(rN_i, D + l);
    }
    int QL = 1;
#pragma unroll
    for (int l = 0; l < L; l++) QL *= Q;

    unsigned long long n_eval = 0, n_acc = 0, n_deg = 0, n_it = 0, n_terms = 0, n_fin = 0;
    double best = order_unbits(load_agent_u64(&A.ctr->best_bits));
    double rej_best = order_unbits(load_agent_u64(&A.ctr->rej_bits));

    while (remaining > 0) {
        // ---------------- group tile of the prefix --------------------------------------------
        int G = 0;
        double S1p = 0.0, S2p = 0.0;
        {
            const bool inp = lane < D;
            const int myslot = st & 0xff;
            unsigned long long todo = ballot64(inp);
            while (todo) {
                int leader = __builtin_ctzll(todo);
                int q = __builtin_amdgcn_readlane(myslot, leader);
                bool match = inp && myslot == q;
                todo &= ~ballot64(match);
                double Rs = match ? r_i : 0.0, Ns = match ? rN_i : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {  // integer-valued doubles < 2^53: the sums are exact
                    Rs += __shfl_xor(Rs, o, WAVE);
                    Ns += __shfl_xor(Ns, o, WAVE);
                }
                double a = (double)(q % K1), b = (double)(q / K1);
                if (lane == 0) {
                    gX[G] = a;
                    gY[G] = b;
                    gR[G] = Rs;
                }
                S1p += a * Ns;
                S2p += b * Ns;
                G++;
            }
        }
        wave_lds_sync();

        // ---------------- scan the L leaf levels, compact, solve ------------------------------
        const N3State par = n3_unpack((unsigned)__builtin_amdgcn_readlane((int)st, D - 1));
        unsigned long long leaf_idx = 0;
        int qcount = 0;

        auto process = [&](int cnt) {
            const bool have = lane < cnt;
            unsigned code = have ? qCode[lane] : 0u;
            const unsigned long long rel = processed + (have ? qOff[lane] : 0u);
            double lx[L], ly[L];
            {
                unsigned cc = code;
#pragma unroll
                for (int l = L - 1; l >= 0; l--) {
                    int s = cc % Q;
                    cc /= Q;
                    lx[l] = (double)(s % K1);
                    ly[l] = (double)(s / K1);
                }
            }
            double S1 = S1p, S2 = S2p;
#pragma unroll
            for (int l = 0; l < L; l++) {
                S1 = __builtin_fma(lx[l], leafN[l], S1);
                S2 = __builtin_fma(ly[l], leafN[l], S2);
            }
            const bool degenerate = have && (S1 == 0.0 || S2 == 0.0);
            const double s1 = S1 / P.N, s2 = S2 / P.N;
            auto terms = [&](auto &&body) {
                for (int g = 0; g < G; g++) body(gX[g], gY[g], gR[g]);
#pragma unroll
                for (int l = 0; l < L; l++) body(lx[l], ly[l], leafR[l]);
            };
            N3Newton Sv;
            Sv.u1 = (1.0 / 3.0) / s1;  // nu = (1/3,1/3,1/3): the reference's start (Optimizer.py:147)
            Sv.u2 = (1.0 / 3.0) / s2;
            Sv.p1 = Sv.u1; Sv.p2 = Sv.u2;
            Sv.h11 = Sv.h12 = Sv.h22 = Sv.g1 = Sv.g2 = 0.0;
            Sv.lam = 0.0;
            Sv.iters = 0;
            Sv.status = 0;
            bool run = have && !degenerate;
            while (ballot64(run)) {
                if (run) {
                    n3_newton_step(terms, s1, s2, P.Rtot, Sv);
                    run = (Sv.status == 0);
                }
            }
            // ---- value at the optimum / lower bound for rejected candidates
            bool solved = have && !degenerate;
            bool conv = solved && Sv.status == 1;
            bool accept = conv && n3_admissible(Sv, s1, s2);
            double eu1 = conv ? Sv.u1 : Sv.p1, eu2 = conv ? Sv.u2 : Sv.p2;   // non-converged: last feasible iterate
            double acc = 0.0, g1 = 0.0, g2 = 0.0;
            if (solved) {
                terms([&](double x, double y, double R) {
                    double a = x - s1, b = y - s2;
                    double q = __builtin_fma(a, eu1, __builtin_fma(b, eu2, 1.0));
                    acc = __builtin_fma(R, log(q), acc);
                    double t = R / q;
                    g1 = __builtin_fma(t, a, g1);
                    g2 = __builtin_fma(t, b, g2);
                });
            }
            double nll = P.K0 - acc;
            // Frank-Wolfe bound: NLL(z) >= NLL(u) + grad.(z - u) for every z in the simplex; minimum at a vertex
            double lbnd = nll;
            if (solved && !accept) {
                double e0 = g1 * eu1 + g2 * eu2;                 // -grad.(v0 - u), v0 = (0,0)
                double e1 = e0 - g1 / s1, e2 = e0 - g2 / s2;     // v1 = (1/s1,0), v2 = (0,1/s2)
                lbnd = nll + fmin(e0, fmin(e1, e2));
            }
            double u0 = (1.0 - s1 * Sv.u1 - s2 * Sv.u2) / tau;
            double usum = u0 + Sv.u1 + Sv.u2;
            double mu0 = u0 / usum, mu1 = Sv.u1 / usum, mu2 = Sv.u2 / usum;  // closed form of M3 (Optimizer.py:318-330)

            if (accept && nll <= best + A.window) {
                best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
                if (nll <= best + A.window) {
                    tie_append(A.ctr, A.list, A.list_cap, base + rel, nll, mu0, mu1, mu2);
                    if (nll < best) atomicMin(&A.ctr->best_bits, order_bits(nll));
                }
            }
            double wbest = wave_min(accept ? nll : __builtin_inf());
            best = fmin(best, wbest);
            if (solved && !accept && lbnd < rej_best) {
                unsigned long long old = atomicMin(&A.ctr->rej_bits, order_bits(lbnd));
                if (old > order_bits(lbnd)) {  // we hold the minimum (racy pair, diagnostic only)
                    u128 rk = base + rel;
                    A.ctr->rej_rank_lo = (unsigned long long)rk;
                    A.ctr->rej_rank_hi = (unsigned long long)(rk >> 64);
                }
                rej_best = lbnd;
            }
            if (DUMP && have) {
                double nan = __builtin_nan("");
                const unsigned long long di = dump_base + rel;
                A.dump_nll[di] = accept ? nll : nan;
                A.dump_mu[di * 3 + 0] = accept ? mu0 : nan;
                A.dump_mu[di * 3 + 1] = accept ? mu1 : nan;
                A.dump_mu[di * 3 + 2] = accept ? mu2 : nan;
            }
            n_eval += have;
            n_acc += accept;
            n_deg += degenerate;
            n_it += solved ? Sv.iters : 0;
            n_terms += solved ? (unsigned long long)Sv.iters * (G + L) : 0;
            n_fin += solved ? (G + L) : 0;
        };

        for (int cbase = 0; cbase < QL; cbase += WAVE) {
            int code = cbase + lane;
            bool feas = code < QL;
            if (feas) {
                int dig[L];
                int cc = code;
#pragma unroll
                for (int l = L - 1; l >= 0; l--) {
                    dig[l] = cc % Q;
                    cc /= Q;
                }
                N3State cur = par, nx;
#pragma unroll
                for (int l = 0; l < L; l++) {
                    if (feas) {
                        feas = n3_edge(P, cur, dig[l], D + l, nx);
                        cur = nx;
                    }
                }
            }
            unsigned long long mask = ballot64(feas);
            if (!mask) continue;
            unsigned long long off = leaf_idx + mbcnt(mask);
            leaf_idx += __builtin_popcountll(mask);
            bool sel = feas && off >= skip && (off - skip) < remaining;
            unsigned long long smask = ballot64(sel);
            if (smask) {
                int pos = qcount + mbcnt(smask);
                if (sel) {
                    qCode[pos] = (unsigned)code;
                    qOff[pos] = (unsigned)(off - skip);
                }
                qcount += __builtin_popcountll(smask);
                wave_lds_sync();
                if (qcount >= WAVE) {
                    process(WAVE);
                    // move the tail of the queue to the front
                    unsigned c2 = 0, o2 = 0;
                    bool mv = WAVE + lane < qcount;
                    if (mv) {
                        c2 = qCode[WAVE + lane];
                        o2 = qOff[WAVE + lane];
                    }
                    wave_lds_sync();
                    if (mv) {
                        qCode[lane] = c2;
                        qOff[lane] = o2;
                    }
                    qcount -= WAVE;
                    wave_lds_sync();
                }
            }
            if (leaf_idx >= skip + remaining) break;  // the task's quota ends inside this subtree
        }
        if (qcount > 0) {
            process(qcount);
            qcount = 0;
        }
        unsigned long long consumed = 0;
        if (leaf_idx > skip) {
            consumed = leaf_idx - skip;
            if (consumed > remaining) consumed = remaining;
            skip = 0;
        } else {
            skip -= leaf_idx;
        }
        processed += consumed;
        remaining -= consumed;
        if (remaining == 0) break;

        // ---------------- next prefix in DFS order (wave-uniform) -----------------------------
        {
            int d = D - 1;
            bool fresh = false, alive = true;
            while (true) {
                int cur_slot = __builtin_amdgcn_readlane((int)st, d) & 0xff;
                int start = fresh ? 0 : cur_slot + 1;
                N3State pst = n3_unpack((unsigned)__builtin_amdgcn_readlane((int)st, d > 0 ? d - 1 : 0));
                bool found = false;
                unsigned packed = 0;
                for (int cb = (start / WAVE) * WAVE; cb < Q && !found; cb += WAVE) {
                    int s = cb + lane;
                    N3State nx;
                    bool ok = s >= start && s < Q && (d == 0 ? n3_first_row(P, s, nx) : n3_edge(P, pst, s, d, nx));
                    unsigned long long mk = ballot64(ok);
                    if (mk) {
                        int first = __builtin_ctzll(mk);
                        unsigned mine = ok ? n3_pack(nx) : 0u;
                        packed = (unsigned)__builtin_amdgcn_readlane((int)mine, first);
                        found = true;
                    }
                }
                if (found) {
                    if (lane == d) st = packed;
                    if (d == D - 1) break;
                    d++;
                    fresh = true;
                } else {
                    d--;
                    fresh = false;
                    if (d < 0) {
                        alive = false;
                        break;
                    }
                }
            }
            if (!alive) break;  // end of the enumeration (cannot happen inside a valid rank range)
        }
    }

    n_eval = wave_sum_u64(n_eval);
    n_acc = wave_sum_u64(n_acc);
    n_deg = wave_sum_u64(n_deg);
    n_it = wave_sum_u64(n_it);
    n_terms = wave_sum_u64(n_terms);
    n_fin = wave_sum_u64(n_fin);
    if (lane == 0) {
        atomicAdd(&A.ctr->evaluated, n_eval);
        atomicAdd(&A.ctr->accepted, n_acc);
        atomicAdd(&A.ctr->degenerate, n_deg);
        atomicAdd(&A.ctr->iterations, n_it);
        atomicAdd(&A.ctr->terms, n_terms);
        atomicAdd(&A.ctr->final_terms, n_fin);
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
int n3_run_dp(const N3Dev &P, u128 *cnt, unsigned *overflow_dev, unsigned long long *total_dev, hipStream_t st) {
    size_t per_level = (size_t)P.Q * 2 * (P.NT + 1) * (P.NT + 1);
    unsigned blocks = (unsigned)((per_level + 255) / 256);
    for (int d = P.m - 1; d >= 0; d--) hipLaunchKernelGGL(n3_dp_kernel, dim3(blocks), dim3(256), 0, st, P, cnt, d, overflow_dev);
    hipLaunchKernelGGL(n3_total_kernel, dim3(1), dim3(1), 0, st, P, total_dev, overflow_dev);
    return THETA_OK;
}

void n3_launch_tasks(const N3Dev &P, u128 begin, u128 end, uint64_t per_task, int ntasks, N3Task *tasks,
                     unsigned *stbuf, hipStream_t st) {
    hipLaunchKernelGGL(n3_task_kernel, dim3((ntasks + 63) / 64), dim3(64), 0, st, P, (uint64_t)begin, (uint64_t)(begin >> 64),
                       (uint64_t)end, (uint64_t)(end >> 64), per_task, ntasks, tasks, stbuf);
}

void n3_launch_search(const N3Dev &P, const SearchArgs &A, const N3Task *tasks, const unsigned *stbuf, int ntasks,
                      uint64_t per_task, hipStream_t st) {
    dim3 grid((ntasks + N3_WAVES - 1) / N3_WAVES), block(64 * N3_WAVES);
    bool dump = A.dump_nll != nullptr;
#define LAUNCH(LL)                                                                                       \
    if (dump) hipLaunchKernelGGL((n3_search_kernel<LL, true>), grid, block, 0, st, P, A, tasks, stbuf, ntasks, per_task); \
    else hipLaunchKernelGGL((n3_search_kernel<LL, false>), grid, block, 0, st, P, A, tasks, stbuf, ntasks, per_task);
    if (P.L == 1) { LAUNCH(1) } else if (P.L == 2) { LAUNCH(2) } else { LAUNCH(3) }
#undef LAUNCH
}

void n3_launch_unrank_list(const N3Dev &P, const TieRecord *recs, int count, unsigned char *out, hipStream_t st) {
    hipLaunchKernelGGL(n3_unrank_list_kernel, dim3((count + 63) / 64), dim3(64), 0, st, P, recs, count, out);
}

void n3_launch_enumerate(const N3Dev &P, u128 begin, unsigned long long count, unsigned char *out, hipStream_t st) {
    hipLaunchKernelGGL(n3_enumerate_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, P, (uint64_t)begin,
                       (uint64_t)(begin >> 64), count, out);
}
