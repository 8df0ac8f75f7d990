// Branch and bound ABOVE the sieve's prefix: the exact arg-min of an n = 3 space no linear walk finishes.
//
// Reference path replaced: the loop of do_optimization_single (python/RunTHetA.py:173-220) over Enumerator._generate_next_C_3
// (Enumerator.py:172-214) -- for spaces like BASELINE config 3 / 4 (m = 50: 4e27 / 2.6e38 matrices), which that loop (and any
// kernel that visits every rank) can only sample.
//
// The relaxation behind n3_sieve.hip's sv_prefix_beyond holds for ANY number of fixed rows.  With rows 0 .. d-1 of a matrix
// fixed and every later interval l fitted perfectly (its term q_l a free t_l > 0, minimised out in closed form),
//      min over the completions of NLL  >=  min_w [K0 - sum' R_g ln q_g(w) + R' ln(z''.w)] + const_d,
//      const_d = R' ln(om) + R' ln(Rtot / R') - sum_{l >= d} r_l ln(r_l / (Rtot N_l)),
// sum' over the DISTINCT rows g of the prefix (R_g, N_g: tumour / normal counts of the intervals holding row g), R' = sum' R_g,
// om = sum' N_g, z'' = (1, s1, s2) the prefix's column sums over om.  The bracket is the likelihood of the prefix alone: a convex
// two-parameter problem over at most 64 terms, the kind the sieve solves all the time (tests/test_prefix_bound_cpu.py checks
// the inequality against the minimum of every completion).  So the tree Enumerator._generate_next_C_3 walks depth first is
// walked here LEVEL BY LEVEL from the root: one wave per frontier node, one lane per alphabet slot = per child; a child's
// rows are the reference's (same row graph, symmetry switch and ratio window: n3_edge_ab), its subtree's size comes from the
// counting table -- so every surviving node IS a contiguous range of the reference's ranks --, and its bound is a damped
// Newton solve from its parent's optimum with the self-concordance lower bound of the sieve (sv_beyond).  Children whose
// bound lies beyond the threshold (an attainable NLL + the collection window) are dropped with everything below them; the
// survivors of the emit depth (the sieve's prefix depth m - 6, or any node with few matrices left) are handed to theta_search
// as rank ranges, where the sieve's own bounds finish the job.  The host driver (api.hip: theta_bnb) walks the levels depth
// first in chunks when a level outgrows its buffer, so memory is bounded by depth x chunk x alphabet.
#include "bnb.hpp"
#include "smx_log.hpp"            // the table-driven FP64 logarithm of the scorers (tests/test_smx_log_cpu.py)

struct BnbWave {
    double binR[N3_MAX_Q], binN[N3_MAX_Q];      // tumour / normal counts of the prefix's intervals, per alphabet slot
    double gx[N3_MAX_Q], gy[N3_MAX_Q], gR[N3_MAX_Q];   // the distinct rows of the prefix, dense: row and tumour weight
    int gs[N3_MAX_Q];                           // ... and their slots
    unsigned path[N3_MAX_M_WIDE / 4];           // the parent's path (bytes), for the children's copies
};

__host__ __device__ inline unsigned bnb_line_pack(const N3Line &s) {
    return (unsigned)s.kind | ((unsigned)s.a0 << 2) | ((unsigned)s.b0 << 6) | ((unsigned)(s.da + 16) << 10) | ((unsigned)(s.db + 16) << 15);
}
__host__ __device__ inline N3Line bnb_line_unpack(unsigned v) {
    N3Line s;
    s.kind = (int)(v & 3u);
    s.a0 = (int)((v >> 2) & 15u);
    s.b0 = (int)((v >> 6) & 15u);
    s.da = (int)((v >> 10) & 31u) - 16;
    s.db = (int)((v >> 15) & 31u) - 16;
    return s;
}

__device__ __forceinline__ double bnb_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ unsigned long long bnb_wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

#define BNB_WAVES 4
#define BNB_MAXIT 64

__global__ __launch_bounds__(64 * BNB_WAVES) void bnb_expand_kernel(N3Dev P, BnbArgs A) {
    __shared__ BnbWave S[BNB_WAVES];
    __shared__ unsigned blk_next[BNB_WAVES], blk_emit[BNB_WAVES];
    __shared__ unsigned long long blk_base[2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned node = blockIdx.x * BNB_WAVES + wv;
    const bool live_wave = node < A.n_in;
    BnbWave &W = S[wv];
    const int d = A.d, Q = P.Q;

    bool ok = false, keep = false, emit = false;
    u128 cnt = 0, mybase = 0;
    double bound = -__builtin_inf(), o_w0 = __builtin_nan(""), o_u1 = 0.0, o_u2 = 0.0;
    N3State nx{0, 0, 0, 0, 0, 0};
    N3Line ln{0, 0, 0, 0, 0};
    unsigned long long st_solves = 0, st_iters = 0, st_pruned = 0, st_line = 0, st_open = 0, st_why5 = 0, st_why6 = 0, st_why7 = 0;

    if (live_wave) {
        const BnbNode pn = A.in[node];
        const unsigned char *pp = A.in_path + (size_t)node * A.path_stride;
        // ---- the prefix's intervals by row: exact sums (integer-valued doubles below 2^53, any order)
        W.binR[lane] = 0.0;
        W.binN[lane] = 0.0;
        for (int i = lane; i < A.path_stride / 4; i += WAVE) W.path[i] = ((const unsigned *)pp)[i];
        wave_lds_sync();
        for (int i = lane; i < d; i += WAVE) {
            const unsigned s = pp[i];
            atomicAdd(&W.binR[s], P.r[i]);
            atomicAdd(&W.binN[s], P.rN[i]);
        }
        wave_lds_sync();
        const double invN = 1.0 / P.N;
        const unsigned myrow = lane < Q ? P.rowtab[lane] : 0u;
        const int sa = (int)(myrow & 15u), sb = (int)(myrow >> 4);
        const bool has = lane < Q && W.binN[lane] > 0.0;
        const unsigned long long hm = ballot64(has);
        const int G = __builtin_popcountll(hm);
        if (has) {
            const int idx = mbcnt(hm);
            W.gx[idx] = (double)sa;
            W.gy[idx] = (double)sb;
            W.gR[idx] = W.binR[lane];
            W.gs[idx] = lane;
        }
        const double Ns = has ? W.binN[lane] * invN : 0.0;
        const double Z1 = bnb_wave_sum(Ns * (double)sa), Z2 = bnb_wave_sum(Ns * (double)sb);
        wave_lds_sync();

        // ---- the children: the reference's edges (Enumerator.py:192-212), their subtrees' sizes, their first ranks
        const N3State par = n3_unpack(pn.state);
        ok = lane < Q && (d == 0 ? n3_first_row_ab(P, sa, sb, lane, nx) : n3_edge_ab(P, par, sa, sb, lane, d, nx));
        if (ok) cnt = P.cnt[n3_cnt_index(P, d, nx.slot, nx.sw, nx.lo, nx.hi)];
        {
            const unsigned v0 = (unsigned)cnt, v1 = (unsigned)(cnt >> 32), v2 = (unsigned)(cnt >> 64), v3 = (unsigned)(cnt >> 96);
            unsigned long long mk = ballot64(ok);
            u128 run = ((u128)pn.base_hi << 64) | pn.base_lo;
            while (mk) {
                const int b = __builtin_ctzll(mk);
                mk &= mk - 1;
                const u128 vb = ((u128)(unsigned)__builtin_amdgcn_readlane((int)v3, b) << 96) | ((u128)(unsigned)__builtin_amdgcn_readlane((int)v2, b) << 64) |
                                ((u128)(unsigned)__builtin_amdgcn_readlane((int)v1, b) << 32) | (u128)(unsigned)__builtin_amdgcn_readlane((int)v0, b);
                if (lane == b) mybase = run;
                run += vb;
            }
        }
        ln = bnb_line_unpack(pn.line);
        if (ok) n3_line_add(ln, sa, sb);
        const bool collinear = ln.kind < 3;
        keep = ok;

        // ---- the child's bound: Newton on the likelihood of its d + 1 rows alone, from the parent's optimum
        if (ok && (collinear && A.follow_line)) {
            st_line = 1;
        } else if (ok && ln.kind <= 1) {
            // every row so far is the same row: q = 1 whatever the mixture, the relaxed problem is the constant K0 + const_d
            // (a Newton iteration on its rounding residue would only run away)
            bound = P.K0 + A.constc - 1e-3 - 1e-12 * fabs(P.K0);
            if (!A.full_bound && bound > A.thr) {
                keep = false;
                st_pruned = 1;
            }
        } else if (ok) {
            st_solves = 1;
            const double z0 = A.Z0;
            const double s1 = (Z1 + A.nd * (double)sa) / z0, s2 = (Z2 + A.nd * (double)sb) / z0;
            const double xs = (double)sa, ys = (double)sb;
            double u1 = 0.0, u2 = 0.0, pu1 = 0.0, pu2 = 0.0;       // the iterate; the last point known to lie in the domain (the centre does)
            double pval = -__builtin_inf();                         // sum R ln q at that point (-inf: none evaluated yet)
            bool warm = false;
            if (pn.w0 == pn.w0) {
                const double zw = pn.w0 + s1 * pn.u1 + s2 * pn.u2;
                if (zw > 0.0 && zw < 1e300) {
                    u1 = pn.u1 / zw;
                    u2 = pn.u2 / zw;
                    warm = true;
                }
            }
            bool decided = false;
            for (int it = 0; it < BNB_MAXIT && !decided; it++) {
                double val = 0.0, g1 = 0.0, g2 = 0.0, h11 = 0.0, h12 = 0.0, h22 = 0.0, rmin = __builtin_inf(), rsum = 0.0;
                bool bad = false;
                auto term = [&](double x, double y, double R) {
                    if (!(R > 0.0)) return;                       // (intervals without tumour reads weigh nothing)
                    const double a = x - s1, b = y - s2;
                    const double q = __builtin_fma(a, u1, __builtin_fma(b, u2, 1.0));
                    bad |= !(q > 0.0);
                    const double qq = q > 0.0 ? q : 1.0, w = 1.0 / qq, t = R * w, tw = t * w;
                    val = __builtin_fma(R, log(qq), val);
                    g1 = __builtin_fma(t, a, g1);
                    g2 = __builtin_fma(t, b, g2);
                    h11 = __builtin_fma(tw * a, a, h11);
                    h12 = __builtin_fma(tw * a, b, h12);
                    h22 = __builtin_fma(tw * b, b, h22);
                    rmin = fmin(rmin, R);
                    rsum += R;
                };
                for (int g = 0; g < G; g++) term(W.gx[g], W.gy[g], W.gR[g] + (W.gs[g] == lane ? A.rd : 0.0));
                if (!has) term(xs, ys, A.rd);
                st_iters++;
                // A step that leaves the child's domain, or does not lower the (convex) objective -- an overshoot from a point close
                // to the domain's boundary, where a child's warm start may well lie --, is halved back towards the point it left: the
                // values descend monotonically, so the iteration cannot cycle.  (A warm start outside the domain gives way to the
                // centre of the slice.)
                if (bad || (pval > -__builtin_inf() && val < pval - 1e-9 * fabs(pval) - 1e-9)) {
                    if (bad && warm) {
                        u1 = u2 = pu1 = pu2 = 0.0;
                        warm = false;
                    } else {
                        u1 = 0.5 * (u1 + pu1);
                        u2 = 0.5 * (u2 + pu2);
                    }
                    continue;
                }
                warm = false;
                pu1 = u1;
                pu2 = u2;
                pval = val;
                const double value = P.K0 - val + A.constc;       // the relaxed problem's value at a point of its domain: >= its minimum
                o_w0 = 1.0 - s1 * u1 - s2 * u2;
                o_u1 = u1;
                o_u2 = u2;
                if (!A.full_bound && value <= A.thr) {             // within the threshold already: no bound can prune the child
                    bound = -__builtin_inf();
                    decided = true;
                    break;
                }
                if (!(rmin < __builtin_inf())) {                   // no tumour reads in the prefix: the relaxed problem is the constant
                    bound = value;
                    keep = A.full_bound || !(value - 1e-3 - 1e-12 * fabs(P.K0) > A.thr);
                    decided = true;
                    break;
                }
                if (collinear) {                                    // rank <= 1: the problem lives on one direction; a floor keeps the solve finite
                    const double fl = 1e-9 * (h11 + h22);           // (and is all the conditioning there is: det ~ fl (h11 + h22) by construction)
                    h11 += fl;
                    h22 += fl;
                }
                const double hh = h11 * h22, det = hh - h12 * h12;
                // (a point close to the boundary of the child's domain -- its parent's optimum may be -- has one term that dwarfs the
                // others: an ill-conditioned Hessian there is no reason to give up, the damped step leads away; only a bound is
                // not built on it)
                const bool solid = collinear || det > 1e-9 * hh;
                const double d1 = (h22 * g1 - h12 * g2) / det, d2 = (h11 * g2 - h12 * g1) / det;
                const double lam2 = g1 * d1 + g2 * d2;
                if (!(det > 0.0) || !(lam2 == lam2) || !(fabs(d1) + fabs(d2) < 1e30)) {
                    if (pu1 != 0.0 || pu2 != 0.0) {                 // (once more from the centre of the child's slice)
                        u1 = u2 = pu1 = pu2 = 0.0;
                        pval = -__builtin_inf();
                        continue;
                    }
                    bound = -__builtin_inf();
                    st_open = 1;
                    st_why6 = 1;
                    decided = true;
                    break;
                }
                const double tt = sqrt(fmax(lam2, 0.0) / rmin);
                if (tt < 0.25 && solid) {
                    // min >= value - (lambda^2 / 2)(1 + t + 2 t^2): self-concordance with parameter 2 / sqrt(Rmin), t = lambda / sqrt(Rmin) < 1/2;
                    // 5 % on top like sv_beyond, the floor's share of lambda^2 (collinear rows) and the value's rounding besides
                    const double lb = value - 0.525 * lam2 * (1.0 + tt + 2.0 * tt * tt) * (collinear ? 1.000001 : 1.0) - 1e-3 - 1e-12 * fabs(P.K0);
                    if (!A.full_bound && lb > A.thr) {
                        keep = false;
                        bound = lb;
                        st_pruned = 1;
                        decided = true;
                        break;
                    }
                    if (lam2 < 1e-4) {                              // converged (NLL units): the child's bound stands
                        bound = lb;
                        decided = true;
                        break;
                    }
                }
                if (tt < 0.25 && !solid && lam2 < 1e-4) {           // converged where the Hessian is nearly singular: no bound is built on that
                    bound = -__builtin_inf();
                    st_open = 1;
                    st_why7 = 1;
                    decided = true;
                    break;
                }
                // The step.  The textbook damping 1 / (1 + t) with t = lambda / sqrt(Rmin) never leaves the domain but crawls when one
                // interval's count is small against the others' (t in the hundreds: tens of thousands of steps); the sieve's rule --
                // damp by lambda / sqrt(sum R), take the full step below 0.3 -- gets there in a handful, and a step that does leave the
                // domain is halved back above.  Nothing rests on the iterates: a bound is only ever built where t < 1/4.
                const double tn = sqrt(fmax(lam2, 0.0) / rsum);
                const double step = tt <= 0.25 ? 1.0 : (tn > 0.3 ? 1.0 / (1.0 + tn) : 1.0);
                u1 = __builtin_fma(step, d1, u1);
                u2 = __builtin_fma(step, d2, u2);
            }
            if (!decided) {                                         // (kept: nothing established)
                st_open = 1;
                st_why5 = 1;
#ifdef BNB_DEBUG_PRINT
                if (atomicAdd(&A.stats[7], 1ull) < 4) {
                    printf("NOTCONV d=%d slot=%d (%d,%d) G=%d has=%d u=(%.6g,%.6g) pu=(%.6g,%.6g) pval=%.10g s=(%.6g,%.6g) collinear=%d path:", d, lane, sa, sb, G, (int)has, u1, u2, pu1, pu2, pval, s1, s2, (int)collinear);
                    for (int i = 0; i < d; i++) printf(" %d", (int)pp[i]);
                    printf("\n");
                }
#endif
            }
        }
        const int dc = d + 1;
        emit = keep && (dc >= A.emit_depth || cnt <= (u128)A.emit_max);
    }

    // ---- where the survivors go: one pair of atomics per BLOCK (the memory-side atomic unit serialises same-line atomics)
    const unsigned long long km = ballot64(keep && !emit), em = ballot64(emit);
    if (lane == 0) {
        blk_next[wv] = (unsigned)__builtin_popcountll(km);
        blk_emit[wv] = (unsigned)__builtin_popcountll(em);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned tn = 0, te = 0;
        for (int w = 0; w < BNB_WAVES; w++) {
            tn += blk_next[w];
            te += blk_emit[w];
        }
        blk_base[0] = tn ? atomicAdd(&A.counters[0], (unsigned long long)tn) : 0ull;
        blk_base[1] = te ? atomicAdd(&A.counters[1], (unsigned long long)te) : 0ull;
    }
    __syncthreads();
    if (!live_wave) return;
    unsigned long long nb = blk_base[0], eb = blk_base[1];
    for (int w = 0; w < wv; w++) {
        nb += blk_next[w];
        eb += blk_emit[w];
    }
    if (keep && !emit) {
        const unsigned long long idx = nb + (unsigned)mbcnt(km);
        if (idx < A.out_cap) {
            BnbNode o;
            o.base_lo = (uint64_t)mybase;
            o.base_hi = (uint64_t)(mybase >> 64);
            o.w0 = o_w0;
            o.u1 = o_u1;
            o.u2 = o_u2;
            o.bound = bound;
            o.state = n3_pack(nx);
            o.line = bnb_line_pack(ln);
            A.out[idx] = o;
            unsigned *dst = (unsigned *)(A.out_path + (size_t)idx * A.path_stride);
            const int nw = A.path_stride / 4, wd = d >> 2, sh = (d & 3) * 8;
            for (int i = 0; i < nw; i++) {
                unsigned v = W.path[i];
                if (i == wd) v = (v & ~(0xffu << sh)) | ((unsigned)lane << sh);
                dst[i] = v;
            }
        }
    }
    if (emit) {
        const unsigned long long idx = eb + (unsigned)mbcnt(em);
        if (idx < A.range_cap) {
            BnbRange rg;
            rg.base_lo = (uint64_t)mybase;
            rg.base_hi = (uint64_t)(mybase >> 64);
            rg.count_lo = (uint64_t)cnt;
            rg.count_hi = (uint64_t)(cnt >> 64);
            A.ranges[idx] = rg;
        }
    }
    // statistics: one slot of 64 per wave (cache lines of their own)
    const unsigned long long s0 = bnb_wave_sum_u64(st_solves), s1 = bnb_wave_sum_u64(st_iters), s2 = bnb_wave_sum_u64(st_pruned),
                             s3 = bnb_wave_sum_u64(st_line), s4 = bnb_wave_sum_u64(st_open), s5 = bnb_wave_sum_u64(st_why5),
                             s6 = bnb_wave_sum_u64(st_why6), s7 = bnb_wave_sum_u64(st_why7);
    if (lane == 0) {
        unsigned long long *sl = A.stats + (size_t)(node & (BNB_STAT_SLOTS - 1)) * BNB_STAT_STRIDE;
        atomicAdd(&sl[0], s0);
        atomicAdd(&sl[1], s1);
        if (s2) atomicAdd(&sl[2], s2);
        if (s3) atomicAdd(&sl[3], s3);
        if (s4) atomicAdd(&sl[4], s4);
        if (s5) atomicAdd(&sl[5], s5);
        if (s6) atomicAdd(&sl[6], s6);
        if (s7) atomicAdd(&sl[7], s7);
    }
}

// the bounds of a level, dense (beam search: the host picks the cut)
__global__ void bnb_bounds_kernel(const BnbNode *nodes, unsigned long long n, double *out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double b = nodes[i].bound;
        out[i] = b == b ? b : -__builtin_inf();
    }
}

// keep the nodes whose bound is at most `cut` (beam search)
__global__ void bnb_compact_kernel(const BnbNode *nodes, const unsigned char *paths, unsigned long long n, int path_stride, double cut, BnbNode *out,
                                   unsigned char *out_paths, unsigned long long out_cap, unsigned long long *counter) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const BnbNode nd = nodes[i];
    if (!(nd.bound <= cut) && nd.bound == nd.bound) return;
    const unsigned long long idx = atomicAdd(counter, 1ull);
    if (idx >= out_cap) return;
    out[idx] = nd;
    const unsigned *src = (const unsigned *)(paths + (size_t)i * path_stride);
    unsigned *dst = (unsigned *)(out_paths + (size_t)idx * path_stride);
    for (int k = 0; k < path_stride / 4; k++) dst[k] = src[k];
}

void bnb_launch_expand(const N3Dev &P, const BnbArgs &A, hipStream_t st) {
    if (A.n_in == 0) return;
    hipLaunchKernelGGL(bnb_expand_kernel, dim3((A.n_in + BNB_WAVES - 1) / BNB_WAVES), dim3(64 * BNB_WAVES), 0, st, P, A);
}
void bnb_launch_bounds(const BnbNode *nodes, unsigned long long n, double *out, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(bnb_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, nodes, n, out);
}
void bnb_launch_compact(const BnbNode *nodes, const unsigned char *paths, unsigned long long n, int path_stride, double cut, BnbNode *out,
                        unsigned char *out_paths, unsigned long long out_cap, unsigned long long *counter, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(bnb_compact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, nodes, paths, n, path_stride, cut, out, out_paths,
                       out_cap, counter);
}

// ====================================================================================================================================
// Branch and bound over the MIXTURE space (the arg-min of spaces the row tree above cannot finish: BASELINE configs 3 and 4).
//
// The reference's objective (Optimizer.py:236-244, 273-330) in Poisson form: with lambda_i = rN_i c_i.v, v = s (mu0, mu1, mu2) >= 0,
//      NLL(C, mu) = min over the scale s of  sum_i phi_i(c_i.v)  -  Rtot + Rtot ln Rtot,      phi_i(t) = rN_i t - r_i ln(rN_i t),
// (the multinomial likelihood is the Poisson one at its best total rate) -- SEPARABLE over the intervals once v is fixed.  So
//      min over the matrices C and v in a box B of the objective  >=  sum_i min over the rows c of [a lower bound of phi_i(c.v) on B]
// whatever the row graph allows: a bound for every matrix whose reported mixture lies in B, from m x Q one-dimensional problems.
// Two such bounds, the larger counts: (1) phi_i at the point of [c.lo, c.hi] nearest its minimiser r_i / rN_i (tight for large
// boxes, off by a term LINEAR in the box's size for small ones: every interval picks its own v); (2) the tangents of the convex
// phi_i at the box's centre, summed and minimised over the box's eight corners (a concave function of v: its minimum over a box is
// at a corner) -- one v for all intervals, off by sum r (dt / t)^2 / 2 only.  An octree over v >= 0 keeps the boxes whose bound is
// within the threshold; in the leaves (boxes a few 1e-4 wide) the same per-interval tangent costs bound every single matrix from
// below, and a depth-first walk over the intervals with the budget `threshold` lists the few matrices that fit (mix_list_kernel).
// The host values those with the reference's own procedure (theta_solve_batch) and replays them in enumeration order.
// ====================================================================================================================================
#define MIX_MAX_M 256
#define MIX_MAX_Q 256      // rows of the alphabet here (a search over mixtures needs no 64-bit child masks)
//
// RANK-DEFICIENT MATRICES (round 6).  What the reference reports for a matrix is the objective at the mixture its solver stops at.
// For a matrix of full rank that mixture is >= 0 (its own optimum, the nu = 1/3 fallback) or the value is NaN.  For a matrix whose
// rows (x_i, y_i) lie on ONE LINE of the alphabet's grid hybrj runs on a singular Jacobian and may stop at a nu in [0,1]^3 that does
// not sum to one; M3 (Optimizer.py:318-330) turns it into a mu with a negative entry, never range-checked, and L3 reports a FINITE
// value wherever all products c_i.mu keep one sign -- a value BELOW the matrix's minimum over mu >= 0.  But for rows on the line
// (x0, y0) + t (dx, dy) the product is c.v = alpha + t beta, alpha = tau v0 + x0 v1 + y0 v2, beta = dx v1 + dy v2: whatever the signs
// of v, the value is the same separable objective at SOME (alpha, beta) in R^2 with alpha + t_i beta > 0.  So a quadtree over (alpha, beta)
// per line of the grid, rows restricted to the line, bounds every such outcome: the same kernels, boxes with line != 0, coefficients
// (1, t, 0) in place of (tau, a, b) -- non-negative, so [c.lo, c.hi] still brackets c.v on a box of either sign.
//
// THE LOOP (round 6).  Round 5 walked the octree level by level: a launch, a read-back and a host decision per level, ~60 levels of
// ~0.1 ms for a handful of boxes each, two lists of 2^23 boxes allocated per call, and a give-up once a level held 2^21 boxes.
// Now the boxes live on a STACK in HBM; one iteration = two launches, no host in between: mix_split_kernel takes the top `chunk`
// boxes (depth first: the stack stays a few chunks deep however many boxes the threshold leaves), bounds both halves of each, and
// appends the survivors to a work list; mix_push_kernel copies that list over the boxes just taken and publishes the new top.
// The host enqueues a BATCH of iterations and reads the counters once per batch.
// the bound of a box: max of the two (see above).  One WAVE per box: lane l takes the intervals l, 64 + l, ... (m x rows x 2 logarithms
// are half a millisecond of one thread), the nine partial sums meet by shuffles.
__device__ __forceinline__ double mix_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
// The rows a box ranges over and what multiplies its coordinates: the whole alphabet with (tau, a, b), or the points of its line
// with (1, t, 0).  row(s, a, b, x, y): the s-th row's copy numbers and coefficients; false: not a row of the alphabet.
struct MixRows {
    int n;                       // rows to try
    double c0;                   // coefficient of the first coordinate
    bool ln;
    int x0, y0, dx, dy;
    const uchar2 *rows;          // LDS: slot -> (a, b)
    const unsigned char *slot_of;    // LDS: a | b << 4 -> slot
    __device__ __forceinline__ MixRows(const MixArgs &A, unsigned line, const uchar2 *rows_, const unsigned char *slot_of_) : rows(rows_), slot_of(slot_of_) {
        ln = line != 0;
        if (ln) {
            const MixLine L = A.lines[line - 1];
            n = L.T;
            c0 = 1.0;
            x0 = L.x0; y0 = L.y0; dx = L.dx; dy = L.dy;
        } else {
            n = A.Q;
            c0 = (double)A.tau;
            x0 = y0 = dx = dy = 0;
        }
    }
    __device__ __forceinline__ bool row(int s, int &a, int &b, double &x, double &y) const {
        if (ln) {
            a = x0 + s * dx;
            b = y0 + s * dy;
            x = (double)s;
            y = 0.0;
            return slot_of[a | (b << 4)] != 0xffu;
        }
        const uchar2 rw = rows[s];
        a = rw.x;
        b = rw.y;
        x = (double)a;
        y = (double)b;
        return true;
    }
    __device__ __forceinline__ int slot(int s) const { return ln ? (int)slot_of[(x0 + s * dx) | ((y0 + s * dy) << 4)] : s; }
};
__device__ __forceinline__ void mix_stage_rows(const MixArgs &A, uchar2 *rows, unsigned char *slot_of) {
    for (int s = threadIdx.x; s < A.Q; s += blockDim.x) rows[s] = make_uchar2((unsigned char)(A.rowtab[s] & 15u), (unsigned char)(A.rowtab[s] >> 4));
    for (int s = threadIdx.x; s < 256; s += blockDim.x) slot_of[s] = A.slot_of[s];
}

// What a box needs of a ROW, whatever the interval (round 6): c.v at the centre and at the two extreme corners, their LOGARITHMS --
// ln(rN_i c.v) = ln rN_i + ln c.v: the logarithm of round 5's kernel, one per (interval, row, box), is one per (row, box) plus a
// constant of the interval --, the reciprocal at the centre and the eight corner offsets of the tangent.  Lane s of the wave fills the
// entry of row s0 + s; then every lane walks the table for its intervals (all lanes read the same entry: LDS broadcasts).
#define MIX_CHUNK 64
struct MixRowTab {
    double tc[MIX_CHUNK], ltc[MIX_CHUNK], itc[MIX_CHUNK];
    double tlo[MIX_CHUNK], ltlo[MIX_CHUNK], thi[MIX_CHUNK], lthi[MIX_CHUNK];
    double E[8][MIX_CHUNK];                 // sum_j (+-) c_j h_j, the corner's offset from the centre in units of c.v
    unsigned char a[MIX_CHUNK], b[MIX_CHUNK], ok[MIX_CHUNK];
};

__device__ double mix_cell_bound(const MixArgs &A, const MixCell &c, const uchar2 *rows, const unsigned char *slot_of, const double2 *ltab, int lane, double &centre,
                                 MixRowTab &T) {
    const MixRows R(A, c.line, rows, slot_of);
    double vc[3], h[3];
    for (int j = 0; j < 3; j++) {
        vc[j] = 0.5 * (c.lo[j] + c.hi[j]);
        h[j] = 0.5 * (c.hi[j] - c.lo[j]);
    }
    // per interval of this lane (at most MIX_IVL: m <= 256), across the chunks of rows
    constexpr int MIX_IVL = (MIX_MAX_M + WAVE - 1) / WAVE;
    double lb1 = 0.0, cs = 0.0, lbv[8];
    for (int k = 0; k < 8; k++) lbv[k] = 0.0;
    // (a box of m <= 64 intervals: one interval per lane, its state in registers across the chunks; more: the chunks are walked per
    // interval -- the table of a chunk is rebuilt for each group of 64 intervals only when the alphabet has more than 64 rows)
    for (int g = 0; g < MIX_IVL; g++) {
        const int i = lane + WAVE * g;
        if (g >= (A.m + WAVE - 1) / WAVE) break;            // (wave-uniform)
        const bool mine = i < A.m;
        const MixIv iv = mine ? A.iv[i] : MixIv{1.0, 0.0, 0.0, 0.0, 0.0};
        const int l = mine ? A.lb[i] : 0, u = mine ? A.ub[i] : 0;
        bool any = false, holds = false;
        double tbelow = -__builtin_inf(), lbelow = 0.0, tabove = __builtin_inf(), labove = 0.0, bestv[8], cbest = __builtin_inf();
        for (int k = 0; k < 8; k++) bestv[k] = __builtin_inf();
        for (int s0 = 0; s0 < R.n; s0 += MIX_CHUNK) {
            if (g == 0 || R.n > MIX_CHUNK) {
                // ---- the table of rows s0 .. s0 + 63
                wave_lds_sync();
                const int s = s0 + lane;
                int a = 0, b = 0;
                double x = 0.0, y = 0.0;
                const bool ok = s < R.n && R.row(s, a, b, x, y) && (A.tau - a) * (A.tau - b) >= 0;
                const double tcs = R.c0 * vc[0] + x * vc[1] + y * vc[2];
                const double tls = R.c0 * c.lo[0] + x * c.lo[1] + y * c.lo[2], ths = R.c0 * c.hi[0] + x * c.hi[1] + y * c.hi[2];
                T.ok[lane] = ok ? 1 : 0;
                T.a[lane] = (unsigned char)a;
                T.b[lane] = (unsigned char)b;
                T.tc[lane] = tcs;
                T.ltc[lane] = tcs > 0.0 ? smx_log(tcs, ltab) : 0.0;
                T.itc[lane] = tcs > 0.0 ? 1.0 / tcs : 0.0;
                T.tlo[lane] = tls;
                T.ltlo[lane] = tls > 0.0 ? smx_log(tls, ltab) : 0.0;
                T.thi[lane] = ths;
                T.lthi[lane] = ths > 0.0 ? smx_log(ths, ltab) : 0.0;
                const double e0 = R.c0 * h[0], e1 = x * h[1], e2 = y * h[2];
#pragma unroll
                for (int k = 0; k < 8; k++) T.E[k][lane] = ((k & 1) ? e0 : -e0) + ((k & 2) ? e1 : -e1) + ((k & 4) ? e2 : -e2);
                wave_lds_sync();
            }
            const int ns = R.n - s0 < MIX_CHUNK ? R.n - s0 : MIX_CHUNK;
            for (int s = 0; s < ns; s++) {
                if (!T.ok[s]) continue;                                  // (wave-uniform)
                const int a = T.a[s], b = T.b[s];
                if (a < l || a > u || b < l || b > u) continue;
                any = true;
                // (1) the clamp bound: phi_i is convex with its minimum at ts -- over the rows the smallest clamped value is phi(ts) if
                // some row's interval [tlo, thi] holds ts, else the better of phi(largest thi below ts) and phi(smallest tlo above it).
                // (The coefficients are >= 0: c.lo <= c.v <= c.hi on the box whatever the signs of its corners.)
                const double tlo = T.tlo[s], thi = T.thi[s];
                if (thi < iv.ts) {
                    if (thi > tbelow) {
                        tbelow = thi;
                        lbelow = T.lthi[s];
                    }
                } else if (tlo > iv.ts) {
                    if (tlo < tabove) {
                        tabove = tlo;
                        labove = T.ltlo[s];
                    }
                } else {
                    holds = true;
                }
                // (2) the tangent at the centre, at the eight corners
                const double tc = T.tc[s];
                if (tc > 0.0) {
                    const double f0 = __builtin_fma(iv.N, tc, -iv.r * (iv.lnN + T.ltc[s])), f1 = __builtin_fma(-iv.r, T.itc[s], iv.N);
                    cbest = fmin(cbest, f0);
#pragma unroll
                    for (int k = 0; k < 8; k++) bestv[k] = fmin(bestv[k], __builtin_fma(f1, T.E[k][s], f0));
                } else {
                    // not positive at the centre (a box of either sign astride the row's zero line): the CONSTANT bound phi_i at the
                    // point of (0, thi] nearest its minimiser -- finite, where -inf would let every matrix through -- or, never
                    // positive on the box, no row of a matrix with a value here
                    double val = __builtin_inf();
                    if (thi > 0.0) val = thi < iv.ts ? __builtin_fma(iv.N, thi, -iv.r * (iv.lnN + T.lthi[s])) : iv.phimin;
                    for (int k = 0; k < 8; k++) bestv[k] = fmin(bestv[k], val);
                }
            }
        }
        if (mine) {
            double best1 = __builtin_inf();
            if (any) {
                if (holds) {
                    best1 = iv.phimin;
                } else {
                    if (tbelow > 0.0) best1 = __builtin_fma(iv.N, tbelow, -iv.r * (iv.lnN + lbelow));
                    if (tabove < __builtin_inf()) best1 = fmin(best1, __builtin_fma(iv.N, tabove, -iv.r * (iv.lnN + labove)));
                }
            }
            lb1 += best1;
            cs += cbest;
            for (int k = 0; k < 8; k++) lbv[k] += bestv[k];
        }
    }
    centre = mix_wave_sum(cs) + A.cst;
    lb1 = mix_wave_sum(lb1);
    double lb2 = __builtin_inf();
    for (int k = 0; k < 8; k++) lb2 = fmin(lb2, mix_wave_sum(lbv[k]));
    // (ln(rN c.v) is taken as ln rN + ln c.v: a few 1e-16 of r_i ln per term -- 1e-9 of the bound at most; the bound is a LOWER bound)
    return fmax(lb1, lb2) + A.cst - 1e-7;
}

// One wave per CHILD of a box off the stack: the parent is cut in two along its widest side (widths weighted by the leaf size of
// the side), the child's bound decides whether it goes on -- to the work list, or, small enough, to the leaves.  A persistent
// grid: the number of boxes is read from the device counters (ctr[MIX_TOP + parity], written by the previous iteration's push).
#define MIX_WAVES 4
__global__ __launch_bounds__(64 * MIX_WAVES) void mix_split_kernel(MixArgs A, const MixCell *stack, MixCell *work, MixCell *leaves, unsigned long long leaf_cap,
                                                                    unsigned long long *ctr, int par, int drop_leaves) {
    __shared__ uchar2 rows[MIX_MAX_Q];
    __shared__ unsigned char slot_of[256];
    __shared__ double2 ltab[128];                    // smx_log's table (the scorers' logarithm: 15 vector instructions, within an ulp)
    __shared__ MixRowTab rtab[MIX_WAVES];            // per wave: the rows of the box it bounds (mix_cell_bound)
    const unsigned long long top = ctr[MIX_TOP0 + par];
    const unsigned long long n = top < (unsigned long long)A.chunk ? top : (unsigned long long)A.chunk;
    const unsigned long long first = (unsigned long long)blockIdx.x * MIX_WAVES;
    if (first >= 2 * n) return;                       // (whole blocks leave: the grid is sized for a full chunk)
    mix_stage_rows(A, rows, slot_of);
    smx_log_stage(ltab);
    __syncthreads();
    const MixCell *in = stack + (top - n);
    const int lane = threadIdx.x & 63;
    const unsigned long long stride = (unsigned long long)gridDim.x * MIX_WAVES;
    for (unsigned long long k = first + (threadIdx.x >> 6); k < 2 * n; k += stride) {
        MixCell c = in[k >> 1];
        const double *lf = c.line ? A.leaf_line : A.leaf;
        int ax = 0;
        double wbest = -1.0;
        for (int j = 0; j < 3; j++) {
            const double w = (c.hi[j] - c.lo[j]) / lf[j];
            if (w > wbest) {
                wbest = w;
                ax = j;
            }
        }
        const double mid = 0.5 * (c.lo[ax] + c.hi[ax]);
        // a sharded search: the boxes that have just become as small as `shard_mult` leaves a side are dealt out -- each belongs to ONE rank,
        // by its position --; above that size all ranks walk alike (a handful of boxes per level), below it a rank walks its own.  (A
        // fixed number of cuts below the root would not do: the root is hundreds of times the region the data allows, and the first
        // thirty levels hold one or two boxes each.)
        bool was_small = true;
        if (A.shard_G > 1)
            for (int j = 0; j < 3; j++) was_small = was_small && (c.hi[j] - c.lo[j]) <= A.shard_mult * lf[j];
        if (k & 1) c.lo[ax] = mid; else c.hi[ax] = mid;
        c.key = (c.key << 1) | (unsigned)(k & 1);
        c.depth++;
        if (A.shard_G > 1 && !was_small) {
            bool now_small = true;
            for (int j = 0; j < 3; j++) now_small = now_small && (c.hi[j] - c.lo[j]) <= A.shard_mult * lf[j];
            if (now_small) {
                const unsigned long long hsh = mix_ord(c.lo[0]) * 0x9E3779B97F4A7C15ull + mix_ord(c.lo[1]) * 0xC2B2AE3D27D4EB4Full +
                                               mix_ord(c.lo[2]) * 0x165667B19E3779F9ull + (unsigned long long)c.line * 0x27D4EB2F165667C5ull;
                if ((int)((hsh >> 17) % (unsigned long long)A.shard_G) != A.shard_g) continue;
            }
        }
        double centre;
        const double lb = mix_cell_bound(A, c, rows, slot_of, ltab, lane, centre, rtab[threadIdx.x >> 6]);
        if (!(lb <= A.thr) || lb == __builtin_inf() || lane != 0) continue;      // (+inf: no row of some interval has a value in the box)
        // a DIVE ranks by what the best assignment of rows comes to AT the box's centre -- an attainable value of the relaxed problem,
        // where the bounds of large boxes are all the saturated model's and tell nothing apart
        c.lb = A.dive ? lb + A.dive_blend * (centre - lb) : lb;
        bool leaf = true;
        for (int j = 0; j < 3; j++) leaf = leaf && (c.hi[j] - c.lo[j]) <= lf[j];
        if (leaf) {
            atomicAdd(&ctr[MIX_LEAVES_ALL], 1ull);
            if (c.line) atomicAdd(&ctr[MIX_LEAVES_LINE], 1ull);
            atomicMin(&ctr[c.line ? MIX_MINB_LINE : MIX_MINB], mix_ord(lb));
            if (drop_leaves && ctr[MIX_LEAVES] >= leaf_cap) continue;       // (a proposal pass keeps the first leaf_cap leaves)
            const unsigned long long idx = atomicAdd(&ctr[MIX_LEAVES], 1ull);
            if (idx < leaf_cap) leaves[idx] = c;
            else atomicOr(&ctr[MIX_OVERFLOW], 2ull);
        } else {
            const unsigned long long idx = atomicAdd(&ctr[MIX_WK0 + par], 1ull);
            work[idx] = c;                                                     // (2 x chunk entries: cannot overflow)
        }
    }
}

// ... and the survivors go back on the stack, over the boxes the iteration took; the new top is published for the next iteration.
__global__ __launch_bounds__(256) void mix_push_kernel(MixCell *stack, unsigned long long stack_cap, const MixCell *work, unsigned long long *ctr, unsigned chunk,
                                                       int par) {
    const unsigned long long top = ctr[MIX_TOP0 + par];
    const unsigned long long n = top < (unsigned long long)chunk ? top : (unsigned long long)chunk;
    const unsigned long long base = top - n;
    unsigned long long c = ctr[MIX_WK0 + par];
    const bool over = base + c > stack_cap;
    if (over) c = stack_cap - base;
    const uint4 *src = (const uint4 *)work;
    uint4 *dst = (uint4 *)(stack + base);
    const unsigned long long words = c * (sizeof(MixCell) / sizeof(uint4));
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * blockDim.x) dst[i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ctr[MIX_TOP0 + 1 - par] = base + c;
        ctr[MIX_WK0 + 1 - par] = 0ull;
        if (n) {
            ctr[MIX_TESTED] += 2 * n;
            ctr[MIX_ITERS] += 1ull;
            if (base + c > ctr[MIX_MAXTOP]) ctr[MIX_MAXTOP] = base + c;
            if (over) ctr[MIX_OVERFLOW] |= 1ull;
        }
    }
}

// The matrices of a leaf box: for corner k of the box, cost_i(c) = the tangent of phi_i at the centre, evaluated at the corner, bounds
// phi_i(c.v) from below on the whole box for the v that minimises the matrix's (linear) tangent sum -- which is a corner.  So every
// matrix whose objective is within `thr` somewhere in the box has sum_i cost_i(c_i) <= thr for at least one corner: a depth-first
// walk over the intervals (rows in slot order, valid and within bounds, the reference's edge rule between consecutive rows) with
// that budget and the suffix minima as look-ahead lists them.  A record is m slot bytes.  (A leaf of a line walks the rows of its
// line only: every matrix it lists is rank deficient.)
//
// One WAVE per (leaf, corner) (round 6).  Round 5's walk -- one THREAD each -- evaluated cost_i(c) with its logarithm at every step
// of the walk, for all rows of the alphabet at every depth: 25 ms for the 934 leaves of config 5's shape (a hundred waves on the
// chip, each a serial chain of library logarithms), and the whole of a flat likelihood's minutes.  Now the lanes take the
// intervals: every cost once (phase 1: the per-interval minima, their suffix sums), then per interval the VIABLE rows -- those
// whose cost exceeds the interval's minimum by no more than the slack thr - (sum of the minima): no other row can be part of a
// matrix within the budget -- are kept in LDS with their costs (phase 2: one to three rows per interval on the bench's data), and
// lane 0 walks those lists (phase 3: table look-ups, no arithmetic beyond the running sum).  An interval with more than MIX_VIA
// viable rows (a likelihood that flat) is walked over all its rows with the cost taken on the fly, as before.
#define MIX_VIA 8
struct MixWalk {
    double suf[MIX_MAX_M + 1];
    double cost[MIX_MAX_M][MIX_VIA];
    unsigned char srow[MIX_MAX_M][MIX_VIA];   // the viable rows (row index: slot, or t on a line), ascending
    unsigned char nvia[MIX_MAX_M];            // how many (0xff: more than MIX_VIA -- all rows, costs on the fly)
    unsigned char cur[MIX_MAX_M], pa[MIX_MAX_M], pb[MIX_MAX_M], rec[MIX_MAX_M];
    double part[MIX_MAX_M + 1];
};
__global__ __launch_bounds__(64) void mix_list_kernel(MixArgs A, const MixCell *leaves, unsigned long long n_leaves, unsigned char *out, unsigned long long out_cap,
                                                      unsigned long long per_thread_cap, unsigned long long max_steps, unsigned long long *ctr, unsigned *seen_tab,
                                                      unsigned long long seen_mask) {
    __shared__ uchar2 rows[MIX_MAX_Q];
    __shared__ unsigned char slot_of[256];
    __shared__ MixWalk W;
    mix_stage_rows(A, rows, slot_of);
    __syncthreads();
    const unsigned long long k = blockIdx.x;
    if (k >= 8 * n_leaves) return;
    const MixCell c = leaves[k >> 3];
    const int corner = (int)(k & 7);
    if (c.line && (corner & 4)) return;               // (a line's box has two sides: four corners)
    if (!(c.lb <= A.thr)) return;                     // (a leaf kept under an earlier, looser threshold)
    const int lane = threadIdx.x;
    const MixRows R(A, c.line, rows, slot_of);
    double vc[3], dv[3];
    for (int j = 0; j < 3; j++) {
        vc[j] = 0.5 * (c.lo[j] + c.hi[j]);
        const double h = 0.5 * (c.hi[j] - c.lo[j]);
        dv[j] = ((corner >> j) & 1) ? h : -h;
    }
    auto cost = [&](int i, int s, bool &ok, int &a, int &b) -> double {
        double x, y;
        const int l = A.lb[i], u = A.ub[i];
        ok = R.row(s, a, b, x, y) && !(a < l || a > u || b < l || b > u || (A.tau - a) * (A.tau - b) < 0);
        if (!ok) return __builtin_inf();
        const double r = A.r[i], N = A.rN[i];
        const double tc = R.c0 * vc[0] + x * vc[1] + y * vc[2];
        if (!(tc > 0.0)) {
            // (see mix_cell_bound: the constant bound on (0, thi], or not a row of any matrix with a value in this box)
            const double thi = R.c0 * c.hi[0] + x * c.hi[1] + y * c.hi[2];
            if (!(thi > 0.0)) {
                ok = false;
                return __builtin_inf();
            }
            const double t = fmin(thi, r / N);
            return r > 0.0 ? N * t - r * log(N * t) : 0.0;
        }
        const double f0 = r > 0.0 ? N * tc - r * log(N * tc) : N * tc, f1 = N - r / tc;
        return f0 + f1 * (R.c0 * dv[0] + x * dv[1] + y * dv[2]);
    };
    // ---- phase 1: the cheapest row of every interval, the suffix sums of those minima
    for (int i = lane; i < A.m; i += WAVE) {
        double best = __builtin_inf();
        for (int s = 0; s < R.n; s++) {
            bool ok;
            int a, b;
            const double v = cost(i, s, ok, a, b);
            if (ok) best = fmin(best, v);
        }
        W.part[i] = best;                              // (the walk's own array: free until phase 3)
    }
    wave_lds_sync();
    if (lane == 0) {
        W.suf[A.m] = A.cst;
        for (int i = A.m - 1; i >= 0; i--) W.suf[i] = W.suf[i + 1] + W.part[i];
    }
    wave_lds_sync();
    const double total0 = W.suf[0];
    if (!(total0 <= A.thr)) return;
    if (__hip_atomic_load(&ctr[MIX_CUT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;
    // ---- phase 2: the viable rows of every interval.  A matrix within the budget pays at least the minimum in every OTHER interval,
    // so its row in interval i costs at most that interval's minimum + the slack (less a rounding allowance on the sums).
    const double slack = (A.thr - total0) + 1e-9 * (fabs(total0) + 1.0);
    for (int i = lane; i < A.m; i += WAVE) {
        const double lim = (W.suf[i] - W.suf[i + 1]) + slack;
        int n = 0;
        for (int s = 0; s < R.n; s++) {
            bool ok;
            int a, b;
            const double v = cost(i, s, ok, a, b);
            if (ok && v <= lim) {
                if (n < MIX_VIA) {
                    W.cost[i][n] = v;
                    W.srow[i][n] = (unsigned char)s;
                }
                n++;
            }
        }
        W.nvia[i] = n > MIX_VIA ? 0xffu : (unsigned char)n;
    }
    wave_lds_sync();
    if (lane != 0) return;
    // ---- phase 3: the walk (lane 0).  cur[i]: the next entry of interval i's list to try (or, list overflowed, the next row).
    int i = 0;
    unsigned long long found = 0, steps = 0;
    W.cur[0] = 0;
    W.part[0] = 0.0;
    while (i >= 0) {
        // (a budget on the walk itself: no likelihood, however flat, may hang the device -- and once one walk has run out, the
        // list is void: the others stop at their next look)
        if (++steps > max_steps) {
            atomicAdd(&ctr[MIX_CUT], 1ull);
            return;
        }
        if ((steps & 4095ull) == 0 && __hip_atomic_load(&ctr[MIX_CUT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull) return;
        const bool full = W.nvia[i] == 0xffu;
        const int lim_n = full ? R.n : (int)W.nvia[i];
        if ((int)W.cur[i] >= lim_n) {       // this depth is exhausted
            i--;
            if (i >= 0) W.cur[i]++;
            continue;
        }
        const int e = W.cur[i];
        int s, a, b;
        double v;
        bool ok = true;
        if (full) {
            s = e;
            v = cost(i, s, ok, a, b);
        } else {
            s = W.srow[i][e];
            v = W.cost[i][e];
            double x, y;
            (void)R.row(s, a, b, x, y);
        }
        bool go = ok && W.part[i] + v + W.suf[i + 1] <= A.thr;
        if (go && i > 0)                 // Enumerator._is_valid_edge (Enumerator.py:258-260): the same row, or some component larger
            go = (a == (int)W.pa[i - 1] && b == (int)W.pb[i - 1]) || a > (int)W.pa[i - 1] || b > (int)W.pb[i - 1];
        if (!go) {
            W.cur[i]++;
            continue;
        }
        W.pa[i] = (unsigned char)a;
        W.pb[i] = (unsigned char)b;
        if (i == A.m - 1) {
            if (found < per_thread_cap) {
                // the matrix as slot bytes (in LDS: the walk's own row-index array is needed on), and its hash
                unsigned long long hsh = 0x9E3779B97F4A7C15ull;
                for (int d = 0; d < A.m; d++) {
                    const int ed = W.cur[d];
                    const unsigned char sl = (unsigned char)R.slot(W.nvia[d] == 0xffu ? ed : (int)W.srow[d][ed]);
                    W.rec[d] = sl;
                    hsh = (hsh ^ sl) * 0x100000001B3ull;
                    hsh ^= hsh >> 29;
                }
                // SEEN BEFORE?  Every (leaf, corner) whose budget a matrix meets lists it -- hundreds of times in a wide region, and
                // a flat likelihood's ten million raw records were four seconds of sorting on the host.  A table of record indices,
                // open addressing: an entry is published AFTER its record is written, a hit compares the BYTES (exact: a matrix is
                // never dropped for another's hash), a lost race leaves a duplicate for the host's sort, which stays.
                bool seen = false;
                unsigned long long pos = (hsh * 0xD6E8FEB86659FD93ull) >> 32;
                unsigned long long my = ~0ull;
                if (seen_tab) {
                    for (unsigned probe = 0; probe < 64u; probe++, pos++) {
                        unsigned *slot = seen_tab + (pos & seen_mask);
                        unsigned e = __hip_atomic_load(slot, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        if (e == 0u) {
                            if (my == ~0ull) {                      // write the record first, then publish it
                                my = atomicAdd(&ctr[MIX_LISTED], 1ull);
                                if (my >= out_cap || my >= 0xfffffffeull) break;
                                unsigned char *dst = out + my * (size_t)A.m;
                                for (int d = 0; d < A.m; d++) dst[d] = W.rec[d];
                                __threadfence();
                            }
                            const unsigned prev = atomicCAS(slot, 0u, (unsigned)my + 1u);
                            if (prev == 0u) break;                   // published
                            e = prev;                                // somebody else took the slot meanwhile: is it the same matrix?
                        }
                        if (my == ~0ull) {
                            const unsigned char *other = out + (size_t)(e - 1u) * A.m;
                            bool same = true;
                            for (int d = 0; d < A.m && same; d++) same = other[d] == W.rec[d];
                            if (same) {
                                seen = true;
                                break;
                            }
                        }
                    }
                }
                if (!seen && my == ~0ull) {                          // (no table, or 64 probes without an answer: listed as it is)
                    const unsigned long long idx = atomicAdd(&ctr[MIX_LISTED], 1ull);
                    if (idx < out_cap) {
                        unsigned char *dst = out + idx * (size_t)A.m;
                        for (int d = 0; d < A.m; d++) dst[d] = W.rec[d];
                    }
                }
            } else {
                atomicAdd(&ctr[MIX_CUT], 1ull);      // (this walk found more than its share: the host must not trust the list)
            }
            found++;
            W.cur[i]++;
            continue;
        }
        W.part[i + 1] = W.part[i] + v;
        i++;
        W.cur[i] = 0;
    }
}

// The best `beam` of the iteration's children (by their score) go back on the stack -- a DIVE: no threshold, a fixed number of boxes
// per level of the tree, all the way down to the leaf size; its leaves propose the matrices whose value starts the real search.
// c <= 2 x beam <= 2048 children ranked by counting.
#define MIX_BEAM_MAX 1024
#define MIX_SEL_ITEMS 64           // children ranked per block of 256 threads: four lanes share one child's comparisons
// (ties -- the mirror image of a box under the exchange of the tumour columns scores alike -- are broken by the box's position, not by
// where the atomics happened to put it in the list: the same input gives the same dive)
// NICHES: the boxes are ranked within groups of like mixtures first -- 8 bands of the normal fraction mu0 x 4 of the split mu1 : mu2
// at the box's centre -- and the beam takes the best of every group before the second best of any: on BASELINE config 4 the plain
// beam settled in a basin at mu0 = 0.44, 58 units above the minimum at mu0 = 0.61, whatever its width and ranking.
// Two launches (a rank needs every child's group rank): mix_rank_kernel<false> writes the rank within the group, <true> the final
// position and the boxes; c <= 2048 children, four lanes per child, 64 children per block.
struct MixSelLds {
    unsigned long long key[2 * MIX_BEAM_MAX], key2[2 * MIX_BEAM_MAX];
    unsigned short grp[2 * MIX_BEAM_MAX];
};
template <bool FINAL>
__global__ __launch_bounds__(256) void mix_rank_kernel(MixCell *stack, const MixCell *work, unsigned long long *ctr, unsigned short *grank, unsigned beam, int par,
                                                       int niches) {
    __shared__ MixSelLds L;
    const unsigned long long top = ctr[MIX_TOP0 + par];
    unsigned long long c = ctr[MIX_WK0 + par];
    if (c > 2ull * MIX_BEAM_MAX) c = 2ull * MIX_BEAM_MAX;
    if ((unsigned long long)blockIdx.x * MIX_SEL_ITEMS < c) {
        for (unsigned i = threadIdx.x; i < (unsigned)c; i += blockDim.x) {
            const MixCell w = work[i];
            L.key[i] = mix_ord(w.lb);
            L.key2[i] = mix_ord(w.lo[0]) * 0x9E3779B97F4A7C15ull + mix_ord(w.lo[1]) * 0xC2B2AE3D27D4EB4Full + mix_ord(w.lo[2]) * 0x165667B19E3779F9ull + mix_ord(w.hi[0]);
            if (FINAL) {
                L.grp[i] = grank[i];                     // (the group rank takes the group's place in the order)
            } else {
                const double v0 = w.lo[0] + w.hi[0], v1 = w.lo[1] + w.hi[1], v2 = w.lo[2] + w.hi[2], sum = v0 + v1 + v2, t12 = v1 + v2;
                int g0 = sum > 0.0 ? (int)(8.0 * v0 / sum) : 0, g1 = t12 > 0.0 ? (int)(4.0 * v1 / t12) : 0;
                g0 = g0 < 0 ? 0 : (g0 > 7 ? 7 : g0);
                g1 = g1 < 0 ? 0 : (g1 > 3 ? 3 : g1);
                L.grp[i] = (unsigned short)(niches ? g0 * 4 + g1 : 0);
            }
        }
        __syncthreads();
        const unsigned i = blockIdx.x * MIX_SEL_ITEMS + (threadIdx.x >> 2), sub = threadIdx.x & 3u;
        unsigned rank = 0;
        if (i < (unsigned)c) {
            const unsigned long long k = L.key[i], k2 = L.key2[i];
            const unsigned short g = L.grp[i];
            for (unsigned j = sub; j < (unsigned)c; j += 4) {
                const bool better = L.key[j] < k || (L.key[j] == k && (L.key2[j] < k2 || (L.key2[j] == k2 && j < i)));
                // <false>: among the children of the same group; <true>: a smaller group rank first, then the better child
                const bool before = FINAL ? (L.grp[j] < g || (L.grp[j] == g && better)) : (L.grp[j] == g && better);
                rank += before ? 1u : 0u;
            }
        }
        rank += __shfl_xor(rank, 1, WAVE);
        rank += __shfl_xor(rank, 2, WAVE);
        if (i < (unsigned)c && sub == 0) {
            if (FINAL) {
                if (rank < beam) stack[rank] = work[i];
            } else {
                grank[i] = (unsigned short)rank;
            }
        }
    }
    if (FINAL && blockIdx.x == 0 && threadIdx.x == 0) {
        ctr[MIX_TOP0 + 1 - par] = c < beam ? c : beam;
        ctr[MIX_WK0 + 1 - par] = 0ull;
        if (top) {
            ctr[MIX_TESTED] += 2 * top;
            ctr[MIX_ITERS] += 1ull;
            if (c > ctr[MIX_MAXTOP]) ctr[MIX_MAXTOP] = c;
        }
    }
}

// n_max: an upper bound of the boxes this iteration can find on the stack (the host knows the top of the last batch; an iteration at
// most doubles it) -- the grids are sized for that, not for a full chunk: a tree's first and last levels hold a handful of boxes,
// and 4096 blocks that look at a counter and leave were most of a small search's time.
void mix_launch_iteration(const MixArgs &A, MixCell *stack, unsigned long long stack_cap, MixCell *work, MixCell *leaves, unsigned long long leaf_cap,
                          unsigned long long *ctr, int parity, int drop_leaves, unsigned long long n_max, unsigned beam, hipStream_t st) {
    n_max = std::max<unsigned long long>(1, std::min<unsigned long long>(n_max, A.chunk));
    // (a wave bounds up to ~4 boxes of a full chunk)
    const unsigned blocks = (unsigned)std::min<unsigned long long>(4096ull, (2ull * n_max + MIX_WAVES - 1) / MIX_WAVES);
    hipLaunchKernelGGL(mix_split_kernel, dim3(blocks), dim3(64 * MIX_WAVES), 0, st, A, stack, work, leaves, leaf_cap, ctr, parity, drop_leaves);
    if (beam) {
        // (the group ranks live behind the work list's 2 x chunk boxes: 2 x MIX_BEAM_MAX shorts)
        unsigned short *grank = (unsigned short *)(work + 2ull * A.chunk);
        const unsigned sblocks = (2 * beam + MIX_SEL_ITEMS - 1) / MIX_SEL_ITEMS;
        hipLaunchKernelGGL(mix_rank_kernel<false>, dim3(sblocks), dim3(256), 0, st, stack, work, ctr, grank, beam, parity, A.dive_niches);
        hipLaunchKernelGGL(mix_rank_kernel<true>, dim3(sblocks), dim3(256), 0, st, stack, work, ctr, grank, beam, parity, A.dive_niches);
    } else {
        const unsigned pblocks = (unsigned)std::min<unsigned long long>(256ull, (2ull * n_max * (sizeof(MixCell) / 16) + 255) / 256);
        hipLaunchKernelGGL(mix_push_kernel, dim3(pblocks), dim3(256), 0, st, stack, stack_cap, work, ctr, A.chunk, parity);
    }
}
void mix_launch_list(const MixArgs &A, const MixCell *leaves, unsigned long long n_leaves, unsigned char *out, unsigned long long out_cap,
                     unsigned long long per_thread_cap, unsigned long long max_steps, unsigned long long *ctr, unsigned *seen_tab, unsigned long long seen_mask,
                     hipStream_t st) {
    if (!n_leaves) return;
    hipLaunchKernelGGL(mix_list_kernel, dim3((unsigned)(8 * n_leaves)), dim3(64), 0, st, A, leaves, n_leaves, out, out_cap, per_thread_cap, max_steps, ctr, seen_tab,
                       seen_mask);
}
