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
#define MIX_MAX_Q 256      // rows of the alphabet here (a search over mixtures needs no 64-bit child masks)
// the bound of a box: max of the two (see above).  One WAVE per box: lane l takes the intervals l, 64 + l, ... (m x rows x 2 logarithms
// are half a millisecond of one thread -- a level of the octree holds a handful of boxes as often as a million, and its 60 levels
// are walked one launch after the other), the nine partial sums meet by shuffles.
__device__ __forceinline__ double mix_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ double mix_cell_bound(const MixArgs &A, const MixCell &c, const float2 *rows, const double2 *ltab, int lane) {
    const double tau = (double)A.tau;
    double vc[3], h[3];
    for (int j = 0; j < 3; j++) {
        vc[j] = 0.5 * (c.lo[j] + c.hi[j]);
        h[j] = 0.5 * (c.hi[j] - c.lo[j]);
    }
    double lb1 = 0.0, lbv[8];
    for (int k = 0; k < 8; k++) lbv[k] = 0.0;
    for (int i = lane; i < A.m; i += WAVE) {
        const double r = A.r[i], N = A.rN[i], ts = r / N;
        const int l = A.lb[i], u = A.ub[i];
        // (1): phi_i is convex with its minimum at ts, so over the rows the smallest clamped value is phi(ts) if some row's interval
        // [tlo, thi] holds ts, else the better of phi(largest thi below ts) and phi(smallest tlo above it): two logarithms per
        // interval, not one per row (round 5's first version: m x rows library logarithms were two thirds of the kernel)
        bool any = false, holds = false;
        double tbelow = -__builtin_inf(), tabove = __builtin_inf(), bestv[8];
        for (int k = 0; k < 8; k++) bestv[k] = __builtin_inf();
        for (int s = 0; s < A.Q; s++) {
            const float2 rw = rows[s];
            const int a = (int)rw.x, b = (int)rw.y;
            if (a < l || a > u || b < l || b > u || (A.tau - a) * (A.tau - b) < 0) continue;
            any = true;
            const double x = (double)a, y = (double)b;
            const double tlo = tau * c.lo[0] + x * c.lo[1] + y * c.lo[2], thi = tau * c.hi[0] + x * c.hi[1] + y * c.hi[2];
            if (thi < ts) tbelow = fmax(tbelow, thi);
            else if (tlo > ts) tabove = fmin(tabove, tlo);
            else holds = true;
            // (2) tangent at the centre, at the eight corners
            const double tc = tau * vc[0] + x * vc[1] + y * vc[2];
            if (tc > 0.0) {
                const double f0 = r > 0.0 ? N * tc - r * smx_log(N * tc, ltab) : N * tc, f1 = N - r / tc;
                const double d0 = f1 * tau * h[0], d1 = f1 * x * h[1], d2 = f1 * y * h[2];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const double val = f0 + ((k & 1) ? d0 : -d0) + ((k & 2) ? d1 : -d1) + ((k & 4) ? d2 : -d2);
                    bestv[k] = fmin(bestv[k], val);
                }
            } else {
                for (int k = 0; k < 8; k++) bestv[k] = -__builtin_inf();
            }
        }
        auto phi = [&](double t) { return r > 0.0 ? (t > 0.0 ? N * t - r * smx_log(N * t, ltab) : __builtin_inf()) : N * t; };
        double best1 = __builtin_inf();
        if (any) {
            if (holds) {
                best1 = phi(ts);
            } else {
                if (tbelow > -__builtin_inf()) best1 = phi(tbelow);
                if (tabove < __builtin_inf()) best1 = fmin(best1, phi(tabove));
            }
        }
        lb1 += best1;
        for (int k = 0; k < 8; k++) lbv[k] += bestv[k];
    }
    lb1 = mix_wave_sum(lb1);
    double lb2 = __builtin_inf();
    for (int k = 0; k < 8; k++) lb2 = fmin(lb2, mix_wave_sum(lbv[k]));
    return fmax(lb1, lb2) + A.cst;
}

// One wave per CHILD of a surviving box: the parent is cut in two along its widest side (widths weighted by the largest copy
// number they multiply), the child's bound decides whether it goes on -- to the next level's list, or, small enough, to the leaves.
__global__ __launch_bounds__(256) void mix_split_kernel(MixArgs A, const MixCell *in, unsigned long long n_in, MixCell *out, unsigned long long out_cap,
                                                        MixCell *leaves, unsigned long long leaf_cap, unsigned long long *counters) {
    __shared__ float2 rows[MIX_MAX_Q];
    __shared__ double2 ltab[128];                    // smx_log's table (the scorers' logarithm: 15 vector instructions, within an ulp)
    for (int s = threadIdx.x; s < A.Q; s += blockDim.x) rows[s] = make_float2((float)(A.rowtab[s] & 15u), (float)(A.rowtab[s] >> 4));
    smx_log_stage(ltab);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned long long k = (unsigned long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= 2 * n_in) return;            // (whole waves leave)
    MixCell c = in[k >> 1];
    int ax = 0;
    double wbest = -1.0;
    for (int j = 0; j < 3; j++) {
        const double w = (c.hi[j] - c.lo[j]) / A.leaf[j];
        if (w > wbest) {
            wbest = w;
            ax = j;
        }
    }
    const double mid = 0.5 * (c.lo[ax] + c.hi[ax]);
    if (k & 1) c.lo[ax] = mid; else c.hi[ax] = mid;
    const double lb = mix_cell_bound(A, c, rows, ltab, lane);
    if (!(lb <= A.thr) || lane != 0) return;
    c.lb = lb;
    bool leaf = true;
    for (int j = 0; j < 3; j++) leaf = leaf && (c.hi[j] - c.lo[j]) <= A.leaf[j];
    if (leaf) {
        const unsigned long long idx = atomicAdd(&counters[1], 1ull);
        if (idx < leaf_cap) leaves[idx] = c;
    } else {
        const unsigned long long idx = atomicAdd(&counters[0], 1ull);
        if (idx < out_cap) out[idx] = c;
    }
}

// The matrices of a leaf box: for corner k of the box, cost_i(c) = the tangent of phi_i at the centre, evaluated at the corner, bounds
// phi_i(c.v) from below on the whole box for the v that minimises the matrix's (linear) tangent sum -- which is a corner.  So every
// matrix whose objective is within `thr` somewhere in the box has sum_i cost_i(c_i) <= thr for at least one corner: a depth-first
// walk over the intervals (rows in slot order, valid and within bounds, the reference's edge rule between consecutive rows) with
// that budget and the suffix minima as look-ahead lists them.  One thread per (leaf, corner); a record is m slot bytes.
#define MIX_MAX_M 256
__global__ __launch_bounds__(64) void mix_list_kernel(MixArgs A, const MixCell *leaves, unsigned long long n_leaves, unsigned char *out, unsigned long long out_cap,
                                                      int per_thread_cap, unsigned long long *counters) {
    __shared__ float2 rows[MIX_MAX_Q];
    for (int s = threadIdx.x; s < A.Q; s += blockDim.x) rows[s] = make_float2((float)(A.rowtab[s] & 15u), (float)(A.rowtab[s] >> 4));
    __syncthreads();
    const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 8 * n_leaves) return;
    const MixCell c = leaves[k >> 3];
    const int corner = (int)(k & 7);
    const double tau = (double)A.tau;
    double vc[3], dv[3];
    for (int j = 0; j < 3; j++) {
        vc[j] = 0.5 * (c.lo[j] + c.hi[j]);
        const double h = 0.5 * (c.hi[j] - c.lo[j]);
        dv[j] = ((corner >> j) & 1) ? h : -h;
    }
    auto cost = [&](int i, int s, bool &ok) -> double {
        const float2 rw = rows[s];
        const int a = (int)rw.x, b = (int)rw.y, l = A.lb[i], u = A.ub[i];
        ok = !(a < l || a > u || b < l || b > u || (A.tau - a) * (A.tau - b) < 0);
        if (!ok) return __builtin_inf();
        const double x = (double)a, y = (double)b, r = A.r[i], N = A.rN[i];
        const double tc = tau * vc[0] + x * vc[1] + y * vc[2];
        if (!(tc > 0.0)) return -__builtin_inf();
        const double f0 = r > 0.0 ? N * tc - r * log(N * tc) : N * tc, f1 = N - r / tc;
        return f0 + f1 * (tau * dv[0] + x * dv[1] + y * dv[2]);
    };
    double suf[MIX_MAX_M + 1];
    suf[A.m] = A.cst;
    for (int i = A.m - 1; i >= 0; i--) {
        double best = __builtin_inf();
        for (int s = 0; s < A.Q; s++) {
            bool ok;
            const double v = cost(i, s, ok);
            if (ok) best = fmin(best, v);
        }
        suf[i] = suf[i + 1] + best;
    }
    if (!(suf[0] <= A.thr)) return;
    unsigned char cur[MIX_MAX_M];      // slot chosen at each depth (the next to try while descending)
    double part[MIX_MAX_M + 1];        // cost of the rows chosen above each depth
    int i = 0, found = 0;
    cur[0] = 0;
    part[0] = 0.0;
    while (i >= 0) {
        if ((int)cur[i] >= A.Q) {       // this depth is exhausted
            i--;
            if (i >= 0) cur[i]++;
            continue;
        }
        const int s = cur[i];
        bool ok;
        const double v = cost(i, s, ok);
        bool go = ok && part[i] + v + suf[i + 1] <= A.thr;
        if (go && i > 0) {               // Enumerator._is_valid_edge (Enumerator.py:258-260): the same row, or some component larger
            const float2 p = rows[cur[i - 1]], q = rows[s];
            go = (cur[i - 1] == s) || q.x > p.x || q.y > p.y;
        }
        if (!go) {
            cur[i]++;
            continue;
        }
        if (i == A.m - 1) {
            if (found < per_thread_cap) {
                const unsigned long long idx = atomicAdd(&counters[2], 1ull);
                if (idx < out_cap) {
                    unsigned char *dst = out + idx * (size_t)A.m;
                    for (int d = 0; d < A.m; d++) dst[d] = cur[d];
                }
            } else {
                atomicAdd(&counters[3], 1ull);      // (this walk found more than its share: the host must not trust the list)
            }
            found++;
            cur[i]++;
            continue;
        }
        part[i + 1] = part[i] + v;
        i++;
        cur[i] = 0;
    }
}

void mix_launch_split(const MixArgs &A, const MixCell *in, unsigned long long n_in, MixCell *out, unsigned long long out_cap, MixCell *leaves,
                      unsigned long long leaf_cap, unsigned long long *counters, hipStream_t st) {
    if (!n_in) return;
    hipLaunchKernelGGL(mix_split_kernel, dim3((unsigned)((2 * n_in + 3) / 4)), dim3(256), 0, st, A, in, n_in, out, out_cap, leaves, leaf_cap, counters);
}
void mix_launch_list(const MixArgs &A, const MixCell *leaves, unsigned long long n_leaves, unsigned char *out, unsigned long long out_cap, int per_thread_cap,
                     unsigned long long *counters, hipStream_t st) {
    if (!n_leaves) return;
    hipLaunchKernelGGL(mix_list_kernel, dim3((unsigned)((8 * n_leaves + 63) / 64)), dim3(64), 0, st, A, leaves, n_leaves, out, out_cap, per_thread_cap, counters);
}
