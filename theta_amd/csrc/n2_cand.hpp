// n = 2 candidates in break-point form, their rank -> candidate unranking and the reference's successor.  Host and device
// (N2_HD): the kernels of n2.hip use them on the GPU, tools/n2_render_emul.hip runs the generator's per-lane logic on the CPU.
#pragma once
#include "n2.hpp"

#if defined(__HIPCC__)
#define N2_HD __host__ __device__
#else
#define N2_HD
#endif

template <int KV>
struct N2Cand {
    int s[KV + 1];  // s[v] = first position with c >= v; s[KV] = m
};

// Rank -> break-points (colex order: c_{m-1} is the most significant digit).
template <int KV>
N2_HD void n2_unrank(const N2Dev &P, const unsigned long long *Pl, unsigned long long rho, N2Cand<KV> &c) {
#pragma unroll
    for (int v = 0; v <= KV; v++) c.s[v] = P.m;
    c.s[0] = 0;
    for (int i = P.m - 1; i >= 0; i--) {
        const unsigned long long *row = Pl + i * N2_KVS;
        int v = 0;
        while (v < KV - 1 && row[v] <= rho) v++;  // smallest v with P[i][v] > rho
        if (v > 0) rho -= row[v - 1];
#pragma unroll
        for (int w = 1; w < KV; w++)
            if (w <= v) c.s[w] = i;  // c_i >= w, and i decreases, so the last write is the first position
    }
}

// The reference's successor (Enumerator.py:134-152) on the break-point form.  Returns the LEVEL of the step -- the value nv the
// raised position takes: break-points 1 .. nv move to that position, the others stay -- or 0 at the end of the space.
template <int KV>
N2_HD int n2_next_level(const N2Dev &P, const unsigned char *ubl, const short *lbposl, N2Cand<KV> &c) {
    int e = -1, nv = 0;
#pragma unroll
    for (int v = 0; v < KV; v++) {
        if (e < 0 && c.s[v + 1] > c.s[v]) {  // first run not yet handled
            int end = c.s[v + 1] - 1;
            if (v < (int)ubl[end]) {
                e = end;
                nv = v + 1;
            } else if (end == P.m - 1) {
                return 0;  // last run cannot be raised: enumeration exhausted
            }
        }
    }
    if (e < 0) return 0;
#pragma unroll
    for (int w = 1; w < KV; w++) {
        int lp = lbposl[w];
        c.s[w] = (lp < e) ? lp : ((w <= nv) ? e : c.s[w]);
    }
    return nv;
}
template <int KV>
N2_HD bool n2_next(const N2Dev &P, const unsigned char *ubl, const short *lbposl, N2Cand<KV> &c) {
    return n2_next_level<KV>(P, ubl, lbposl, c) != 0;
}

// The same successor with both bound tests on wave-uniform tables (the form of n2_render.hpp's n2r_next, returning the level):
//   lbp[w] = first position whose lower bound is >= w (m if none),  ubp[w] = first position whose upper bound is >= w (m if none).
// Run v = [s[v], s[v+1]) can be raised at its end iff that end lies at or behind ubp[v+1] (the bounds are non-decreasing after
// _check_bound_order), i.e. iff s[v+1] > max(s[v], ubp[v+1]): no per-lane bound reads, no data-dependent chain of LDS round trips.
template <int KV>
N2_HD inline int n2_next_tab(const int (&lbp)[KV + 1], const int (&ubp)[KV + 1], N2Cand<KV> &c) {
    int e = -1, nv = 0;
    bool found = false;
#pragma unroll
    for (int v = 0; v < KV; v++) {
        const int hi = c.s[v + 1], b = ubp[v + 1];
        const int lo = c.s[v] > b ? c.s[v] : b;
        const bool take = hi > lo && !found;
        e = take ? hi - 1 : e;
        nv = take ? v + 1 : nv;
        found = found || hi > lo;
    }
    if (!found) return 0;
#pragma unroll
    for (int w = 1; w < KV; w++) {
        const int lp = lbp[w];
        c.s[w] = w <= nv ? (lp < e ? lp : e) : c.s[w];
    }
    return nv;
}
