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

// The reference's successor (Enumerator.py:134-152) on the break-point form. Returns false at the end.
template <int KV>
N2_HD bool n2_next(const N2Dev &P, const unsigned char *ubl, const short *lbposl, N2Cand<KV> &c) {
    int e = -1, nv = 0;
#pragma unroll
    for (int v = 0; v < KV; v++) {
        if (e < 0 && c.s[v + 1] > c.s[v]) {  // first run not yet handled
            int end = c.s[v + 1] - 1;
            if (v < (int)ubl[end]) {
                e = end;
                nv = v + 1;
            } else if (end == P.m - 1) {
                return false;  // last run cannot be raised: enumeration exhausted
            }
        }
    }
    if (e < 0) return false;
#pragma unroll
    for (int w = 1; w < KV; w++) {
        int lp = lbposl[w];
        c.s[w] = (lp < e) ? lp : ((w <= nv) ? e : c.s[w]);
    }
    return true;
}
