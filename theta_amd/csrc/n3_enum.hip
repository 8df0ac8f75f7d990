// n = 3 materialised generator for gfx950 (theta_enumerate / theta_enumerate_device): repeated
// Enumerator.generate_next_C() (Enumerator.py:74-87, 172-242) for a rank range, candidates written to HBM
// as m x 2 bytes each, in the reference's DFS order.
//
// HBM-write bound: 2 m bytes per candidate and nothing else.  The write path decides everything, so the
// kernel is built around it: ONE wave owns ONE contiguous output stream and writes it front to back in
// lane-consecutive, 16-byte-aligned stores (a wave store is one contiguous kilobyte).  To do that the wave
// expands the last LB <= 6 rows of the matrices breadth-first, level by level, with all 64 lanes working on
// 64 nodes of the same level (no lane-private DFS, no per-lane output position):
//
//   rows 0 .. D-1 (D = m - LB)  wave-uniform "prefix"; lane d holds the packed DFS node of depth d; its rows sit
//                               in LDS as 16-bit units {a, b} and are the same for every record of the prefix
//   level l = 0 .. LB-1         the nodes of depth D+l-1 are taken 64 at a time; a lane computes its node's child
//                               mask (static rules & symmetry & ratio window: two table reads), the wave scans the
//                               child counts and the lanes write their children -- as {parent, slot} pairs -- to the
//                               next level's list in LDS.  Lists are bounded (128 nodes, 512 records): a round takes
//                               as many nodes as fit, so the walk is a DFS over rounds and keeps the rank order.
//   last level                  children are records: the list holds the LB leaf rows of each, and the wave "fills"
//                               the burst  records x (prefix units from LDS | leaf units from the list)  straight
//                               into global memory, 16 bytes per lane, consecutive lanes consecutive addresses.
//
// Units are 32-bit row pairs when m is even (LB even, so that the prefix is whole units) and 16-bit rows when
// m is odd.  LB is 4 -- 6 when the instance branches so little that four rows give a prefix fewer than ~600 records
// (a round of 64 nodes should be a full round) -- and 2 / 1 for m < 6.  Task boundaries (a task starts `skip` leaves into its first prefix and ends after `count`
// candidates) are a window on the record index of a burst.
#include <stddef.h>

#include <type_traits>

#include "n3_core.hpp"

#define EB_WAVES 4
#ifndef EB_CAP
#define EB_CAP 128        // nodes per intermediate level list
#endif
#ifndef EB_OCC4
#define EB_OCC4 5         // waves per SIMD the register budget is sized for, LB <= 4 (LDS: 32 KB per 4-wave block)
#endif
#ifndef EB_OCC6
#define EB_OCC6 4         // ... LB = 6 (LDS: 40 KB per block)
#endif

typedef unsigned eb_u4v __attribute__((ext_vector_type(4)));

// records per burst (leaf list): sized so that the block's LDS allows the occupancy above
template <int ML> struct EbCapR { static constexpr int v = ML <= 4 ? 512 : 288; };

template <int ML>
struct EbWave {
    alignas(16) unsigned short pre[N3_MAX_M_WIDE + 8]; // rows of the prefix, a | b << 8 (up to 128 intervals: two per lane)
    uint2 list0[N3_MAX_Q];                             // level 1 nodes (children of the prefix's last node)
    uint2 list[ML > 2 ? ML - 2 : 1][EB_CAP];           // level l >= 2: {packed parent node, ancestor slots (6 bits each) | slot << 24}
    alignas(8) unsigned short lwr[(EbCapR<ML>::v + 4) * ML];   // records of the current burst: their LB last rows, a | b << 8
};
template <int ML>
struct EbLds {
    EbWave<ML> w[EB_WAVES];
    unsigned long long smask[ML][N3_MAX_Q];             // static child masks of the LB last depths
    unsigned char lb[N3_MAX_M_WIDE], ub[N3_MAX_M_WIDE];
    unsigned char ridx[N3_RIDX_W * N3_RIDX_W + 3];
    unsigned char rowtab[N3_MAX_Q + 3];
    unsigned short row16[N3_MAX_Q];                     // slot -> a | b << 8
};

template <int ML>
struct EbCtx {
    EbLds<ML> *S;
    EbWave<ML> *W;
    const unsigned long long *dynmask;
    unsigned long long swm;
    int NT1, lane, LB;
    N3State par;
    // output stream of the task
    unsigned char *out, *obase;
    unsigned long long written, remaining, skip;
    int NU, NPU;              // units per record, units of the prefix
    unsigned magic;           // ceil(2^32 / NU)
    size_t RS;                // bytes per record
};

// inclusive scan over the wave with DPP row shifts / broadcasts (no LDS traffic)
__device__ __forceinline__ int eb_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

template <int ML>
__device__ __forceinline__ unsigned long long eb_child_mask(const EbCtx<ML> &c, const N3State &node, int l) {
    unsigned long long mk = c.S->smask[l][node.slot] & c.dynmask[((size_t)node.slot * c.NT1 + node.lo) * c.NT1 + (node.hi - 1)];
    return node.sw ? (mk & c.swm) : mk;
}

// Write records [0, total) of the leaf list (less the task window) to the wave's output stream: the burst is the
// periodic pattern  (NPU prefix units from `pre` | NU - NPU leaf units from `lwr`)  per record; every lane builds
// 16-byte-aligned chunks of it (one LDS read per unit, the address picks the source) and stores them whole.
template <int U, int ML>
__device__ __forceinline__ void eb_emit(EbCtx<ML> &c, int total) {
    const unsigned long long sk = c.skip < (unsigned long long)total ? c.skip : (unsigned long long)total;
    c.skip -= sk;
    const int lo_c = (int)sk;
    const unsigned long long room = (unsigned long long)(total - lo_c);
    const int nrec = (int)(room < c.remaining ? room : c.remaining);
    if (nrec <= 0) return;
    constexpr int UB = 16 / U;                                   // bytes per unit
    typedef typename std::conditional<U == 4, unsigned, unsigned short>::type unit_t;
    const int NU = c.NU, NPU = c.NPU, NLU = NU - NPU;
    const int nunits = nrec * NU;
    const uintptr_t A0 = (uintptr_t)c.obase + (size_t)c.written * c.RS;
    const uintptr_t A1 = A0 + (size_t)nrec * c.RS;
    const uintptr_t q0 = A0 >> 4, q1 = (A1 + 15) >> 4;
    const unit_t *wu = (const unit_t *)c.W;                      // the wave's LDS block as units
    constexpr int PRE_OFF = (int)(offsetof(EbWave<ML>, pre) / UB), LW_OFF = (int)(offsetof(EbWave<ML>, lwr) / UB);
#ifdef EB_NOFILL
    if (nunits < 0)
#endif
    for (uintptr_t q = q0 + (uintptr_t)c.lane; q < q1; q += WAVE) {
        const uintptr_t ca = q << 4;
        unsigned char *const cp = c.out + ((long long)ca - (long long)(uintptr_t)c.out);
        const int rel0 = (int)(((long long)ca - (long long)A0) / UB);   // unit index of the chunk's first unit (head chunk: < 0)
        const int cur = rel0 + 4 * NU;                                  // >= 0
        const int rq = (int)__umulhi((unsigned)cur, c.magic);           // record + 4
        int pos = cur - rq * NU;
        int lwo = LW_OFF + (lo_c + rq - 4) * NLU - NPU;                 // leaf unit j of the record: wu[lwo + NPU + j]
        unsigned val[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            val[u] = (unsigned)wu[pos < NPU ? PRE_OFF + pos : lwo + pos];
            pos++;
            if (pos == NU) {
                pos = 0;
                lwo += NLU;
            }
        }
#ifdef EB_NOSTORE
        if (val[0] != 0xdeadbeefu) continue;
#endif
        if (rel0 >= 0 && rel0 + U <= nunits) {
            eb_u4v o;
            if (U == 4) o = eb_u4v{val[0], val[1], val[2], val[3]};
            else
                o = eb_u4v{val[0] | (val[1 % U] << 16), val[2 % U] | (val[3 % U] << 16), val[4 % U] | (val[5 % U] << 16),
                           val[6 % U] | (val[7 % U] << 16)};
            *(eb_u4v *)cp = o;
        } else {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int rel = rel0 + u;
                if (rel >= 0 && rel < nunits) ((unit_t *)cp)[u] = (unit_t)val[u];
            }
        }
    }
    c.written += (unsigned long long)nrec;
    c.remaining -= (unsigned long long)nrec;
}

// Expand the nodes of leaf level LVL (LVL = 0: the prefix's last node; else the level's list [0 .. n_in)) in rank order.
template <int U, int ML, int LVL>
__device__ void eb_expand(EbCtx<ML> &c, int n_in) {
    const bool last = (LVL == c.LB - 1);
    const int cap = last ? EbCapR<ML>::v : (LVL == 0 ? N3_MAX_Q : EB_CAP);
    int pos = 0;
    while (pos < n_in) {
        const int i = pos + c.lane;
        bool live = i < n_in;
        N3State node = c.par;
        unsigned code = 0;                                 // slots of rows D .. D+LVL-1, 6 bits each
        if (LVL > 0) {
            if (live) {
                const uint2 e = (LVL == 1) ? c.W->list0[i] : c.W->list[LVL >= 2 ? LVL - 2 : 0][i];
                const N3State pst = n3_unpack(e.x);
                const unsigned slot = e.y >> 24;
                n3_child_dyn(c.S->ridx, c.S->rowtab, pst, (int)slot, node);
                code = (e.y & 0xffffffu) | (slot << (6 * (LVL > 0 ? LVL - 1 : 0)));
            }
        } else {
            live = c.lane == 0;
        }
        unsigned long long mk = live ? eb_child_mask(c, node, LVL) : 0ull;
        const int cnt = __builtin_popcountll(mk);
        const int incl = eb_incl_scan(cnt);
        const int t = __builtin_popcountll(ballot64(live && incl <= cap));   // nodes whose children all fit (a prefix of the lanes)
        const int total = __builtin_amdgcn_readlane(incl, t - 1);
        const int off = incl - cnt;
        const bool take = live && c.lane < t;
        if (last) {
            if (take) {
                // rows 0 .. LVL-1 of the record are the node's ancestors and itself (the same for all its children)
                constexpr int LBc = LVL + 1;
                if constexpr ((LBc & 1) == 0) {
                    unsigned un[LBc / 2];
#pragma unroll
                    for (int j = 0; j < LBc / 2; j++) {
                        un[j] = c.S->row16[(code >> (12 * j)) & 63u];
                        if (j < LBc / 2 - 1) un[j] |= (unsigned)c.S->row16[(code >> (12 * j + 6)) & 63u] << 16;
                    }
                    unsigned *dst = (unsigned *)c.W->lwr + off * (LBc / 2);
                    while (mk) {
                        const int s = __builtin_ctzll(mk);
                        mk &= mk - 1;
#pragma unroll
                        for (int j = 0; j < LBc / 2 - 1; j++) dst[j] = un[j];
                        dst[LBc / 2 - 1] = un[LBc / 2 - 1] | ((unsigned)c.S->row16[s] << 16);
                        dst += LBc / 2;
                    }
                } else {   // (LB = 1: one expanded row; other odd LB are never chosen)
                    unsigned short *dst = c.W->lwr + off * LBc;
                    while (mk) {
                        const int s = __builtin_ctzll(mk);
                        mk &= mk - 1;
#pragma unroll
                        for (int j = 0; j < LBc - 1; j++) dst[j] = c.S->row16[(code >> (6 * j)) & 63u];
                        dst[LBc - 1] = c.S->row16[s];
                        dst += LBc;
                    }
                }
            }
            wave_lds_sync();
            eb_emit<U, ML>(c, total);
            wave_lds_sync();
        } else {
            if constexpr (LVL + 1 < ML) {
                const unsigned ps = n3_pack(node);
                if (take) {
                    uint2 *dst = ((LVL == 0) ? c.W->list0 : c.W->list[LVL >= 1 ? LVL - 1 : 0]) + off;
                    while (mk) {
                        const int s = __builtin_ctzll(mk);
                        mk &= mk - 1;
                        *dst++ = make_uint2(ps, code | ((unsigned)s << 24));
                    }
                }
                wave_lds_sync();
                eb_expand<U, ML, LVL + 1>(c, total);
                wave_lds_sync();
            }
        }
        if (c.remaining == 0) return;
        pos += t;
    }
}

#define EB_NS 4            // prefix intervals per lane (n3_core.hpp: n3_lane_state / n3_next_prefix): up to 256 + LB intervals

template <int U, int ML>
__global__ __launch_bounds__(64 * EB_WAVES, (ML <= 4 ? EB_OCC4 : EB_OCC6)) void n3_enumerate_burst_kernel(
    N3Dev Pg, const N3Task *tasks, const unsigned *stbuf, int ntasks, uint64_t per_task, unsigned char *out) {
    __shared__ EbLds<ML> S;
    const int m = Pg.m, LB = Pg.L, D = m - LB, Q = Pg.Q;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        S.lb[i] = Pg.lb[i];
        S.ub[i] = Pg.ub[i];
    }
    for (int i = threadIdx.x; i < N3_RIDX_W * N3_RIDX_W; i += blockDim.x) S.ridx[i] = Pg.ridx[i];
    for (int i = threadIdx.x; i < Q; i += blockDim.x) {
        const unsigned rw = Pg.rowtab[i];
        S.rowtab[i] = (unsigned char)rw;
        S.row16[i] = (unsigned short)((rw & 15u) | ((rw >> 4) << 8));
    }
    for (int i = threadIdx.x; i < LB * N3_MAX_Q; i += blockDim.x) (&S.smask[0][0])[i] = Pg.smask[(size_t)D * N3_MAX_Q + i];
    __syncthreads();
    N3Dev P = Pg;
    P.lb = S.lb;
    P.ub = S.ub;
    P.ridx = S.ridx;
    P.rowtab = S.rowtab;

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * EB_WAVES + wv;
    if (task >= ntasks) return;
    const N3Task tk = tasks[task];
    unsigned st[EB_NS];                                // lane l: the packed prefix nodes of depths l, 64 + l, ...
#pragma unroll
    for (int j = 0; j < EB_NS; j++) st[j] = WAVE * j + lane < D ? stbuf[(size_t)task * N3_STB + WAVE * j + lane] : 0u;

    EbCtx<ML> c;
    c.S = &S;
    c.W = &S.w[wv];
    c.dynmask = Pg.dynmask;
    c.swm = Pg.swmask;
    c.NT1 = Pg.NT + 1;
    c.lane = lane;
    c.LB = LB;
    c.RS = (size_t)m * 2;
    c.out = out;
    c.obase = out + (size_t)task * per_task * c.RS;       // tasks are consecutive rank ranges of per_task candidates
    c.written = 0;
    c.remaining = tk.count;
    c.skip = tk.skip;
    c.NU = (U == 4) ? m / 2 : m;
    c.NPU = (U == 4) ? D / 2 : D;
    c.magic = (unsigned)((0x100000000ull + (unsigned)c.NU - 1) / (unsigned)c.NU);

    while (c.remaining > 0) {
#pragma unroll
        for (int j = 0; j < EB_NS; j++)
            if (WAVE * j + lane < D) c.W->pre[WAVE * j + lane] = (unsigned short)(((st[j] >> 24) & 15u) | ((st[j] >> 28) << 8));
        wave_lds_sync();
        c.par = n3_unpack(n3_lane_state<EB_NS>(st, D - 1));
        eb_expand<U, ML, 0>(c, 1);
        c.skip = 0;                                        // only the first prefix of a task starts mid-way
        if (c.remaining == 0) break;
        if (!n3_next_prefix<EB_NS>(P, st, D, lane)) break;
        wave_lds_sync();                                   // the prefix rows in LDS are rewritten next
    }
}

// Leaf levels of the burst generator for an instance: even for even m (whole 32-bit units in the prefix), at least
// one prefix row; six instead of four when four rows would give a prefix fewer than ~600 records on average (growth
// factor of the candidate count per interval ^ 4).  0: m too small (the lane-private generator of n3.hip is used).
int n3_enumerate_burst_levels(const N3Dev &P) {
    const int m = P.m;
    if (m < 2) return 0;
    int want = 4;
    if (m >= 8) {
        const double lg = log((double)P.total_hi * 18446744073709551616.0 + (double)P.total_lo) / (double)m;
        if (lg > 0.0 && 4.0 * lg < log(600.0)) want = 6;
    }
    if (const char *e = getenv("THETA_ENUM_LEVELS")) {
        const int v = atoi(e);
        if (v >= 1 && v <= 6) want = v;
    }
    if (want > m - 1) want = m - 1;
    if ((m & 1) == 0) {                 // 32-bit units: LB even
        if (m < 4) return 1;            // (m = 2: 16-bit units)
        want &= ~1;
        if (want < 2) want = 2;
    } else if (want == 3 || want == 5) want--;   // odd LB only as LB = 1
    return want;
}

// P.L must be n3_enumerate_burst_levels(P) -- for the task kernel as well (tasks are cut at depth m - P.L).
void n3_launch_enumerate_burst(const N3Dev &P, const N3Task *tasks, const unsigned *stbuf, int ntasks, uint64_t per_task,
                               unsigned char *out, hipStream_t st) {
    dim3 grid((ntasks + EB_WAVES - 1) / EB_WAVES), block(64 * EB_WAVES);
    const bool words = (P.m & 1) == 0 && P.m >= 4;
#define EB_LAUNCH(UU, MLL) hipLaunchKernelGGL((n3_enumerate_burst_kernel<UU, MLL>), grid, block, 0, st, P, tasks, stbuf, ntasks, per_task, out)
    if (P.L <= 4) {
        if (words) EB_LAUNCH(4, 4); else EB_LAUNCH(8, 4);
    } else {
        if (words) EB_LAUNCH(4, 6); else EB_LAUNCH(8, 6);
    }
#undef EB_LAUNCH
}

// ------------------------------------------------------------------------------------------------
// Rank-deficient candidates of a materialised rank range (n3_core.hpp: N3Line): one thread per record walks its rows through the
// collinearity test and appends the rank of a record whose rows all lie on one line to the degenerate list.  Used by the
// sieve path of theta_search on the (rare) tasks that met a prefix with collinear rows -- api.hip: list_deficient.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void n3_collinear_scan_kernel(const unsigned char *C, unsigned long long count, int m, uint64_t base_lo,
                                                                uint64_t base_hi, SearchCounters *ctr, TieRecord *deg, unsigned deg_cap) {
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const unsigned char *row = C + (size_t)k * 2 * m;
    N3Line ln = {0, 0, 0, 0, 0};
    for (int i = 0; i < m && ln.kind < 3; i++) n3_line_add(ln, (int)row[2 * i], (int)row[2 * i + 1]);
    if (ln.kind < 3) degenerate_append(ctr, deg, deg_cap, (((u128)base_hi << 64) | base_lo) + k);
}

void n3_launch_collinear_scan(const unsigned char *C, unsigned long long count, int m, u128 base, SearchCounters *ctr, TieRecord *deg,
                              unsigned deg_cap, hipStream_t st) {
    if (count == 0) return;
    hipLaunchKernelGGL(n3_collinear_scan_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, C, count, m, (uint64_t)base,
                       (uint64_t)(base >> 64), ctr, deg, deg_cap);
}

// ------------------------------------------------------------------------------------------------
// The sweep for NaN outcomes (api.hip: nan_sweep): after the reference's own per-candidate procedure (solve_batch_n3_kernel) has
// run over a materialised rank range, append the rank of every candidate the reference REPORTS (ok != 0) with a NaN likelihood
// to the degenerate list -- such a tuple joins `best` wherever it stands (Misc.py:44-46) -- and of every candidate it reports at or
// below `near` (the search's minimum + the collection window): whatever the search kernels made of those, the swept range's
// result is then the replay over the procedure's own outcomes by construction.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void n3_nan_scan_kernel(const unsigned char *ok, const double *nll, unsigned long long count, uint64_t base_lo,
                                                          uint64_t base_hi, double near, SearchCounters *ctr, TieRecord *deg, unsigned deg_cap) {
    const unsigned long long k = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= count) return;
    const double v = nll[k];
    if (ok[k] && (v != v || v <= near)) degenerate_append(ctr, deg, deg_cap, (((u128)base_hi << 64) | base_lo) + k);
}

void n3_launch_nan_scan(const unsigned char *ok, const double *nll, unsigned long long count, u128 base, double near, SearchCounters *ctr,
                        TieRecord *deg, unsigned deg_cap, hipStream_t st) {
    if (count == 0) return;
    hipLaunchKernelGGL(n3_nan_scan_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, ok, nll, count, (uint64_t)base,
                       (uint64_t)(base >> 64), near, ctr, deg, deg_cap);
}
