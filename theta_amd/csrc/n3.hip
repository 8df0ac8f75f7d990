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
#include <type_traits>

#include "n3_core.hpp"
#define HYBRJ4_MANAGE_CONTRACT     // this unit allows fused multiply-adds; the hybrj restatement must not use them
#include "n3_refbfgs.hpp"

// ------------------------------------------------------------------------------------------------
// host: bounds, ratio table
// ------------------------------------------------------------------------------------------------
int n3_build_host(int m, int tau, const int32_t *lb_in, const int32_t *ub_in, N3Host &h) {
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
    if (top > N3_MAX_COPY) {
        theta_set_error("n=3 search supports copy numbers up to %d (max upper bound is %d)", N3_MAX_COPY, top);
        return THETA_ERR_ARG;
    }
    h.K = top;                    // Enumerator.py:58: k = max(upper_bound)
    // The row alphabet (n3_core.hpp): the whole grid up to K = 7; beyond, the valid rows within the bounds of some interval,
    // in grid order (a fastest: the order of Enumerator._create_graph, Enumerator.py:272-298).
    const int K1g = top + 1;
    std::vector<int> rows_a, rows_b;
    for (int s = 0; s < K1g * K1g; s++) {
        const int a = s % K1g, b = s / K1g;
        bool use = top <= N3_GRID_K;
        if (!use && n3_valid_row(a, b, tau))
            for (int i = 0; i < m && !use; i++) use = a >= h.lb[i] && a <= h.ub[i] && b >= h.lb[i] && b <= h.ub[i];
        if (use) {
            rows_a.push_back(a);
            rows_b.push_back(b);
        }
    }
    h.Q = (int)rows_a.size();
    if (h.Q > N3_MAX_Q) {
        // more rows than a child mask holds (one 64-bit word: the rank-walking kernels).  Round 5: such a problem still has its row
        // table -- the valid rows in grid order -- and is served by the search that needs no ranks (theta_mix_search, bnb.hip)
        if (h.Q > 255) {
            theta_set_error("n=3 search with copy numbers up to %d: %d distinct rows", top, h.Q);
            return THETA_ERR_ARG;
        }
        h.mix_only = true;
        h.rowtab.assign(h.Q, 0);
        for (int s = 0; s < h.Q; s++) h.rowtab[s] = (unsigned char)(rows_a[s] | (rows_b[s] << 4));
        h.NT = 0;
        return THETA_OK;
    }
    // distinct values of dy/(-dx) over the differences of two rows of the alphabet, sorted ascending; compared by cross-multiplication
    struct Fr { int n, d; };
    std::vector<Fr> fr;
    std::vector<unsigned char> seen(N3_RIDX_W * N3_RIDX_W, 0);
    for (int p = 0; p < h.Q; p++)
        for (int s = 0; s < h.Q; s++) {
            const int dx = rows_a[s] - rows_a[p], dy = rows_b[s] - rows_b[p];
            if (dx == 0 || dy == 0 || seen[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)]) continue;
            seen[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)] = 1;
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
            if (dx == 0 || dy == 0 || !seen[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)]) continue;
            Fr v{dy, -dx};
            if (v.d < 0) { v.n = -v.n; v.d = -v.d; }
            int idx = (int)(std::lower_bound(fr.begin(), fr.end(), v, less) - fr.begin());
            h.ridx[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)] = (unsigned char)(idx + 1);
        }
    // slot -> row; for every depth d and parent row the set of rows that may follow it by the static rules
    // (valid row, bounds of depth d, Enumerator._is_valid_edge); and for every (parent row, lo, hi) the set of
    // rows that keep the ratio window non-empty (Enumerator._get_mu_bounds + the lo <= hi test, :204-212).
    const int NT1 = h.NT + 1;
    h.rowtab.assign(h.Q, 0);
    for (int s = 0; s < h.Q; s++) {
        const int a = rows_a[s], b = rows_b[s];
        h.rowtab[s] = (unsigned char)(a | (b << 4));
        if (a <= b) h.swmask |= 1ull << s;
    }
    // (Round 6: both tables from per-row masks -- the static rule of an edge does not depend on the depth but for the bounds, and the
    // ratio window keeps a row s iff its threshold t_s <= hi (dx > 0) resp. >= lo (dx < 0): m Q + Q NT^2 mask operations instead of
    // m Q^2 + Q^2 NT^2 edge tests.  At m = 200, K = 7 the old loops were 11 ms of host time, a quarter of the whole-space search.
    // THETA_N3_TABLES_CHECK=1 builds them the old way as well and compares.)
    h.smask.assign((size_t)m * N3_MAX_Q, 0ull);
    {
        std::vector<unsigned long long> edge(h.Q, 0ull), inb(m, 0ull);
        unsigned long long valid = 0ull;
        for (int s = 0; s < h.Q; s++)
            if (n3_valid_row(rows_a[s], rows_b[s], tau)) valid |= 1ull << s;
        for (int ps = 0; ps < h.Q; ps++)
            for (int s = 0; s < h.Q; s++)
                if (s == ps || rows_a[s] > rows_a[ps] || rows_b[s] > rows_b[ps]) edge[ps] |= 1ull << s;
        for (int d = 0; d < m; d++) {
            unsigned long long mk = 0ull;
            for (int s = 0; s < h.Q; s++) {
                const int a = rows_a[s], b = rows_b[s];
                if (a >= h.lb[d] && a <= h.ub[d] && b >= h.lb[d] && b <= h.ub[d]) mk |= 1ull << s;
            }
            inb[d] = mk & valid;
            for (int ps = 0; ps < h.Q; ps++) h.smask[(size_t)d * N3_MAX_Q + ps] = edge[ps] & inb[d];
        }
    }
    h.dynmask.assign((size_t)h.Q * NT1 * NT1, 0ull);
    {
        std::vector<unsigned long long> up(h.NT + 2), dn(h.NT + 2);
        for (int ps = 0; ps < h.Q; ps++) {
            const int pa = rows_a[ps], pb = rows_b[ps];
            unsigned long long free_rows = 0ull;
            std::fill(up.begin(), up.end(), 0ull);
            std::fill(dn.begin(), dn.end(), 0ull);
            for (int s = 0; s < h.Q; s++) {
                const int dx = rows_a[s] - pa, dy = rows_b[s] - pb;
                if (dx == 0 || dy == 0) {
                    free_rows |= 1ull << s;
                    continue;
                }
                const int t = h.ridx[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)];      // 1 .. NT
                if (dx > 0) up[t] |= 1ull << s; else dn[t] |= 1ull << s;
            }
            for (int t = 1; t <= h.NT + 1; t++) up[t] |= up[t - 1];                              // rows with dx > 0 and t_s <= t
            for (int t = h.NT; t >= 0; t--) dn[t] |= dn[t + 1];                                  // rows with dx < 0 and t_s >= t
            for (int lo = 0; lo <= h.NT; lo++)
                for (int hi = std::max(lo, 1); hi <= h.NT + 1; hi++)
                    h.dynmask[((size_t)ps * NT1 + lo) * NT1 + (hi - 1)] = free_rows | up[hi] | dn[lo];
        }
    }
    if (const char *chk = getenv("THETA_N3_TABLES_CHECK")) {
        if (atoi(chk) != 0) {
            for (int d = 0; d < m; d++)
                for (int ps = 0; ps < h.Q; ps++) {
                    const int pa = rows_a[ps], pb = rows_b[ps];
                    unsigned long long mk = 0ull;
                    for (int s = 0; s < h.Q; s++) {
                        const int a = rows_a[s], b = rows_b[s];
                        if (n3_valid_row(a, b, tau) && a >= h.lb[d] && a <= h.ub[d] && b >= h.lb[d] && b <= h.ub[d] && (s == ps || a > pa || b > pb))
                            mk |= 1ull << s;
                    }
                    if (mk != h.smask[(size_t)d * N3_MAX_Q + ps]) {
                        theta_set_error("n3_build_host: static mask differs at depth %d, row %d", d, ps);
                        return THETA_ERR_ARG;
                    }
                }
            for (int ps = 0; ps < h.Q; ps++) {
                const int pa = rows_a[ps], pb = rows_b[ps];
                for (int lo = 0; lo <= h.NT; lo++)
                    for (int hi = 1; hi <= h.NT + 1; hi++) {
                        if (lo > hi) continue;
                        unsigned long long mk = 0;
                        for (int s = 0; s < h.Q; s++) {
                            const int dx = rows_a[s] - pa, dy = rows_b[s] - pb;
                            int l2 = lo, h2 = hi;
                            if (dx != 0 && dy != 0) {
                                const int t = h.ridx[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)];
                                if (dx > 0) l2 = std::max(lo, t); else h2 = std::min(hi, t);
                            }
                            if (l2 <= h2) mk |= 1ull << s;
                        }
                        if (mk != h.dynmask[((size_t)ps * NT1 + lo) * NT1 + (hi - 1)]) {
                            theta_set_error("n3_build_host: ratio-window mask differs at row %d, window (%d, %d)", ps, lo, hi);
                            return THETA_ERR_ARG;
                        }
                    }
            }
        }
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
            N3State par{slot, sw, lo, hi, (int)(P.rowtab[slot] & 15u), (int)(P.rowtab[slot] >> 4)}, ch;
            for (int c = 0; c < P.Q; c++) {
                if (n3_edge(P, par, c, d + 1, ch)) {
                    // SATURATING: a count of 2^128 - 1 stands for "that many or more".  rank -> path (n3_unrank) takes the first child
                    // whose count exceeds what is left of the rank, so every rank below 2^128 of such a space resolves exactly -- the
                    // counts that matter below a task's prefix are small --, and rank ranges of a space whose TOTAL overflows 128 bits
                    // (m = 100, K = 7 with full bounds: 1e75 matrices) are searched like any other.  `overflow` only says it happened.
                    u128 v = cnt[n3_cnt_index(P, d + 1, ch.slot, ch.sw, ch.lo, ch.hi)];
                    u128 ns = sum + v;
                    if (ns < sum || v == ~(u128)0) {
                        atomicOr(overflow, 1u);
                        ns = ~(u128)0;
                    }
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
            if (ns < sum || v == ~(u128)0) {
                atomicOr(overflow, 1u);
                ns = ~(u128)0;
            }
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

// Wave-cooperative rank -> DFS path: at every level the 64 lanes test the 64 alphabet slots and read their
// children's counts in ONE memory round trip (the serial walk above chains ~5 dependent HBM reads per level);
// the wave then scans the feasible children in slot order.  Lane d ends up holding the packed node of depth d.
// (the packed node of depth d is written to states[d] by lane 0 -- `states` may be global or shared memory).  The walk may start
// below the root: depths [d0, depth) under the node `par` of depth d0 - 1, rho = the rank within that node's subtree.  With
// `trail` the walk also records, per depth, the rank left inside the chosen node's subtree and that subtree's size
// (trail[2 d], trail[2 d + 1]: what n3_task_kernel needs to start other ranks of the same neighbourhood half-way down).
__device__ bool n3_unrank_wave(const N3Dev &P, u128 rho, int depth, int lane, unsigned *states, u128 &rem, int d0 = 0,
                               N3State par = N3State{0, 0, 0, 0, 0, 0}, u128 *trail = nullptr) {
    const unsigned myrow = lane < P.Q ? P.rowtab[lane] : 0u;       // one alphabet slot per lane
    const int sa = (int)(myrow & 15u), sb = (int)(myrow >> 4);
    for (int d = d0; d < depth; d++) {
        N3State nx{0, 0, 0, 0, 0, 0};
        bool ok = lane < P.Q && (d == 0 ? n3_first_row_ab(P, sa, sb, lane, nx) : n3_edge_ab(P, par, sa, sb, lane, d, nx));
        u128 v = 0;
        if (ok) v = P.cnt[n3_cnt_index(P, d, nx.slot, nx.sw, nx.lo, nx.hi)];
        unsigned v0 = (unsigned)v, v1 = (unsigned)(v >> 32), v2 = (unsigned)(v >> 64), v3 = (unsigned)(v >> 96);
        unsigned long long mk = ballot64(ok);
        int chosen = -1;
        u128 vsel = 0;
        while (mk) {
            int b = __builtin_ctzll(mk);
            mk &= mk - 1;
            u128 vb = ((u128)(unsigned)__builtin_amdgcn_readlane((int)v3, b) << 96) |
                      ((u128)(unsigned)__builtin_amdgcn_readlane((int)v2, b) << 64) |
                      ((u128)(unsigned)__builtin_amdgcn_readlane((int)v1, b) << 32) |
                      (u128)(unsigned)__builtin_amdgcn_readlane((int)v0, b);
            if (rho < vb) {
                chosen = b;
                vsel = vb;
                break;
            }
            rho -= vb;
        }
        if (chosen < 0) return false;
        unsigned mine = ok ? n3_pack(nx) : 0u;
        unsigned packed = (unsigned)__builtin_amdgcn_readlane((int)mine, chosen);
        par = n3_unpack(packed);
        if (lane == 0) {
            states[d] = packed;
            if (trail) {
                trail[2 * d] = rho;
                trail[2 * d + 1] = vsel;
            }
        }
    }
    rem = rho;
    return true;
}

// The path of the range's FIRST rank, with its trail (one wave): the tasks of a call are consecutive rank ranges, so their paths
// share all but the last few levels with it -- 2^31 candidates of the bench's space span twelve levels of forty-four.
// anchor: [N3_STB] packed nodes, then [2 N3_STB] u128 trail, then one flag word (1 = valid)
#define N3_ANCHOR_WORDS (N3_STB + 8 * N3_STB + 4)
__global__ __launch_bounds__(64) void n3_anchor_kernel(N3Dev P, uint64_t b_lo, uint64_t b_hi, unsigned *anchor) {
    const int lane = threadIdx.x & 63, D = P.m - P.L;
    u128 rem = 0;
    const bool ok = n3_unrank_wave(P, ((u128)b_hi << 64) | b_lo, D, lane, anchor, rem, 0, N3State{0, 0, 0, 0, 0, 0}, (u128 *)(anchor + N3_STB));
    if (lane == 0) anchor[N3_STB + 8 * N3_STB] = ok ? 1u : 0u;
}

// one wave per task: where does the task start?  From the deepest node of the anchor path whose subtree still holds the task's
// first rank (rank left in the subtree + the task's offset < the subtree's size: true for every shallower node too), not from the root.
__global__ __launch_bounds__(256) void n3_task_kernel(N3Dev P, uint64_t b_lo, uint64_t b_hi, uint64_t e_lo, uint64_t e_hi,
                                                      uint64_t per_task, int ntasks, N3Task *tasks, unsigned *stbuf, const unsigned *anchor) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= ntasks) return;
    const u128 begin = ((u128)b_hi << 64) | b_lo, end = ((u128)e_hi << 64) | e_lo;
    const u128 off = (u128)t * per_task;
    u128 base = begin + off;
    u128 left = end - base;
    uint64_t count = left < (u128)per_task ? (uint64_t)left : per_task;
    const int D = P.m - P.L;
    unsigned *states = stbuf + (size_t)t * N3_STB;
    u128 rem = 0;
    int d0 = 0;
    N3State par{0, 0, 0, 0, 0, 0};
    u128 rho = base;
    if (anchor && anchor[N3_STB + 8 * N3_STB]) {
        const u128 *trail = (const u128 *)(anchor + N3_STB);
        int deepest = -1;                                  // deepest depth whose anchor node still holds this task's first rank
        for (int d = lane; d < D; d += WAVE) {
            const u128 r = trail[2 * d], sz = trail[2 * d + 1], sum = r + off;
            if (sum >= r && sum < sz) deepest = d;         // (sum >= r: no wrap-around of the 128-bit addition)
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(deepest, o, WAVE);
            deepest = other > deepest ? other : deepest;
        }
        if (deepest >= 0) {
            for (int d = lane; d <= deepest; d += WAVE) states[d] = anchor[d];
            par = n3_unpack(anchor[deepest]);
            rho = trail[2 * deepest] + off;
            d0 = deepest + 1;
        }
    }
    bool ok = n3_unrank_wave(P, rho, D, lane, states, rem, d0, par);
    if (lane == 0) {
        N3Task tk;
        tk.base_lo = (uint64_t)base;
        tk.base_hi = (uint64_t)(base >> 64);
        tk.count = ok ? count : 0;
        tk.skip = (uint64_t)rem;
        tasks[t] = tk;
    }
}

// The tasks of a search over SEVERAL rank ranges (theta_search_ranges: the survivors of a branch and bound): spec[t] = {first rank
// lo, hi, candidates} as the host cut them; every task is unranked from the root.
__global__ __launch_bounds__(256) void n3_task_list_kernel(N3Dev P, const uint64_t *spec, int ntasks, N3Task *tasks, unsigned *stbuf) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= ntasks) return;
    const u128 base = ((u128)spec[3 * t + 1] << 64) | spec[3 * t];
    const int D = P.m - P.L;
    u128 rem = 0;
    const bool ok = n3_unrank_wave(P, base, D, lane, stbuf + (size_t)t * N3_STB, rem);
    if (lane == 0) {
        N3Task tk;
        tk.base_lo = (uint64_t)base;
        tk.base_hi = (uint64_t)(base >> 64);
        tk.count = ok ? spec[3 * t + 2] : 0;
        tk.skip = (uint64_t)rem;
        tasks[t] = tk;
    }
}

// one wave per tie record: its matrix
__global__ __launch_bounds__(256) void n3_unrank_list_kernel(N3Dev P, const TieRecord *recs, int count, unsigned char *out) {
    __shared__ unsigned path[4][N3_MAX_M_WIDE];       // the packed nodes of the wave's path
    const int wv = threadIdx.x >> 6, k = blockIdx.x * 4 + wv, lane = threadIdx.x & 63;
    if (k >= count) return;
    u128 rho = ((u128)recs[k].rank_hi << 64) | recs[k].rank_lo, rem;
    bool ok = n3_unrank_wave(P, rho, P.m, lane, path[wv], rem);
    wave_lds_sync();
    for (int i = lane; i < P.m; i += WAVE) {
        unsigned char *dst = out + (size_t)k * P.m * 2 + 2 * i;
        const unsigned st = path[wv][i];
        dst[0] = ok ? (unsigned char)((st >> 24) & 15u) : 255;
        dst[1] = ok ? (unsigned char)(st >> 28) : 255;
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
#ifndef N3_OCC
#define N3_OCC 3       // blocks per CU the register budget is sized for
#endif
#ifndef N3_TRIES
#define N3_TRIES 4    // evaluations a leaf may take in place before it is handed to the queue solver
#endif
#ifndef N3_LAG
#define N3_LAG 4      // lanes that may still be between leaves when the wave evaluates (they skip that round)
#endif
#ifndef N3_QCAP
#define N3_QCAP 128     // capacity of a wave's queue of leaves for the solver (survivors of the first evaluation)
#endif
#ifndef N3_MAX_L
#define N3_MAX_L 8      // leaf levels (one byte of the 64-bit leaf code each)
#endif

// Everything a wave owns sits in ONE struct, so that a single base register (+ immediate offsets) addresses all of it.
template <int L>
struct N3WaveLds {
    double gX[N3_MAX_Q + N3_MAX_L], gY[N3_MAX_Q + N3_MAX_L], gR[N3_MAX_Q + N3_MAX_L];   // group tile {a, b, sum r} of the prefix
    // f32 copy of the tile, two terms per entry for the packed coarse pass and the screen: {a0, a1, b0, b1} and {R0, R1};
    // an odd last term is paired with a copy of itself of weight 0
    float4 fXY[(N3_MAX_Q + 2) / 2];
    float2 fRR[(N3_MAX_Q + 2) / 2];
    float2 fRL[(N3_MAX_L + 2) / 2];   // weights of the leaf rows, paired like fRR (an odd last one with 0)
    float ws[2];                      // wave-wide warm start (mixture of the best candidate so far, blended)
    float resU1[N3_QCAP], resU2[N3_QCAP];   // coarse optimum (f32 is all the screen uses; contenders are polished in f64)
    unsigned long long qCode[N3_QCAP];
    unsigned short resSt[N3_QCAP], qOff[N3_QCAP];   // (a task holds < 65536 candidates)
    float lastN1[WAVE], lastN2[WAVE];     // mixture of the last admissible leaf each lane's chunk produced
    unsigned char qSrc[N3_QCAP];          // lane whose chunk the queue entry comes from
    unsigned char cIdx[N3_QCAP];          // queue entries that still need the values pass (the rest was dismissed)
    unsigned char preRow[N3_MAX_M];       // rows of the prefix, a | b << 4 (for the per-interval hybrj check of contenders)
    unsigned stkS[L > 1 ? L - 1 : 1][WAVE];            // lane-private DFS stack: node chosen at each leaf level but the last
    unsigned long long stkM[L > 1 ? L - 1 : 1][WAVE];  // ... and the siblings still to visit at that level
};
template <int L>
struct N3Lds {
    N3WaveLds<L> w[N3_WAVES];
    unsigned long long smask[N3_MAX_L][N3_MAX_Q];       // static child masks of the leaf depths
    unsigned char lb[N3_MAX_M], ub[N3_MAX_M];
    unsigned char ridx[N3_RIDX_W * N3_RIDX_W + 3];
    unsigned char rowtab[N3_MAX_Q + 3];
};

// status word of a solved leaf: bits 0-1 state (1 converged, 2 failed, 3 degenerate), bit 2 singular Hessian,
// bits 8.. iterations
#define RES_CONV 1u
#define RES_FAIL 2u
#define RES_DEGEN 3u
#define RES_SINGULAR 4u

// ---- the cold path of a candidate whose screened value is within the margin of the running minimum -------------
template <int L>
struct N3Leaf {
    const double *gX, *gY, *gR;   // group tile of the prefix (LDS)
    int G;
    double lx[L], ly[L], lr[L];   // the candidate's own leaf rows and their weights
    double s1, s2, inv_Rtot, Rmin, K0, thr, margin;
    const unsigned char *pre;     // rows of the prefix, a | b << 4 (LDS), D of them
    const double *r, *rN;         // per-interval counts (global)
    int D;
    double tau;
};
struct N3Cold {
    double u1, u2, nll, sc_gain;
    bool accept, contender;
};

// Polishes the coarse optimum to lambda^2 < 1e-12, re-decides admissibility, evaluates the exact FP64 NLL; for a
// converged optimum y OUTSIDE the simplex it first tries to dismiss the candidate with the self-concordance bound
//      NLL(z) >= NLL(y) + Rmin w(d / sqrt(Rmin)),   w(t) = t - ln(1 + t),   d = Hessian-norm distance from y to the simplex,
// which bounds everything the reference could report for it (one extra term pass instead of polish + logs).
// Would the reference's fsolve -- MINPACK hybrj from (1/3,1/3,1/3,1) on the Lagrangian system, restated operation by
// operation in hybrj4.hpp / n3_refsys.hpp -- end inside [0,1]^3 on this candidate?  If not, the reference reports it at its
// nu = 1/3 fallback although the likelihood has its minimum inside the simplex -- or not at all, when its BFGS call leaves
// the start (n3_refbfgs.hpp).  The candidate's rows
// are put back together (prefix rows from LDS, leaf rows from the lane) and the system is evaluated per interval in the
// reference's order, exactly as theta_solve_batch does: the landing point of hybrj on a rank-deficient system depends on
// the last bit, so an aggregated evaluation would decide some of them differently.
template <int L>
__device__ __noinline__ int n3_reference_outcome(const N3Leaf<L> &c, double nu[3]) {
    unsigned char rows[2 * N3_MAX_M];
    for (int i = 0; i < c.D; i++) {
        rows[2 * i] = c.pre[i] & 15u;
        rows[2 * i + 1] = c.pre[i] >> 4;
    }
#pragma unroll
    for (int l = 0; l < L; l++) {
        rows[2 * (c.D + l)] = (unsigned char)c.lx[l];
        rows[2 * (c.D + l) + 1] = (unsigned char)c.ly[l];
    }
    N3RefSystem sys;
    sys.m = c.D + L;
    sys.tau = c.tau;
    sys.r = c.r;
    sys.rN = c.rN;
    sys.c = rows;
    sys.init();
    return n3_ref_outcome(sys, nu);      // 1 own iterate (nu), 2 the nu = 1/3 fallback, 0 None
}

template <int L>
__device__ __noinline__ N3Cold n3_cold_path(N3Leaf<L> c, double u1, double u2, double nll, bool conv, bool accept, bool dump) {
    N3Cold out;
    out.sc_gain = 0.0;
    out.contender = true;
    const double s1 = c.s1, s2 = c.s2;
    auto terms = [&](auto &&body) {
        for (int g = 0; g < c.G; g++) body(c.gX[g], c.gY[g], c.gR[g]);
#pragma unroll
        for (int l = 0; l < L; l++) body(c.lx[l], c.ly[l], c.lr[l]);
    };
    if (!dump && conv && !accept) {
        double h11 = 0.0, h12 = 0.0, h22 = 0.0;
        terms([&](double x, double y, double R) {
            double a = x - s1, b = y - s2;
            double w = rcp_nr1(__builtin_fma(a, u1, __builtin_fma(b, u2, 1.0)));
            double tw = R * w * w;
            h11 = __builtin_fma(tw * a, a, h11);
            h12 = __builtin_fma(tw * a, b, h12);
            h22 = __builtin_fma(tw * b, b, h22);
        });
        // squared H-distance from y = (u1,u2) to the triangle (0,0), (1/s1,0), (0,1/s2): nearest point on an edge
        const double vx[3] = {0.0, 1.0 / s1, 0.0}, vy[3] = {0.0, 0.0, 1.0 / s2};
        double d2 = __builtin_inf();
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int f = (e + 1) % 3;
            double ex = vx[f] - vx[e], ey = vy[f] - vy[e];
            double px = u1 - vx[e], py = u2 - vy[e];
            double hex = h11 * ex + h12 * ey, hey = h12 * ex + h22 * ey;
            double t = (px * hex + py * hey) / (ex * hex + ey * hey);
            t = fmin(fmax(t, 0.0), 1.0);
            double rx = px - t * ex, ry = py - t * ey;
            d2 = fmin(d2, rx * (h11 * rx + h12 * ry) + ry * (h12 * rx + h22 * ry));
        }
        double tt = sqrt(fmax(d2, 0.0) / c.Rmin);
        double gain = 0.98 * c.Rmin * (tt - log1p(tt));       // (2 % off for the rcp arithmetic above)
        if (gain == gain) out.sc_gain = gain;
        if (!(nll - c.margin + out.sc_gain <= c.thr)) {       // dismissed: cannot come within the window of the minimum
            out.contender = false;
            out.u1 = u1; out.u2 = u2; out.nll = nll; out.accept = accept;
            return out;
        }
    }
    if (!dump && conv) {   // polish the coarse optimum, then decide admissibility again
        N3Newton T;
        T.u1 = T.p1 = u1;
        T.u2 = T.p2 = u2;
        T.iters = 0;
        T.status = 0;
        T.singular = false;
        while (T.status == 0 && T.iters < 12) n3_newton_step(terms, s1, s2, c.inv_Rtot, T, 1e-12);
        if (T.status == 1) {
            u1 = T.u1;
            u2 = T.u2;
            double n1 = s1 * u1, n2 = s2 * u2, n0 = 1.0 - n1 - n2;
            accept = (n0 >= 0.0 && n0 <= 1.0 && n1 >= 0.0 && n1 <= 1.0 && n2 >= 0.0 && n2 <= 1.0);
            if (!accept && T.singular) {
                N3Hess T2;
                T2.u1 = u1; T2.u2 = u2;
                T2.h11 = T2.h12 = T2.h22 = 0.0;
                terms([&](double x, double y, double R) {
                    double a = x - s1, b = y - s2;
                    double q = __builtin_fma(a, u1, __builtin_fma(b, u2, 1.0));
                    double tw = R / (q * q);
                    T2.h11 = __builtin_fma(tw * a, a, T2.h11);
                    T2.h12 = __builtin_fma(tw * a, b, T2.h12);
                    T2.h22 = __builtin_fma(tw * b, b, T2.h22);
                });
                accept = n3_admissible(T2, s1, s2);
                u1 = T2.u1;
                u2 = T2.u2;
            }
        }
    }
    // The minimum lies in the simplex -- but does the reference find it?  Its fsolve run may end on another root of the
    // rational system, outside [0,1]^3; it then reports the candidate at nu = (1/3,1/3,1/3), and so does this kernel: the
    // running minimum and the tie list follow what the reference reports, not what the likelihood could reach.
    // (Only a candidate whose exact minimum is within the window needs the answer -- its fallback value is no smaller.  The
    // screening margin that made it a contender is a thousand times wider than the window: in a stretch of near-ties,
    // e.g. the first ranks of the space, millions of contenders end here after one exact value pass.)
    double acc = 0.0;
    terms([&](double x, double y, double R) {
        double q = __builtin_fma(x - s1, u1, __builtin_fma(y - s2, u2, 1.0));
        acc = __builtin_fma(R, log(q), acc);
    });
    if (!dump && accept && c.K0 - acc <= c.thr) {
        double nu[3];
        const int outcome = n3_reference_outcome<L>(c, nu);
        if (outcome != 0) {
            // the point the reference reports: the fallback, or the iterate its fsolve stopped at (which may fall short
            // of the polished minimum by more than the tie margin)
            u1 = nu[1] / s1;
            u2 = nu[2] / s2;
            acc = 0.0;
            terms([&](double x, double y, double R) {
                double q = __builtin_fma(x - s1, u1, __builtin_fma(y - s2, u2, 1.0));
                acc = __builtin_fma(R, log(q), acc);
            });
        } else if (outcome == 0) {   // the reference returns None for it: neither a finalist nor a suspect
            accept = false;
            out.contender = false;
        }
    }
    out.u1 = u1; out.u2 = u2;
    out.nll = c.K0 - acc;
    out.accept = accept;
    return out;
}

template <int L, bool DUMP>
__global__ __launch_bounds__(64 * N3_WAVES, N3_OCC) void n3_search_kernel(N3Dev Pg, SearchArgs A, const N3Task *tasks,
                                                                const unsigned *stbuf, int ntasks, uint64_t per_task) {
    __shared__ N3Lds<L> S;
    const int m = Pg.m, D = m - L, Q = Pg.Q;
    // stage what every wave of the block shares
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        S.lb[i] = Pg.lb[i];
        S.ub[i] = Pg.ub[i];
    }
    for (int i = threadIdx.x; i < N3_RIDX_W * N3_RIDX_W; i += blockDim.x) S.ridx[i] = Pg.ridx[i];
    for (int i = threadIdx.x; i < Q; i += blockDim.x) S.rowtab[i] = Pg.rowtab[i];
    for (int i = threadIdx.x; i < L * N3_MAX_Q; i += blockDim.x) (&S.smask[0][0])[i] = Pg.smask[(size_t)D * N3_MAX_Q + i];
    __syncthreads();
    N3Dev P = Pg;
    P.lb = S.lb;
    P.ub = S.ub;
    P.ridx = S.ridx;
    P.rowtab = S.rowtab;

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * N3_WAVES + wv;
    if (task >= ntasks) return;  // whole wave leaves together; no block barrier below
    const double tau = (double)P.tau;
    N3WaveLds<L> &W = S.w[wv];
    double *gX = W.gX, *gY = W.gY, *gR = W.gR;
    float4 *fXY = W.fXY;
    float2 *fRR = W.fRR;
    float *resU1 = W.resU1, *resU2 = W.resU2;
    unsigned long long *qCode = W.qCode;
    unsigned short *resSt = W.resSt, *qOff = W.qOff;
    float *lastN1 = W.lastN1, *lastN2 = W.lastN2;
    unsigned char *qSrc = W.qSrc;
    unsigned char *cIdx = W.cIdx;
    const unsigned long long swm = Pg.swmask;
    const int NT1 = Pg.NT + 1;

    // lane i holds interval i; lane s (+64) also stands for alphabet slot s in the prefix successor
    unsigned st = lane < D ? stbuf[(size_t)task * N3_STB + lane] : 0u;
    const unsigned myrow0 = lane < P.Q ? P.rowtab[lane] : 0u;      // one alphabet slot per lane (Q <= 64: the second never exists)
    const int sa0 = (int)(myrow0 & 15u), sb0 = (int)(myrow0 >> 4), sa1 = 0, sb1 = 0;

    const N3Task tk = tasks[task];
    const u128 base = ((u128)tk.base_hi << 64) | tk.base_lo;
    unsigned long long remaining = tk.count, skip = tk.skip, processed = 0;
    const unsigned long long dump_base = (unsigned long long)task * per_task;  // position of the task in the dump arrays

    // leaf rows' shared data (wave-uniform)
    double leafR[L], leafN[L];
    float leafRf[L];
#pragma unroll
    for (int l = 0; l < L; l++) {
        leafR[l] = Pg.r[D + l];      // (uniform address: scalar loads)
        leafN[l] = Pg.rN[D + l];
        leafRf[l] = (float)leafR[l];
    }
    // screening margin for the single-precision NLL: |error| <= Rtot * (|ln q| * 2^-23 + 2^-22) stays far below this
    const double screen_margin = 2e-5 * P.Rtot + 1.0;
    const double inv_N = 1.0 / P.N, inv_Rtot = 1.0 / P.Rtot;
    // First pass: stop once the squared Newton decrement BEFORE the last step is below conv_l2 (1e-4): by
    // self-concordance the step then leaves lambda^2 <= ~1e-8, i.e. an NLL error <= ~1e-8 sum(r)/2 -- far inside the
    // screening margin.  Only contenders are polished to 1e-12 (below); the --GET_VALUES dump polishes everything.
    const double conv_main = DUMP ? 1e-12 : P.conv_l2;

    // statistics are wave-uniform scalars (popcounts of ballots): no vector registers
    unsigned long long n_eval = 0, n_acc = 0, n_deg = 0, n_it = 0, n_terms = 0, n_terms64 = 0, n_fin = 0, n_dis = 0;
    double best = order_unbits(load_agent_u64(&A.ctr->best_bits));
    double rej_best = order_unbits(load_agent_u64(&A.ctr->rej_bits));
    // warm start (wave-uniform): mixture fractions of the best candidate of the previous batch, pulled
    // towards the centre of the simplex so that it is interior for every candidate
    if (lane == 0) {
        W.ws[0] = W.ws[1] = 1.0f / 3.0f;
#pragma unroll
        for (int l = 0; l < L; l += 2) W.fRL[l >> 1] = make_float2(leafRf[l], l + 1 < L ? leafRf[l + 1] : 0.0f);
    }
    wave_lds_sync();

    // ---- lane-private DFS over the leaf levels -----------------------------------------------------------
    // feasible children of `node` when they sit at leaf level l: static rules (LDS) & symmetry & ratio window
    // (one 8-byte read of the precomputed table, L2 resident)
    auto child_mask = [&](const N3State &node, int l) -> unsigned long long {
        unsigned long long mk = S.smask[l][node.slot] & Pg.dynmask[((size_t)node.slot * NT1 + node.lo) * NT1 + (node.hi - 1)];
        return node.sw ? (mk & swm) : mk;
    };
    // state of child `s` of `node` (the child is known to be feasible)
    auto child_state = [&](const N3State &node, int s) -> N3State {
        N3State ch;
        n3_child_dyn(S.ridx, S.rowtab, node, s, ch);
        return ch;
    };

    unsigned long long pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pc4 = 0, pc6 = 0, n_prefix = 0;
    const unsigned long long t_begin = __builtin_readcyclecounter();
    while (remaining > 0) {
        unsigned long long t0 = __builtin_readcyclecounter();
        // the device-wide running minimum, once per prefix (thousands of candidates): the lower it is, the more is dismissed
        best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
        // ---------------- group tile of the prefix --------------------------------------------
        int G = 0;
        double S1p = 0.0, S2p = 0.0, Rmin = __builtin_inf();   // Rmin: smallest weight of a likelihood term
        N3Line pline = {0, 0, 0, 0, 0};                        // collinearity state of the prefix rows (n3_core.hpp)
        {
            const bool inp = lane < D;
            const unsigned myrow = st >> 24;  // a | b << 4
            if (inp) W.preRow[lane] = (unsigned char)myrow;
            // lane i stands for interval i here; its counts are re-read per prefix (L2) instead of living in registers
            const double r_i = inp ? Pg.r[lane] : 0.0, rN_i = inp ? Pg.rN[lane] : 0.0;
            unsigned long long todo = ballot64(inp);
            while (todo) {
                int leader = __builtin_ctzll(todo);
                unsigned q = (unsigned)__builtin_amdgcn_readlane((int)myrow, leader);
                bool match = inp && myrow == q;
                todo &= ~ballot64(match);
                double Rs = match ? r_i : 0.0, Ns = match ? rN_i : 0.0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {  // integer-valued doubles < 2^53: the sums are exact
                    Rs += __shfl_xor(Rs, o, WAVE);
                    Ns += __shfl_xor(Ns, o, WAVE);
                }
                double a = (double)(q & 15u), b = (double)(q >> 4);
                n3_line_add(pline, (int)(q & 15u), (int)(q >> 4));        // (scalar: q is wave-uniform)
                if (lane == 0) {
                    gX[G] = a;
                    gY[G] = b;
                    gR[G] = Rs;
                    float *xy = (float *)&fXY[G >> 1];
                    float *rr = (float *)&fRR[G >> 1];
                    if (G & 1) {
                        xy[1] = (float)a; xy[3] = (float)b; rr[1] = (float)Rs;
                    } else {   // also fills the second half: stays as the weight-0 pad when this is the last term
                        xy[0] = xy[1] = (float)a; xy[2] = xy[3] = (float)b; rr[0] = (float)Rs; rr[1] = 0.0f;
                    }
                }
                S1p += a * Ns;
                S2p += b * Ns;
                if (Rs > 0.0) Rmin = fmin(Rmin, Rs);
                G++;
            }
        }
#pragma unroll
        for (int l = 0; l < L; l++)
            if (leafR[l] > 0.0) Rmin = fmin(Rmin, leafR[l]);
        if (!(Rmin < __builtin_inf())) Rmin = 1.0;
        const float rtot_f = (float)P.Rtot, rtot_over_rmin = (float)(P.Rtot / Rmin);
        wave_lds_sync();
        pc0 += __builtin_readcyclecounter() - t0;

        const N3State par = n3_unpack((unsigned)__builtin_amdgcn_readlane((int)st, D - 1));
        // number of leaves below the prefix (exact, from the counting table); only its low part matters
        // here because a task never holds more than 2^47 candidates
        unsigned long long T;
        {
            u128 tv = Pg.cnt[n3_cnt_index(Pg, D - 1, par.slot, par.sw, par.lo, par.hi)];
            T = (tv >> 64) ? ~0ull : (unsigned long long)tv;
        }
        const unsigned long long lo_idx = skip < T ? skip : T;
        unsigned long long hi_idx = (T - lo_idx > remaining) ? lo_idx + remaining : T;

        // ---- lane-private enumeration: the prefix's leaves [lo_idx, hi_idx) are cut into 64 contiguous chunks;
        // every lane unranks the first leaf of its chunk ONCE (counting table) and then walks its chunk with
        // the DFS successor, a few leaves per round.
        const unsigned long long nleaf = hi_idx - lo_idx;
        const unsigned long long chunk = (nleaf + WAVE - 1) / WAVE;
        unsigned long long my_first = (unsigned long long)lane * chunk;
        unsigned long long my_left = my_first < nleaf ? ((nleaf - my_first < chunk) ? nleaf - my_first : chunk) : 0;
        unsigned my_rel = (unsigned)(processed + my_first);
        lastN1[lane] = __builtin_nanf("");        // no predecessor yet in this lane's chunk
        unsigned long long code = 0, mcur = 0;   // mcur: children of `cur` still to visit at level lv
        N3State cur = par;                      // parent of the level the lane is enumerating
        int lv = L - 1;
        {
            const unsigned long long tu0 = __builtin_readcyclecounter();
            if (my_left > 0) {
                unsigned long long idx = lo_idx + my_first;
                cur = par;
                bool okp = true;
                for (int l = 0; l < L && okp; l++) {
                    unsigned long long mk = child_mask(cur, l);
                    bool found = false;
                    if (l == L - 1) {                       // leaves count 1 each: take the idx-th set bit
                        while (mk && idx > 0) {
                            mk &= mk - 1;
                            idx--;
                        }
                        found = mk != 0ull;
                        mcur = mk;                          // the chosen leaf is still in the mask: emitted first
                    } else {
                        while (mk) {
                            int s = __builtin_ctzll(mk);
                            mk &= mk - 1;
                            N3State ch = child_state(cur, s);
                            u128 tv = Pg.cnt[n3_cnt_index(Pg, D + l, ch.slot, ch.sw, ch.lo, ch.hi)];
                            unsigned long long cv = (tv >> 64) ? ~0ull : (unsigned long long)tv;
                            if (idx < cv) {
                                found = true;
                                W.stkS[l][lane] = n3_pack(ch);
                                W.stkM[l][lane] = mk;   // siblings after the chosen child
                                code |= (unsigned long long)(ch.a | (ch.b << 4)) << (8 * l);
                                cur = ch;
                                break;
                            }
                            idx -= cv;
                        }
                    }
                    okp = found;
                }
                lv = L - 1;
                if (!okp) my_left = 0;   // cannot happen: the counting table says the leaf exists
            }
            pc6 += __builtin_readcyclecounter() - tu0;
            n_prefix++;
        }
        float wn1 = __builtin_nanf(""), wn2 = wn1;   // the lane's chain: optimum (mixture) of its previous leaf
        while (ballot64(my_left > 0)) {
            const unsigned long long ts0 = __builtin_readcyclecounter();
            int qcount = 0;
            // Enumeration fused with the FIRST evaluation.  Phase A: every lane advances its DFS (pop / descend, one
            // micro-step per trip) until it holds a leaf.  Phase B: all lanes take their leaf at once and evaluate it in
            // place -- packed value, gradient, Hessian at the optimum of the lane's previous leaf -- and nearly always
            // finish it right there by the lower bound of its optimum.  Only the survivors (and, for the dump / FP64
            // mode, everything) go to the LDS queue for the persistent-lane solver and the values pass below.
            {
                float flx[L], fly[L];                 // rows of the lane's current path (the last one is set per leaf)
                double S1u = S1p, S2u = S2p;          // column sums without the last row
#pragma unroll
                for (int l = 0; l + 1 < L; l++) {
                    const unsigned rw = (unsigned)(code >> (8 * l)) & 0xffu;
                    flx[l] = (float)(rw & 15u);
                    fly[l] = (float)(rw >> 4);
                    S1u = __builtin_fma((double)(rw & 15u), leafN[l], S1u);
                    S2u = __builtin_fma((double)(rw >> 4), leafN[l], S2u);
                }
                flx[L - 1] = fly[L - 1] = 0.0f;
                const int GPf = (G + 1) >> 1;
                auto fpairs = [&](auto &&body) {
#pragma unroll 2
                    for (int p = 0; p < GPf; p++) {
                        const float4 xy = fXY[p];
                        const float2 rr = fRR[p];
                        body(v2f{xy.x, xy.y}, v2f{xy.z, xy.w}, v2f{rr.x, rr.y});
                    }
#pragma unroll
                    for (int l = 0; l + 1 < L; l += 2) {
                        const float2 rr = W.fRL[l >> 1];
                        body(v2f{flx[l], flx[l + 1]}, v2f{fly[l], fly[l + 1]}, v2f{rr.x, rr.y});
                    }
                    if (L & 1) {
                        const float2 rr = W.fRL[L >> 1];
                        body(v2f{flx[L - 1], flx[L - 1]}, v2f{fly[L - 1], fly[L - 1]}, v2f{rr.x, rr.y});
                    }
                };
                const bool direct = DUMP || P.force64 != 0;    // no first evaluation here: everything is queued
                int pf_slot = -1;                              // child of the current level-(L-2) node whose mask is prefetched
                unsigned long long pf_mask = 0ull;
                while (qcount + WAVE <= N3_QCAP && ballot64(my_left > 0)) {
                    // ---- phase A
                    bool need = my_left > 0 && !(lv == L - 1 && mcur != 0ull);
                    // (not until EVERY lane holds a leaf: the few that have to pop several levels catch up in later trips;
                    // one trip is always made when any lane needs it, so every lane advances)
                    int lag = 0;
                    while (__builtin_popcountll(ballot64(need)) > lag) {
                        lag = N3_LAG;
                        if (need) {
                            // one trip: pop while the level is exhausted (LDS only), then descend once
#pragma unroll
                            for (int rep = 0; rep < 2; rep++)
                                if (mcur == 0ull && my_left > 0) {
                                    lv--;
                                    if (lv < 0) {
                                        my_left = 0;                   // cannot happen inside the counted range
                                    } else {
                                        mcur = W.stkM[lv][lane];
                                        cur = (lv == 0) ? par : n3_unpack(W.stkS[lv - 1][lane]);
                                    }
                                }
                            if (my_left > 0 && mcur != 0ull && lv < L - 1) {   // descend into child s
                                const int s = __builtin_ctzll(mcur);
                                mcur &= mcur - 1;
                                N3State ch = child_state(cur, s);
                                W.stkS[lv][lane] = n3_pack(ch);
                                W.stkM[lv][lane] = mcur;
                                code = (code & ~(0xffull << (8 * lv))) | ((unsigned long long)(ch.a | (ch.b << 4)) << (8 * lv));
#pragma unroll
                                for (int l = 0; l + 1 < L; l++)
                                    if (l == lv) {
                                        flx[l] = (float)ch.a;
                                        fly[l] = (float)ch.b;
                                    }
                                const unsigned long long sib = mcur;   // siblings still to come at this level
                                const N3State parent = cur;
                                cur = ch;
                                lv++;
                                unsigned long long mk = S.smask[lv][ch.slot];
                                if (ch.sw) mk &= swm;
                                // the ratio-window mask is a dependent read of the L2-resident table: at the deepest inner level
                                // the NEXT sibling's is requested one descent ahead, so that this wait is usually not needed
                                if (lv == L - 1 && pf_slot == s) mk &= pf_mask;
                                else mk &= Pg.dynmask[((size_t)ch.slot * NT1 + ch.lo) * NT1 + (ch.hi - 1)];
                                mcur = mk;
                                pf_slot = -1;
                                if (lv == L - 1) {
                                    if (sib) {
                                        const int s2 = __builtin_ctzll(sib);
                                        N3State c2 = child_state(parent, s2);
                                        pf_mask = Pg.dynmask[((size_t)c2.slot * NT1 + c2.lo) * NT1 + (c2.hi - 1)];
                                        pf_slot = s2;
                                    }
                                    // the path above the leaves is complete again
                                    S1u = S1p;
                                    S2u = S2p;
#pragma unroll
                                    for (int l = 0; l + 1 < L; l++) {
                                        S1u = __builtin_fma((double)flx[l], leafN[l], S1u);
                                        S2u = __builtin_fma((double)fly[l], leafN[l], S2u);
                                    }
                                }
                            }
                            need = my_left > 0 && !(lv == L - 1 && mcur != 0ull);
                        }
                    }
                    // ---- phase B
                    const bool act = my_left > 0 && lv == L - 1 && mcur != 0ull;
                    bool push = false, ev = false;
                    unsigned long long full = 0;
                    double fs1 = 1.0, fs2 = 1.0;
                    N3Newton T;
                    T.u1 = T.u2 = T.p1 = T.p2 = 0.0;
                    T.iters = 0;
                    T.status = 0;
                    T.singular = false;
                    if (act) {
                        const int s = __builtin_ctzll(mcur);
                        mcur &= mcur - 1;
                        const unsigned rw = S.rowtab[s];
                        full = (code & ~(0xffull << (8 * (L - 1)))) | ((unsigned long long)rw << (8 * (L - 1)));
                        push = true;
                        const double S1 = __builtin_fma((double)(rw & 15u), leafN[L - 1], S1u);
                        const double S2 = __builtin_fma((double)(rw >> 4), leafN[L - 1], S2u);
                        bool deficient = false;
                        if (pline.kind < 3) {         // (wave-uniform and rare: the prefix rows lie on one line -- does the whole candidate?)
                            N3Line ln = pline;
#pragma unroll
                            for (int l = 0; l < L; l++) n3_line_add(ln, (int)((full >> (8 * l)) & 15u), (int)((full >> (8 * l + 4)) & 15u));
                            deficient = ln.kind < 3;
                        }
                        if (deficient) full = ~0ull;  // rank-deficient: the queue lists it (RES_DEGEN) for the reference's own procedure
                        if (!direct && !deficient && S1 != 0.0 && S2 != 0.0) {
                            ev = true;
                            flx[L - 1] = (float)(rw & 15u);
                            fly[L - 1] = (float)(rw >> 4);
                            fs1 = S1 * inv_N;
                            fs2 = S2 * inv_N;
                            double n1 = (double)wn1, n2 = (double)wn2;
                            const bool pred = n1 == n1;
                            n1 = __builtin_fma(0.98, n1, 0.02 / 3.0);
                            n2 = __builtin_fma(0.98, n2, 0.02 / 3.0);
                            if (!pred) {
                                n1 = (double)W.ws[0];
                                n2 = (double)W.ws[1];
                            }
                            T.u1 = n1 * (double)__builtin_amdgcn_rcpf((float)fs1);
                            T.u2 = n2 * (double)__builtin_amdgcn_rcpf((float)fs2);
                            T.p1 = T.u1; T.p2 = T.u2;
                        }
                    }
                    // Evaluate in place.  Nearly always ONE trip: the bound finishes the leaf.  Where the optimum moves a lot
                    // from leaf to leaf, many lanes need more steps before the bound applies: they take up to three more
                    // here, in registers, as long as at least 8 of them do; whoever is left goes to the queue.
                    for (int tries = 0;; tries++) {
                        const unsigned long long evm = ballot64(ev);
                        if (!evm || tries >= N3_TRIES || (tries > 0 && __builtin_popcountll(evm) < 8)) break;
                        n_it += (unsigned)__builtin_popcountll(evm);
                        n_terms += (unsigned)__builtin_popcountll(evm) * (unsigned)(G + L);
                        if (ev) {
                            float val2 = 0.0f, l2v = -1.0f;
                            const bool okc = n3_newton_step_pk<true>(fpairs, (float)fs1, (float)fs2, inv_Rtot, T, conv_main, val2, l2v);
                            if (okc && l2v >= 0.0f && T.status != 2) {
                                const float lt2 = l2v * rtot_over_rmin;
                                if (lt2 < 0.25f) {
                                    const float lt = __builtin_sqrtf(lt2);
                                    const double gap = 1.05 * 0.5 * (double)(l2v * rtot_f * __builtin_amdgcn_rcpf(1.0f - lt));
                                    const double lb = (P.K0 - 0.6931471805599453 * (double)val2) - gap - screen_margin;
                                    if (lb > best + A.window && !P.no_dismiss) {
                                        push = false;          // dismissed
                                        ev = false;
                                    }
                                }
                                // the stepped iterate is the start of the lane's next leaf (and of the solver, for a survivor)
                                // (also when it lies outside the simplex: leaves whose optimum is far outside come in runs, and
                                // a start outside the next leaf's domain is repaired by the step function)
                                const double m1 = fs1 * T.u1, m2 = fs2 * T.u2;
                                if (fabs(m1) + fabs(m2) < 1e6) {
                                    wn1 = (float)m1;
                                    wn2 = (float)m2;
                                }
                            }
                            if (!okc || T.status != 0) ev = false;   // ill-conditioned, converged (a contender?) or failed: the queue
                        }
                    }
                    if (act && push && !direct) {
                        lastN1[lane] = wn1;
                        lastN2[lane] = wn2;
                    }
                    {
                        const unsigned ndm = (unsigned)__builtin_popcountll(ballot64(act && !push));
                        n_eval += ndm;
                        n_dis += ndm;
                    }
                    const unsigned long long em = ballot64(push);
                    if (push) {
                        const int pos = qcount + mbcnt(em);
                        qCode[pos] = full;
                        qOff[pos] = (unsigned short)my_rel;
                        qSrc[pos] = (unsigned char)lane;
                    }
                    if (act) {
                        my_rel++;
                        my_left--;
                    }
                    qcount += __builtin_popcountll(em);
                }
            }
            const unsigned long long td0 = __builtin_readcyclecounter();
            pc1 += td0 - ts0;
            if (qcount == 0) continue;     // everything was finished in place
            wave_lds_sync();

            // ---- solve every queued leaf: persistent lanes, refilled from the queue as they converge -------
            {
                int head = 0;
                bool have = false;
                int myidx = 0, mysrc = 0;
                unsigned long long mycode = 0;
                float lx[L], ly[L];   // the candidate's leaf rows (small integers: exact in f32)
                double s1 = 1.0, s2 = 1.0;
                N3Newton Sv;
                Sv.status = 0;
                const bool all64 = DUMP || P.force64 != 0;
                bool use64 = all64;     // FP64 iterations: the dump, and candidates whose Hessian f32 sums cannot resolve
                auto terms = [&](auto &&body) {
#pragma unroll 4
                    for (int g = 0; g < G; g++) body(gX[g], gY[g], gR[g]);
#pragma unroll
                    for (int l = 0; l < L; l++) body((double)lx[l], (double)ly[l], leafR[l]);
                };
                const int GP = (G + 1) >> 1;
                auto pairs = [&](auto &&body) {
#pragma unroll 2
                    for (int p = 0; p < GP; p++) {
                        const float4 xy = fXY[p];
                        const float2 rr = fRR[p];
                        body(v2f{xy.x, xy.y}, v2f{xy.z, xy.w}, v2f{rr.x, rr.y});
                    }
#pragma unroll
                    for (int l = 0; l + 1 < L; l += 2) {
                        const float2 rr = W.fRL[l >> 1];
                        body(v2f{lx[l], lx[l + 1]}, v2f{ly[l], ly[l + 1]}, v2f{rr.x, rr.y});
                    }
                    if (L & 1) {
                        const float2 rr = W.fRL[L >> 1];
                        body(v2f{lx[L - 1], lx[L - 1]}, v2f{ly[L - 1], ly[L - 1]}, v2f{rr.x, rr.y});
                    }
                };
                auto decode = [&](unsigned long long code_, double &S1, double &S2) {   // rows of the L leaf levels
                    S1 = S1p;
                    S2 = S2p;
#pragma unroll
                    for (int l = 0; l < L; l++) {
                        unsigned rw = (unsigned)(code_ >> (8 * l)) & 0xffu;
                        lx[l] = (float)(rw & 15u);
                        ly[l] = (float)(rw >> 4);
                        S1 = __builtin_fma((double)(rw & 15u), leafN[l], S1);
                        S2 = __builtin_fma((double)(rw >> 4), leafN[l], S2);
                    }
                };
                while (true) {
                    unsigned long long idle = ballot64(!have);
                    if (head < qcount && idle) {
                        int want = head + mbcnt(idle);
                        bool take = !have && want < qcount;
                        int ntake = __builtin_popcountll(idle);
                        if (ntake > qcount - head) ntake = qcount - head;
                        head += ntake;
                        if (take) {
                            myidx = want;
                            mycode = qCode[want];
                            double S1, S2;
                            decode(mycode, S1, S2);
                            if (S1 == 0.0 || S2 == 0.0 || mycode == ~0ull) {   // all-zero tumour column: Chat is NaN
                                resSt[want] = (unsigned short)RES_DEGEN;
                            } else {
                                have = true;
                                s1 = S1 * inv_N;
                                s2 = S2 * inv_N;
                                // start: the optimum of the previous leaf of the same chunk (it differs in the last
                                // rows only), else the best candidate of the previous batch; both pulled slightly
                                // towards the simplex centre so that they are interior for every candidate
                                const int src = qSrc[want];
                                mysrc = src;
                                double n1 = (double)lastN1[src], n2 = (double)lastN2[src];
                                const bool pred = n1 == n1;
                                n1 = __builtin_fma(0.98, n1, 0.02 / 3.0);
                                n2 = __builtin_fma(0.98, n2, 0.02 / 3.0);
                                if (!pred) {   // first leaf of a chunk
                                    n1 = (double)W.ws[0];
                                    n2 = (double)W.ws[1];
                                }
                                Sv.u1 = n1 * (double)__builtin_amdgcn_rcpf((float)s1);       // nu -> u (a starting point: f32 will do)
                                Sv.u2 = n2 * (double)__builtin_amdgcn_rcpf((float)s2);
                                Sv.p1 = Sv.u1; Sv.p2 = Sv.u2;
                                Sv.iters = 0;
                                Sv.status = 0;
                                Sv.singular = false;
                                use64 = all64;
                            }
                        }
                    }
                    if (!ballot64(have)) {
                        if (head >= qcount) break;
                        continue;   // only degenerate leaves were taken this round
                    }
                    {
                        const unsigned act = (unsigned)__builtin_popcountll(ballot64(have));
                        n_it += act;
                        n_terms += act * (unsigned)(G + L);
                        n_terms64 += (unsigned)__builtin_popcountll(ballot64(have && use64)) * (unsigned)(G + L);
                    }
                    if (have) {
                        float val2 = 0.0f, l2v = -1.0f;
                        if (use64) n3_newton_step(terms, s1, s2, inv_Rtot, Sv, conv_main);
                        else use64 = !n3_newton_step_pk<!DUMP>(pairs, (float)s1, (float)s2, inv_Rtot, Sv, conv_main, val2, l2v);
                        // "converged" rests on quadratic convergence, which self-concordance only grants once the decrement is
                        // small against the SMALLEST term weight: with an interval of a handful of reads (Rmin << sum r) the
                        // coarse threshold on lambda^2 / sum r is not enough -- keep iterating until lambda^2 < Rmin / 4 as well
                        if (!DUMP && Sv.status == 1 && l2v >= 0.0f && l2v * rtot_over_rmin >= 0.25f && Sv.iters < N3_MAX_ITERS) Sv.status = 0;
                        // Dismissal without solving: NLL is self-concordant with parameter 2 / sqrt(Rmin), so its minimum is
                        // at least NLL(u) - Rmin w*(lt), lt = lambda / sqrt(Rmin) < 1, w*(t) = -t - ln(1 - t) <= t^2 / (2 (1 - t)),
                        // i.e. NLL(u) - lambda^2 / (2 (1 - lt)), evaluated at the iterate BEFORE the step.  A candidate whose
                        // bound (less the f32 margin, plus 5 % for the f32 sums) is beyond the window of the running minimum
                        // can be neither a finalist nor a suspect: it is finished here -- no further iteration, no values pass.
                        if (!DUMP && l2v >= 0.0f && Sv.status != 2) {
                            const float lt2 = l2v * rtot_over_rmin;
                            if (lt2 < 0.25f) {
                                const float lt = __builtin_sqrtf(lt2);
                                const double gap = 1.05 * 0.5 * (double)(l2v * rtot_f * __builtin_amdgcn_rcpf(1.0f - lt));
                                const double lb = (P.K0 - 0.6931471805599453 * (double)val2) - gap - screen_margin;
                                if (lb > best + A.window && !P.no_dismiss) {
                                    resSt[myidx] = (unsigned short)0;          // state 0: dismissed
                                    // the stepped iterate still serves the next leaf of the chunk as a start, if interior
                                    const double n1 = s1 * Sv.u1, n2 = s2 * Sv.u2;
                                    if (fabs(n1) + fabs(n2) < 1e6) {
                                        lastN1[mysrc] = (float)n1;
                                        lastN2[mysrc] = (float)n2;
                                    }
                                    Sv.status = 0;
                                    have = false;
                                }
                            }
                        }
                        n_dis += (unsigned)__builtin_popcountll(ballot64(!have));
                        if (have && Sv.status != 0) {
                            unsigned sing = Sv.singular ? RES_SINGULAR : 0u;
                            bool conv = Sv.status == 1;
                            resU1[myidx] = (float)(conv ? Sv.u1 : Sv.p1);      // failed: last feasible iterate
                            resU2[myidx] = (float)(conv ? Sv.u2 : Sv.p2);
                            resSt[myidx] = (unsigned short)((conv ? RES_CONV : RES_FAIL) | sing | ((unsigned)Sv.iters << 8));
                            have = false;
                        }
                    }
                }
                wave_lds_sync();
                const unsigned long long td1 = __builtin_readcyclecounter();
                pc2 += td1 - td0;

                // ---- values, admissibility, minimum tracking: full 64-wide batches ------------------------
                // dismissed and degenerate leaves are only counted; the rest is compacted into dense batches
                int nkeep = 0;
                for (int b0 = 0; b0 < qcount; b0 += WAVE) {
                    const int idx = b0 + lane;
                    const bool in = idx < qcount;
                    const unsigned kind0 = in ? ((unsigned)resSt[idx] & 3u) : 0u;
                    const bool keep = in && kind0 != 0u && (DUMP || kind0 != RES_DEGEN);
                    n_eval += (unsigned)__builtin_popcountll(ballot64(in));
                    n_deg += (unsigned)__builtin_popcountll(ballot64(in && kind0 == RES_DEGEN));
                    // all-zero tumour column: the reference still reports something for it (Optimizer.py:128-165 on NaNs, then
                    // M3 / L3) -- the host gets its rank and lets theta_solve_batch reproduce that outcome
                    if (!DUMP && in && kind0 == RES_DEGEN && A.deg) degenerate_append(A.ctr, A.deg, A.deg_cap, base + qOff[idx]);
                    const unsigned long long km = ballot64(keep);
                    if (keep) cIdx[nkeep + mbcnt(km)] = (unsigned char)idx;
                    nkeep += __builtin_popcountll(km);
                }
                wave_lds_sync();
                for (int b0 = 0; b0 < nkeep; b0 += WAVE) {
                    const bool live = b0 + lane < nkeep;
                    const int idx = live ? (int)cIdx[b0 + lane] : 0;
                    mycode = live ? qCode[idx] : ~0ull;
                    unsigned stw = live ? resSt[idx] : RES_DEGEN;
                    double u1 = live ? (double)resU1[idx] : 0.0, u2 = live ? (double)resU2[idx] : 0.0;
                    const unsigned long long rel = live ? qOff[idx] : 0u;
                    double S1, S2;
                    decode(mycode, S1, S2);
                    const unsigned kind = stw & 3u;
                    const bool solved = live && kind != RES_DEGEN;
                    const bool conv = solved && kind == RES_CONV;
                    s1 = solved ? S1 * inv_N : 1.0;
                    s2 = solved ? S2 * inv_N : 1.0;
                    // admissibility (Optimizer.py:150-160): all nu_j in [0,1]
                    bool accept = false;
                    if (conv) {
                        double n1 = s1 * u1, n2 = s2 * u2, n0 = 1.0 - n1 - n2;
                        accept = (n0 >= 0.0 && n0 <= 1.0 && n1 >= 0.0 && n1 <= 1.0 && n2 >= 0.0 && n2 <= 1.0);
                        if (!accept && (stw & RES_SINGULAR)) {
                            // rank-deficient candidate: the minimiser is a line; rebuild H and intersect with the simplex
                            N3Hess T2;
                            T2.u1 = u1; T2.u2 = u2;
                            T2.h11 = T2.h12 = T2.h22 = 0.0;
                            terms([&](double x, double y, double R) {
                                double a = x - s1, b = y - s2;
                                double q = __builtin_fma(a, u1, __builtin_fma(b, u2, 1.0));
                                double tw = R / (q * q);
                                T2.h11 = __builtin_fma(tw * a, a, T2.h11);
                                T2.h12 = __builtin_fma(tw * a, b, T2.h12);
                                T2.h22 = __builtin_fma(tw * b, b, T2.h22);
                            });
                            accept = n3_admissible(T2, s1, s2);
                            u1 = T2.u1;
                            u2 = T2.u2;
                        }
                    }
                    // single-precision screen of sum R ln q (f32 tile, f32 log), the exact value only for contenders
                    float accf = 0.0f;
                    if (solved) {
                        const float fs1 = (float)s1, fs2 = (float)s2, fu1 = (float)u1, fu2 = (float)u2;
                        const v2f vs1 = {fs1, fs1}, vs2 = {fs2, fs2}, vu1 = {fu1, fu1}, vu2 = {fu2, fu2}, one = {1.f, 1.f};
                        v2f acc2 = {0.f, 0.f};
                        pairs([&](v2f x, v2f y, v2f R) {
                            v2f q = __builtin_elementwise_fma(x - vs1, vu1, __builtin_elementwise_fma(y - vs2, vu2, one));
                            v2f lg = {__builtin_amdgcn_logf(q.x), __builtin_amdgcn_logf(q.y)};   // log2
                            acc2 = __builtin_elementwise_fma(R, lg, acc2);
                        });
                        accf = (acc2.x + acc2.y) * 0.69314718056f;
                    }
                    double nll = P.K0 - (double)accf;
                    // Lower bound of what the reference could report for a rejected candidate.  Converged outside
                    // the simplex: the unconstrained minimum itself (NLL(u) - O(lambda^2 sum r), inside the margin).
                    // Not converged: Frank-Wolfe, NLL(z) >= NLL(u) + grad.(z - u) for every z in the simplex.
                    double fw = 0.0;
                    if (solved && kind == RES_FAIL) {
                        double g1 = 0.0, g2 = 0.0;
                        bool outside = false;    // (the stored iterate is rounded to f32: it may have left the domain)
                        terms([&](double x, double y, double R) {
                            double a = x - s1, b = y - s2;
                            double q = __builtin_fma(a, u1, __builtin_fma(b, u2, 1.0));
                            outside |= !(q > 0.0);
                            double t = R * rcp_nr1(q);
                            g1 = __builtin_fma(t, a, g1);
                            g2 = __builtin_fma(t, b, g2);
                        });
                        double e0 = g1 * u1 + g2 * u2;                  // vertex nu = e0  <-> u = (0, 0)
                        double e1 = e0 - g1 * rcp_nr2(s1), e2 = e0 - g2 * rcp_nr2(s2);    // vertices (1/s1, 0), (0, 1/s2)
                        fw = fmin(e0, fmin(e1, e2));
                        if (outside || !(fw == fw) || !(nll == nll)) {   // no usable bound: hand it to the host as a suspect
                            fw = 0.0;
                            nll = -__builtin_inf();
                        }
                    }
                    // contenders (accepted or rejected): approximate value within the screening margin of the minimum.
                    // Only they get polished and evaluated exactly.
                    bool contender = solved && (DUMP || (nll + fw - screen_margin <= best + A.window));
                    if (!DUMP && ballot64(contender) != 0) {
                        // other waves lower the device-wide minimum: re-read it before paying for a contender (a
                        // device-scope load goes past the XCD's L2, so not every round -- only when it can save work)
                        best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
                        contender = contender && (nll + fw - screen_margin <= best + A.window);
                    }
                    double sc_gain = 0.0;
                    if (contender) {   // rare: everything else about a contender happens out of line (keeps the hot loops lean)
                        N3Leaf<L> lf;
                        lf.gX = gX; lf.gY = gY; lf.gR = gR;
                        lf.G = G;
#pragma unroll
                        for (int l = 0; l < L; l++) {
                            lf.lx[l] = (double)lx[l];
                            lf.ly[l] = (double)ly[l];
                            lf.lr[l] = leafR[l];
                        }
                        lf.s1 = s1; lf.s2 = s2;
                        lf.inv_Rtot = inv_Rtot; lf.Rmin = Rmin; lf.K0 = P.K0;
                        lf.thr = best + A.window;
                        lf.margin = screen_margin;
                        lf.pre = W.preRow; lf.D = D; lf.r = Pg.r; lf.rN = Pg.rN; lf.tau = tau;
                        N3Cold cr = n3_cold_path<L>(lf, u1, u2, nll, conv, accept, DUMP);
                        u1 = cr.u1; u2 = cr.u2;
                        nll = cr.nll;
                        sc_gain = cr.sc_gain;
                        accept = cr.accept;
                        contender = cr.contender;
                    }
                    // lower bound of a rejected candidate: exact for contenders, else the screened value less its margin
                    double lbnd = (contender ? nll + fw : nll + fw - screen_margin) + sc_gain;
                    if (!(lbnd == lbnd)) lbnd = -__builtin_inf();   // unbounded: always a suspect
                    double mu0 = 0.0, mu1 = 0.0, mu2 = 0.0;
                    if (accept && contender) {   // closed form of M3 (Optimizer.py:318-330)
                        double u0 = (1.0 - s1 * u1 - s2 * u2) / tau;
                        double usum = u0 + u1 + u2;
                        mu0 = u0 / usum;
                        mu1 = u1 / usum;
                        mu2 = u2 / usum;
                    }

                    if (accept && contender && nll <= best + A.window) {
                        best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
                        if (nll <= best + A.window) {
                            tie_append(A.ctr, A.list, A.list_cap, base + rel, nll, mu0, mu1, mu2);
                            if (nll < best) atomicMin(&A.ctr->best_bits, order_bits(nll));
                        }
                    }
                    if (accept) {   // remember it for the next leaves of the same chunk (within a batch any of a chunk's entries may win)
                        const int src = qSrc[idx];
                        lastN1[src] = (float)(s1 * u1);
                        lastN2[src] = (float)(s2 * u2);
                    }
                    // wave-wide (only when some lane holds an exact contender): new minimum, warm start for chunk starts
                    const bool cand = accept && contender;
                    if (ballot64(cand)) {
                        double mine = cand ? nll : __builtin_inf();
                        double wbest = wave_min(mine);
                        best = fmin(best, wbest);
                        unsigned long long who = ballot64(mine == wbest);
                        int src = __builtin_ctzll(who);
                        double b1 = readlane_f64(s1 * u1, src), b2 = readlane_f64(s2 * u2, src);
                        if (lane == 0) {
                            W.ws[0] = (float)(P.warm_blend * b1 + (1.0 - P.warm_blend) / 3.0);
                            W.ws[1] = (float)(P.warm_blend * b2 + (1.0 - P.warm_blend) / 3.0);
                        }
                    }
                    // Suspects: rejected candidates whose lower bound is within the window of the minimum.  The reference
                    // could in principle report a stalled iterate for them; the host computes their exact simplex-
                    // boundary minimum (theta_boundary_min) to certify that none can reach the winner.
                    if (solved && !accept && contender && !DUMP && lbnd <= best + A.window)
                        suspect_append(A.ctr, A.sus, A.sus_cap, base + rel, lbnd, nll);
                    if (solved && !accept && lbnd < rej_best) {
                        unsigned long long old = atomicMin(&A.ctr->rej_bits, order_bits(lbnd));
                        if (old > order_bits(lbnd)) {  // we hold the minimum (racy pair, diagnostic only)
                            u128 rk = base + rel;
                            A.ctr->rej_rank_lo = (unsigned long long)rk;
                            A.ctr->rej_rank_hi = (unsigned long long)(rk >> 64);
                        }
                        rej_best = lbnd;
                    }
                    if (DUMP && live) {
                        double nan = __builtin_nan("");
                        const unsigned long long di = dump_base + rel;
                        A.dump_nll[di] = accept ? nll : nan;
                        A.dump_mu[di * 3 + 0] = accept ? mu0 : nan;
                        A.dump_mu[di * 3 + 1] = accept ? mu1 : nan;
                        A.dump_mu[di * 3 + 2] = accept ? mu2 : nan;
                    }
                    n_acc += (unsigned)__builtin_popcountll(ballot64(accept));
                    n_fin += (unsigned)__builtin_popcountll(ballot64(solved)) * (unsigned)(G + L);
                }
                wave_lds_sync();
                pc3 += __builtin_readcyclecounter() - td1;
            }
        }
        const unsigned long long consumed = hi_idx - lo_idx;
        skip -= lo_idx;
        processed += consumed;
        remaining -= consumed;
        if (remaining == 0) break;

        const unsigned long long tn0 = __builtin_readcyclecounter();
        // ---------------- next prefix in DFS order (wave-uniform) -----------------------------
        {
            int d = D - 1;
            bool fresh = false, alive = true;
            while (true) {
                int cur_slot = __builtin_amdgcn_readlane((int)st, d) & 0x7f;
                int start = fresh ? 0 : cur_slot + 1;
                N3State pst = n3_unpack((unsigned)__builtin_amdgcn_readlane((int)st, d > 0 ? d - 1 : 0));
                bool found = false;
                unsigned packed = 0;
                for (int cb = (start / WAVE) * WAVE; cb < Q && !found; cb += WAVE) {
                    const int s = cb + lane;
                    const int ca = cb ? sa1 : sa0, cbb = cb ? sb1 : sb0;
                    N3State nx;
                    bool ok = s >= start && s < Q &&
                              (d == 0 ? n3_first_row_ab(P, ca, cbb, s, nx) : n3_edge_ab(P, pst, ca, cbb, s, d, nx));
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
        pc4 += __builtin_readcyclecounter() - tn0;
    }

    if (lane == 0) {
        atomicAdd(&A.ctr->evaluated, n_eval);
        atomicAdd(&A.ctr->accepted, n_acc);
        atomicAdd(&A.ctr->degenerate, n_deg);
        atomicAdd(&A.ctr->iterations, n_it);
        atomicAdd(&A.ctr->terms, n_terms);
        atomicAdd(&A.ctr->terms64, n_terms64);
        atomicAdd(&A.ctr->dismissed, n_dis);
        atomicAdd(&A.ctr->final_terms, n_fin);
        atomicAdd(&A.ctr->prof[0], pc0);
        atomicAdd(&A.ctr->prof[1], pc1);
        atomicAdd(&A.ctr->prof[2], pc2);
        atomicAdd(&A.ctr->prof[3], pc3);
        atomicAdd(&A.ctr->prof[4], pc4);
        atomicAdd(&A.ctr->prof[5], __builtin_readcyclecounter() - t_begin);
        atomicAdd(&A.ctr->prof[6], pc6);
        atomicAdd(&A.ctr->prof[7], n_prefix);
    }
}


// ------------------------------------------------------------------------------------------------
// materialised generator (theta_enumerate): the search kernel's enumeration without the solver
// ------------------------------------------------------------------------------------------------
// One wave per rank range (the same tasks as the search).  The prefix rows are wave-uniform and staged once per prefix in
// LDS as 16-bit rows {a, b}; each lane walks its chunk of the leaves with the mask-driven DFS and writes its candidates'
// 2 m bytes itself: 32-bit stores (prefix words straight from LDS) when 2 m is a multiple of 4, 16-bit stores otherwise.
// HBM bound: 2 m bytes written per candidate, nothing read but the 173 MB counting table at chunk starts.
typedef unsigned n3_u4v __attribute__((ext_vector_type(4)));
typedef n3_u4v N3U4 __attribute__((aligned(4)));   // a 16-byte store that is only 4-byte aligned

template <int L>
struct N3EnumLds {
    struct {
        alignas(16) unsigned short pre[N3_MAX_M + 8];       // rows of the prefix, a | b << 8
        unsigned stkS[L > 1 ? L - 1 : 1][WAVE];
        unsigned long long stkM[L > 1 ? L - 1 : 1][WAVE];
    } w[N3_WAVES];
    unsigned long long smask[N3_MAX_L][N3_MAX_Q];
    unsigned char lb[N3_MAX_M], ub[N3_MAX_M];
    unsigned char ridx[N3_RIDX_W * N3_RIDX_W + 3];
    unsigned char rowtab[N3_MAX_Q + 3];
};

template <int L>
__global__ __launch_bounds__(64 * N3_WAVES) void n3_enumerate_wave_kernel(N3Dev Pg, const N3Task *tasks, const unsigned *stbuf,
                                                                          int ntasks, uint64_t per_task, unsigned char *out) {
    __shared__ N3EnumLds<L> S;
    const int m = Pg.m, D = m - L, Q = Pg.Q;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        S.lb[i] = Pg.lb[i];
        S.ub[i] = Pg.ub[i];
    }
    for (int i = threadIdx.x; i < N3_RIDX_W * N3_RIDX_W; i += blockDim.x) S.ridx[i] = Pg.ridx[i];
    for (int i = threadIdx.x; i < Q; i += blockDim.x) S.rowtab[i] = Pg.rowtab[i];
    for (int i = threadIdx.x; i < L * N3_MAX_Q; i += blockDim.x) (&S.smask[0][0])[i] = Pg.smask[(size_t)D * N3_MAX_Q + i];
    __syncthreads();
    N3Dev P = Pg;
    P.lb = S.lb;
    P.ub = S.ub;
    P.ridx = S.ridx;
    P.rowtab = S.rowtab;

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int task = blockIdx.x * N3_WAVES + wv;
    if (task >= ntasks) return;
    auto &W = S.w[wv];
    const unsigned long long swm = Pg.swmask;
    const int NT1 = Pg.NT + 1;
    unsigned st = lane < D ? stbuf[(size_t)task * N3_STB + lane] : 0u;
    const unsigned myrow0 = lane < P.Q ? P.rowtab[lane] : 0u;      // one alphabet slot per lane (Q <= 64: the second never exists)
    const int sa0 = (int)(myrow0 & 15u), sb0 = (int)(myrow0 >> 4), sa1 = 0, sb1 = 0;
    const N3Task tk = tasks[task];
    unsigned long long remaining = tk.count, skip = tk.skip, processed = 0;
    const size_t RS = (size_t)m * 2;                                      // bytes per candidate
    unsigned char *const obase = out + (size_t)task * per_task * RS;     // tasks are consecutive rank ranges of per_task
    const bool words = (RS & 3) == 0;

    auto child_mask = [&](const N3State &node, int l) -> unsigned long long {
        unsigned long long mk = S.smask[l][node.slot] & Pg.dynmask[((size_t)node.slot * NT1 + node.lo) * NT1 + (node.hi - 1)];
        return node.sw ? (mk & swm) : mk;
    };
    auto child_state = [&](const N3State &node, int s) -> N3State {
        N3State ch;
        n3_child_dyn(S.ridx, S.rowtab, node, s, ch);
        return ch;
    };
    auto leaf_row = [&](unsigned long long code, int l) -> unsigned {     // row l of the leaf levels as a | b << 8
        unsigned rw = (unsigned)(code >> (8 * l)) & 0xffu;
        return (rw & 15u) | ((rw >> 4) << 8);
    };

    while (remaining > 0) {
        if (lane < D) W.pre[lane] = (unsigned short)(((st >> 24) & 15u) | ((st >> 28) << 8));
        wave_lds_sync();
        const N3State par = n3_unpack((unsigned)__builtin_amdgcn_readlane((int)st, D - 1));
        unsigned long long T;
        {
            u128 tv = Pg.cnt[n3_cnt_index(Pg, D - 1, par.slot, par.sw, par.lo, par.hi)];
            T = (tv >> 64) ? ~0ull : (unsigned long long)tv;
        }
        const unsigned long long lo_idx = skip < T ? skip : T;
        const unsigned long long hi_idx = (T - lo_idx > remaining) ? lo_idx + remaining : T;
        const unsigned long long nleaf = hi_idx - lo_idx;
        const unsigned long long chunk = (nleaf + WAVE - 1) / WAVE;
        const unsigned long long my_first = (unsigned long long)lane * chunk;
        unsigned long long my_left = my_first < nleaf ? ((nleaf - my_first < chunk) ? nleaf - my_first : chunk) : 0;
        unsigned long long my_rel = processed + my_first;
        unsigned long long code = 0, mcur = 0;
        N3State cur = par;
        int lv = L - 1;
        if (my_left > 0) {   // unrank the first leaf of the lane's chunk
            unsigned long long idx = lo_idx + my_first;
            bool okp = true;
            for (int l = 0; l < L && okp; l++) {
                unsigned long long mk = child_mask(cur, l);
                bool found = false;
                if (l == L - 1) {
                    while (mk && idx > 0) {
                        mk &= mk - 1;
                        idx--;
                    }
                    found = mk != 0ull;
                    mcur = mk;
                } else {
                    while (mk) {
                        int s = __builtin_ctzll(mk);
                        mk &= mk - 1;
                        N3State ch = child_state(cur, s);
                        unsigned long long cv;
                        if (l == L - 2) {   // children of ch are leaves: count them from the masks (L2-resident table) instead
                            cv = (unsigned long long)__builtin_popcountll(child_mask(ch, L - 1));   // of the 173 MB one
                        } else {
                            u128 tv = Pg.cnt[n3_cnt_index(Pg, D + l, ch.slot, ch.sw, ch.lo, ch.hi)];
                            cv = (tv >> 64) ? ~0ull : (unsigned long long)tv;
                        }
                        if (idx < cv) {
                            found = true;
                            W.stkS[l][lane] = n3_pack(ch);
                            W.stkM[l][lane] = mk;
                            code |= (unsigned long long)(ch.a | (ch.b << 4)) << (8 * l);
                            cur = ch;
                            break;
                        }
                        idx -= cv;
                    }
                }
                okp = found;
            }
            if (!okp) my_left = 0;
        }
        bool adv = my_left > 0;
        while (ballot64(adv)) {
            bool emit = false;
            unsigned rw = 0;
            if (adv) {
                if (mcur == 0ull) {
                    lv--;
                    if (lv < 0) {
                        my_left = 0;
                        adv = false;
                    } else {
                        mcur = W.stkM[lv][lane];
                        cur = (lv == 0) ? par : n3_unpack(W.stkS[lv - 1][lane]);
                    }
                } else {
                    const int s = __builtin_ctzll(mcur);
                    mcur &= mcur - 1;
                    if (lv == L - 1) {
                        rw = S.rowtab[s];
                        emit = true;
                    } else {
                        N3State ch = child_state(cur, s);
                        W.stkS[lv][lane] = n3_pack(ch);
                        W.stkM[lv][lane] = mcur;
                        code = (code & ~(0xffull << (8 * lv))) | ((unsigned long long)(ch.a | (ch.b << 4)) << (8 * lv));
                        cur = ch;
                        lv++;
                        mcur = child_mask(ch, lv);
                    }
                }
            }
            if (emit) {
                const unsigned long long full = (code & ~(0xffull << (8 * (L - 1)))) | ((unsigned long long)rw << (8 * (L - 1)));
                unsigned char *dst = obase + (size_t)my_rel * RS;
                if (words) {
                    // 16-byte stores (the record is only 4-byte aligned: the hardware takes unaligned multi-dword stores)
                    unsigned *d32 = (unsigned *)dst;
                    const unsigned *p32 = (const unsigned *)W.pre;
                    const int NW = m >> 1, pw = D >> 1, odd = D & 1;     // words per record; words entirely in the prefix
                    const int pq = pw >> 2;                               // 16-byte chunks entirely in the prefix
                    for (int c = 0; c < pq; c++) {
                        const uint4 v = ((const uint4 *)p32)[c];
                        *(N3U4 *)(d32 + 4 * c) = N3U4{v.x, v.y, v.z, v.w};
                    }
                    unsigned t[8];                                        // the rest: <= 3 prefix words, the straddling word, leaf words
                    const int base = pq << 2, nt = NW - base;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int w = base + j;
                        unsigned v = 0;
                        if (w < pw) v = p32[w];
                        else if (w < NW) {
                            const int r0 = 2 * w - D;                     // first leaf row of the word (-1: the straddling word)
                            const unsigned lo16 = r0 < 0 ? (unsigned)W.pre[D - 1] : leaf_row(full, r0);
                            v = lo16 | (leaf_row(full, r0 + 1) << 16);
                        }
                        t[j] = v;
                    }
                    (void)odd;
                    if (nt >= 4) *(N3U4 *)(d32 + base) = N3U4{t[0], t[1], t[2], t[3]};
                    const int done4 = nt >= 4 ? 4 : 0;
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        if (j >= done4 && j < nt) d32[base + j] = t[j];
                } else {
                    unsigned short *d16 = (unsigned short *)dst;
                    for (int i = 0; i < D; i++) d16[i] = W.pre[i];
#pragma unroll
                    for (int l = 0; l < L; l++) d16[D + l] = (unsigned short)leaf_row(full, l);
                }
                my_rel++;
                my_left--;
                adv = my_left > 0;
            }
        }
        const unsigned long long consumed = hi_idx - lo_idx;
        skip -= lo_idx;
        processed += consumed;
        remaining -= consumed;
        if (remaining == 0) break;
        {   // next prefix in DFS order (wave-uniform) -- same walk as in the search kernel
            int d = D - 1;
            bool fresh = false, alive = true;
            while (true) {
                int cur_slot = __builtin_amdgcn_readlane((int)st, d) & 0x7f;
                int start = fresh ? 0 : cur_slot + 1;
                N3State pst = n3_unpack((unsigned)__builtin_amdgcn_readlane((int)st, d > 0 ? d - 1 : 0));
                bool found = false;
                unsigned packed = 0;
                for (int cb = (start / WAVE) * WAVE; cb < Q && !found; cb += WAVE) {
                    const int s = cb + lane;
                    const int ca = cb ? sa1 : sa0, cbb = cb ? sb1 : sb0;
                    N3State nx;
                    bool ok = s >= start && s < Q &&
                              (d == 0 ? n3_first_row_ab(P, ca, cbb, s, nx) : n3_edge_ab(P, pst, ca, cbb, s, d, nx));
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
            if (!alive) break;
        }
        wave_lds_sync();   // the prefix rows in LDS are rewritten next
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
    // (the anchor lives behind the N3_MAX_TASKS state rows of `stbuf`, which api.hip allocates with room for it)
    unsigned *anchor = getenv("THETA_N3_NO_ANCHOR") ? nullptr : stbuf + (size_t)N3_MAX_TASKS * N3_STB;
    if (anchor && ntasks > 1) hipLaunchKernelGGL(n3_anchor_kernel, dim3(1), dim3(64), 0, st, P, (uint64_t)begin, (uint64_t)(begin >> 64), anchor);
    else anchor = nullptr;
    hipLaunchKernelGGL(n3_task_kernel, dim3((ntasks + 3) / 4), dim3(256), 0, st, P, (uint64_t)begin, (uint64_t)(begin >> 64),
                       (uint64_t)end, (uint64_t)(end >> 64), per_task, ntasks, tasks, stbuf, (const unsigned *)anchor);
}

void n3_launch_task_list(const N3Dev &P, const uint64_t *spec, int ntasks, N3Task *tasks, unsigned *stbuf, hipStream_t st) {
    hipLaunchKernelGGL(n3_task_list_kernel, dim3((ntasks + 3) / 4), dim3(256), 0, st, P, spec, ntasks, tasks, stbuf);
}

void n3_launch_search(const N3Dev &P, const SearchArgs &A, const N3Task *tasks, const unsigned *stbuf, int ntasks,
                      uint64_t per_task, hipStream_t st) {
    dim3 grid((ntasks + N3_WAVES - 1) / N3_WAVES), block(64 * N3_WAVES);
    const bool dump = A.dump_nll != nullptr;
#define LAUNCH(LL)                                                                                                     \
    case LL:                                                                                                           \
        if (dump) hipLaunchKernelGGL((n3_search_kernel<LL, true>), grid, block, 0, st, P, A, tasks, stbuf, ntasks, per_task); \
        else hipLaunchKernelGGL((n3_search_kernel<LL, false>), grid, block, 0, st, P, A, tasks, stbuf, ntasks, per_task);     \
        break;
    #if N3_MAX_L >= 8
    switch (P.L) { LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) default: LAUNCH(8) }
#else
    switch (P.L) { LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) default: LAUNCH(6) }
#endif
#undef LAUNCH
}

void n3_launch_unrank_list(const N3Dev &P, const TieRecord *recs, int count, unsigned char *out, hipStream_t st) {
    hipLaunchKernelGGL(n3_unrank_list_kernel, dim3((count + 3) / 4), dim3(256), 0, st, P, recs, count, out);
}

void n3_launch_enumerate(const N3Dev &P, const N3Task *tasks, const unsigned *stbuf, int ntasks, uint64_t per_task,
                         unsigned char *out, hipStream_t st) {
    dim3 grid((ntasks + N3_WAVES - 1) / N3_WAVES), block(64 * N3_WAVES);
#define LAUNCH(LL)                                                                                                      \
    case LL:                                                                                                            \
        hipLaunchKernelGGL((n3_enumerate_wave_kernel<LL>), grid, block, 0, st, P, tasks, stbuf, ntasks, per_task, out); \
        break;
    #if N3_MAX_L >= 8
    switch (P.L) { LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) default: LAUNCH(8) }
#else
    switch (P.L) { LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) default: LAUNCH(6) }
#endif
#undef LAUNCH
}
