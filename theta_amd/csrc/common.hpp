// Shared helpers for libtheta_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/theta_hip.h"

typedef unsigned __int128 u128;

// ---------------------------------------------------------------------------------------------
// host side: error plumbing
// ---------------------------------------------------------------------------------------------
void theta_set_error(const char *fmt, ...);

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            theta_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                      \
            return THETA_ERR_HIP;                                                           \
        }                                                                                   \
    } while (0)

// entry of an API call: select the device, and forget what earlier calls (of anybody) left in HIP's per-thread error state --
// the hipGetLastError() checks after our launches must report OUR launches
#define HIP_ENTER(dev)                 \
    do {                               \
        HIP_TRY(hipSetDevice(dev));    \
        (void)hipGetLastError();       \
    } while (0)

struct theta_ctx {
    int device;
    hipStream_t stream;
    hipEvent_t ev0, ev1, ev2;
    int cu_count;
    uint64_t hbm_bytes;
    char name[128];
};

// Device buffer that frees itself (host RAII; one search instance owns a handful of these).
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int alloc(size_t n) {
        release();
        if (n == 0) n = 8;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            theta_set_error("hipMalloc(%zu bytes) failed: %s", n, hipGetErrorString(e));
            return THETA_ERR_HIP;
        }
        bytes = n;
        return THETA_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// Record the search kernels append to the device-side tie list.
struct TieRecord {
    uint64_t rank_lo, rank_hi;
    double nll;
    double mu[3];
};

// Counters every search kernel accumulates (one instance in HBM, zeroed per call).
struct SearchCounters {
    unsigned long long evaluated, accepted, degenerate, iterations, terms, final_terms;
    unsigned long long dismissed;      // n=3: candidates finished by the lower bound of their optimum (not solved to convergence)
    unsigned long long terms64;        // of `terms`: term evaluations done by FP64 iterations (n=3: the rest is packed f32)
    unsigned long long best_bits;      // order-preserving bits of the smallest accepted NLL
    unsigned long long rej_bits;       // same for the smallest rejected lower bound
    unsigned long long rej_rank_lo, rej_rank_hi;
    unsigned int list_count;           // records appended (may exceed capacity)
    unsigned int sus_count;            // suspects appended (may exceed capacity; never triggers a re-run)
    unsigned int deg_count;            // n=3: candidates with an all-zero tumour column appended to the degenerate list
    unsigned int line_count;           // n=3 sieve: tasks with a prefix whose rows lie on one line (SearchArgs::line), may exceed capacity
    unsigned long long sieve_survivors;   // n=3 fast path: contenders the sieve kernel handed to the finish kernel
    unsigned long long finish_iterations; // ... and the FP64 Newton iterations (m terms each) that kernel ran on them
    unsigned long long sieve_pterms;      // sieve: likelihood terms evaluated for last-level nodes (shared by a node's children)
    unsigned long long sieve_children;    // sieve: candidates given their shared first evaluation (one own term + the 2-D reduction)
    unsigned long long sieve_pruned;      // sieve: candidates of prefixes finished by the prefix bound (sv_prefix_beyond): no evaluation of their own
    unsigned long long prof[8];        // shader cycles per kernel phase, summed over waves (diagnostic)
};

// What a search kernel writes to (all device pointers).
struct SearchArgs {
    SearchCounters *ctr;
    SearchCounters *stat;              // THETA_STAT_SLOTS copies of the counter block, THETA_STAT_STRIDE bytes apart (each on cache lines of its
                                       // own), for the per-wave STATISTICS (evaluated, iterations, terms, ...): the host adds them up.  Tens
                                       // of thousands of waves adding to ONE line serialise in the memory-side atomic unit (~30 ns each --
                                       // a third of the n=2 search's time before round 4).  null: the statistics go to `ctr` as well
    TieRecord *list;
    unsigned list_cap;
    TieRecord *sus;                    // rejected candidates near the minimum (n=3 certificate)
    unsigned sus_cap;
    TieRecord *deg;                    // n=3 rank-deficient candidates (rank only; all-zero tumour columns among them): what the reference
    unsigned deg_cap;                  // reports for them is not their optimum; valued by theta_solve_batch, its procedure restated
    unsigned long long *line;          // n=3 sieve: first rank {lo, hi} of every task that met a prefix with collinear rows: the host
    unsigned line_cap;                 // materialises those tasks and lists their rank-deficient candidates (api.hip: list_deficient)
    double window;
    double *dump_nll;  // optional per-candidate dump (the reference's --GET_VALUES), else null
    double *dump_mu;
    // n=3 sieve, WITNESS build of the kernel only (n3_sieve.hip compiled with -DSV_WITNESS, theta_search_witness): what the sieve
    // LEFT every 2^wit_shift-th candidate of the call at.  Record i belongs to rank wit_begin + (i << wit_shift).  null otherwise.
    struct SvWitness *wit;
    unsigned long long wit_begin_lo, wit_begin_hi;
    unsigned long long wit_cap;
    unsigned wit_shift;
};

// One sampled candidate of a witnessed sieve run (theta_hip.h: theta_witness, same layout).
struct SvWitness {
    double mu[3];          // the mixture at the point the candidate was LEFT at (after its last Newton step), through M3's closed form
    double nll;            // K0 - ln2 (sum R log2 q) as the kernel computed it at its LAST EVALUATION (single-precision logarithms)
    float l2_last;         // lambda^2 / sum r found by that evaluation
    float l2_first;        // ... by the shared first evaluation (NaN: the candidate had no usable shared point)
    unsigned short evaluations;   // evaluations of the candidate: the shared one + its own
    unsigned short status;        // 0 none (degenerate candidate, prefix finished by its bound, fused-kernel fallback), 1 converged at the
                                  // shared evaluation, 2 converged in the queue, 3 / 4 finished by the lower bound (search mode) at the
                                  // shared evaluation / in the queue, 5 contender (listed for the finish kernel), 6 handed to the finish
                                  // kernel unsolved (ill-conditioned or 40 evaluations)
    float mu_bound;               // n3_mu_tol: what the certificate bounds the distance in mu between that point and the optimum by (0: none asked
                                  // for, or a point outside the simplex -- the reference reports no mu of its own there)
};

// ---------------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------------
#define WAVE 64

#define THETA_STAT_SLOTS 64
#define THETA_STAT_STRIDE 256
static_assert(sizeof(SearchCounters) <= THETA_STAT_STRIDE, "a statistics slot holds one SearchCounters");

// Monotone map double -> uint64 (so unsigned atomicMin orders doubles, negatives included).
__host__ __device__ inline unsigned long long order_bits(double x) {
    unsigned long long b;
    memcpy(&b, &x, 8);
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__host__ __device__ inline double order_unbits(unsigned long long b) {
    b = (b & 0x8000000000000000ull) ? (b & 0x7fffffffffffffffull) : ~b;
    double x;
    memcpy(&x, &b, 8);
    return x;
}

#ifdef __HIPCC__
// 1/x to ~2^-44 relative: v_rcp_f64 seed + one Newton-Raphson step.  Used only inside solver
// iterations, where the fixed point -- not each iterate -- decides the answer.
__device__ __forceinline__ double rcp_nr1(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
// Two steps: full double accuracy (final values).
__device__ __forceinline__ double rcp_nr2(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ unsigned long long ballot64(bool p) { return __ballot(p); }

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ long long wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// LDS hand-off between lanes of ONE wave (no other wave involved): make the compiler keep the
// order and wait for the LDS queue.  Waves of a block run independent tasks, so __syncthreads()
// is not usable here.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// the statistics slot of the calling wave (see SearchArgs::stat)
__device__ __forceinline__ SearchCounters *stat_slot(const SearchArgs &A) {
    if (!A.stat) return A.ctr;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    return (SearchCounters *)((char *)A.stat + (size_t)(wave & (THETA_STAT_SLOTS - 1)) * THETA_STAT_STRIDE);
}

__device__ __forceinline__ unsigned long long load_agent_u64(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Append one suspect (rejected candidate whose lower bound is within the window).
__device__ __forceinline__ void suspect_append(SearchCounters *ctr, TieRecord *list, unsigned cap, u128 rank, double lbound,
                                               double unconstrained) {
    unsigned idx = atomicAdd(&ctr->sus_count, 1u);
    if (idx < cap) {
        TieRecord rec;
        rec.rank_lo = (uint64_t)rank;
        rec.rank_hi = (uint64_t)(rank >> 64);
        rec.nll = lbound;
        rec.mu[0] = unconstrained;
        rec.mu[1] = 0.0;
        rec.mu[2] = 0.0;
        list[idx] = rec;
    }
}

// Append one rank-deficient candidate (rows on one line; an all-zero tumour column is the special case the kernels see
// directly): its rank is all the host needs.
__device__ __forceinline__ void degenerate_append(SearchCounters *ctr, TieRecord *list, unsigned cap, u128 rank) {
    unsigned idx = atomicAdd(&ctr->deg_count, 1u);
    if (idx < cap) {
        TieRecord rec;
        rec.rank_lo = (uint64_t)rank;
        rec.rank_hi = (uint64_t)(rank >> 64);
        rec.nll = __builtin_nan("");
        rec.mu[0] = rec.mu[1] = rec.mu[2] = 0.0;
        list[idx] = rec;
    }
}

// Append one record to the device tie list and lower the global best (device scope atomics).
__device__ __forceinline__ void tie_append(SearchCounters *ctr, TieRecord *list, unsigned cap, u128 rank,
                                           double nll, double mu0, double mu1, double mu2) {
    unsigned idx = atomicAdd(&ctr->list_count, 1u);
    if (idx < cap) {
        TieRecord rec;
        rec.rank_lo = (uint64_t)rank;
        rec.rank_hi = (uint64_t)(rank >> 64);
        rec.nll = nll;
        rec.mu[0] = mu0;
        rec.mu[1] = mu1;
        rec.mu[2] = mu2;
        list[idx] = rec;
    }
}
#endif
