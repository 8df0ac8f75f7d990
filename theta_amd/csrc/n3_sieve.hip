// n = 3 search, fast path for gfx950: a SIEVE kernel fed by the breadth-first burst enumerator, and a FINISH kernel for
// the few candidates the sieve cannot dispose of.
//
// Reference operators replaced (file:line into the reference's python/): the same as n3.hip --
//   Enumerator._generate_next_C_3(_recurse), _in_bounds, _get_mu_bounds          Enumerator.py:172-242
//   Optimizer._solve_n3plus + equations / jacobian / M3 / L3                      Optimizer.py:128-165, 236-330
//   the running minimum of do_optimization_single                                 RunTHetA.py:191-208
//
// Why two kernels.  In the fused kernel of n3.hip 99.998 % of the candidates of a large search are finished by a rigorous
// lower bound of their optimum after ONE packed-FP32 evaluation, yet every wave carries the machinery of the other
// 0.002 % (queue solver, values pass, polish, the hybrj restatement of the reference's outcome): 168 VGPRs plus spills, and
// 9 of its 12.5 vector instructions per candidate are lane-private DFS and queue bookkeeping (profiles/r1).  Here:
//
//   n3_sieve_kernel   persistent waves, one rank range (task) at a time.  Per prefix (the first m - ML rows): the group tile
//                     (intervals with the same row collapse into one likelihood term) and, in search mode, ONE bound that may
//                     finish the whole prefix (sv_prefix_beyond).  Else the last ML rows of the matrices are expanded
//                     breadth-first, level by level, all 64 lanes on 64 nodes of one level (the scheme of the materialised
//                     generator, n3_enum.hip: child mask = static rules & symmetry & ratio window, DPP scan of the child
//                     counts, children written to the next level's list in LDS).  A ROUND of <= 64 last-level nodes evaluates
//                     value / gradient / Hessian sums at ONE point w for all of them (sv_parent: the group tile's part once
//                     per round, the nodes' own path rows per lane), then one lane per CHILD adds its last row and has the
//                     Newton decrement and the self-concordance lower bound
//                         min NLL >= NLL(w) - (lambda^2 / 2)(1 + t + 2 t^2),   t = lambda / sqrt(Rmin) < 1/2
//                     (sv_child_eval): a candidate whose bound lies beyond the window of the running minimum is DONE (search),
//                     one within the coarse tolerance is converged (full solve).  The others take full evaluations at their own
//                     point from a small LDS queue with persistent lanes (sv_drain); what is still within the window once
//                     converged -- a contender -- is written, rows and rank, to a device list.  No DFS stack, no values
//                     pass, no cold path: the hot loops are the arithmetic and little else.
//   n3_finish_kernel  one LANE per listed contender: FP64 Newton per interval from the simplex centre, polish, admissibility
//                     (Optimizer.py:150-160), exact NLL, and for what is within the window the reference's own outcome
//                     (hybrj / BFGS restatement, n3_refbfgs.hpp) -- then the tie list, the suspect list and the device-wide
//                     minimum exactly as n3.hip's cold path maintains them.
//
// The fused kernel of n3.hip stays: it is the --GET_VALUES dump, the path for m < 8, the last resort for a slice part whose
// contender list overflows three times (a stretch of near-ties), and the second implementation the tests compare this one
// with (identical finalists).  F = float | double and "n3_no_dismiss" (every candidate iterated: the bench's full-solve
// legs) are instantiations / modes of THIS kernel.  The TIGHT full-solve modes (a threshold below 1e-6: every child takes a private FP64
// evaluation in place, sv_children) have a shared step of their own since round 6 -- single precision throughout, with a cubic
// correction from third-order sums (sv_third, sv_parent_third, sv_child_eval_third), rounds in whole trips of 64 children (sv_expand)
// -- and, for F = double, a kernel instantiation apart (SEC).
#include <stddef.h>

#include <type_traits>

#include "n3_core.hpp"
#define HYBRJ4_MANAGE_CONTRACT     // this unit allows fused multiply-adds; the hybrj restatement must not use them
#include "n3_refbfgs.hpp"
#include "n3_sieve.hpp"
#include "smx_log.hpp"             // the table-driven FP64 logarithm of the scorers (within one ulp; tests/test_smx_log_cpu.py)

#define SV_WAVES 4
#ifndef SV_CAP
#define SV_CAP 64         // nodes per intermediate level list
#endif
// ... of the list of last-level nodes: as long as the block's LDS allows at the kernel's occupancy (float: 3 blocks of <= 52 KB
// per CU -- one entry more and the third block no longer fits, measured: -15 %; double: 2 blocks of <= 80 KB)
template <class F> struct SvCapL { static constexpr int v = sizeof(F) == 4 ? 160 : 192; };
#ifndef SV_UNR
#define SV_UNR 2          // unroll factor of the group-pair loop of a full evaluation (sv_step)
#endif
#define SV_PRAGMA(x) _Pragma(#x)
#define SV_UNROLL(n) SV_PRAGMA(unroll n)
#ifndef SV_QCAP
#define SV_QCAP 128       // records waiting for further Newton steps
#endif
#ifndef SV_OCC
#define SV_OCC 3          // blocks per CU the register budget is sized for (LDS: ~52 KB per 4-wave block)
#endif
#ifndef SV_OCC64
#define SV_OCC64 2        // ... of the FP64 instantiation (two VGPRs per number, ~74 KB of LDS per block)
#endif
#ifndef SV_TRIES
#define SV_TRIES 2        // evaluations a record may take in place when many lanes need another
#endif

// profiling builds (tools/ab_build.sh): SV_PROF = cycles per phase in the seven diagnostic slots; SV_PROF + SV_PROF2 = COUNTS of
// wave-level passes in the same slots -- 0 last-level rounds, 1 nodes taken by them, 2 trips of the children phase, 3 wave-steps
// of the queue, 4 drains, 5 rounds of the upper levels, 6 children -- : lane utilisation = items / (64 passes)
#ifdef SV_PROF2
#define SV_CYC(x)
#define SV_CNT(x) x
#else
#define SV_CYC(x) x
#define SV_CNT(x)
#endif

#ifndef SV_PULL
#define SV_PULL 0.02      // the round's shared point = the chain point pulled this far towards the simplex centre
#endif
#ifndef SV_FULL_TRIPS
#define SV_FULL_TRIPS 1   // rounds of the last level in whole trips of 64 children (sv_expand)
#endif
#ifndef SV_KIDS
#define SV_KIDS 768       // children (candidates) of one round of <= 64 last-level nodes
#endif

// ---- the scalar type F of the likelihood arithmetic ---------------------------------------------------------------------
// float: the shipped search -- two likelihood terms per packed instruction (v_pk_fma_f32), v_rcp_f32, sums in single
// precision behind a rigorous margin.  double: the same kernel in FP64 throughout -- value, gradient, Hessian, the 2x2 solve
// and the step of every evaluation (v_fma_f64, v_rcp_f64 + one Newton-Raphson step); SURVEY 8(d)'s "passed through the full
// solve" when no candidate is dismissed (n3_no_dismiss).  In both, log2 q is v_log_f32's (the value is a SCREEN: what lies
// within the margin of the threshold goes to the finish kernel, which takes exact FP64 logarithms).
__device__ __forceinline__ float sv_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double sv_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float sv_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double sv_rcp(double x) { return rcp_nr1(x); }
// 1/q and log2 q of a likelihood term from ONE conversion to single precision.  F = double: the reciprocal is v_rcp_f32 of
// (float) q + one Newton-Raphson step in FP64 (relative error ~3e-14: v_rcp_f64 issues at a quarter of the rate and needs the
// same step, tools/micro/valu_rates.hip); only iterates and decrements are built on it -- values come from the logarithms,
// contenders are redone with exact arithmetic by the finish kernel.
__device__ __forceinline__ void sv_rcp_lg2(float q, float &w, float &l) {
    w = __builtin_amdgcn_rcpf(q);
    l = __builtin_amdgcn_logf(q);
}
__device__ __forceinline__ void sv_rcp_lg2(double q, double &w, double &l) {
    const float qf = (float)q;
    const double y = (double)__builtin_amdgcn_rcpf(qf);
    l = (double)__builtin_amdgcn_logf(qf);
    w = __builtin_fma(y, __builtin_fma(-q, y, 1.0), y);
}
__device__ __forceinline__ float sv_lg2(float x) { return __builtin_amdgcn_logf(x); }
__device__ __forceinline__ double sv_lg2(double x) { return (double)__builtin_amdgcn_logf((float)x); }
__device__ __forceinline__ float sv_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double sv_sqrt(double x) { return (double)__builtin_amdgcn_sqrtf((float)x); }   // (damping factor, bound gap)
__device__ __forceinline__ float sv_abs(float x) { return fabsf(x); }
__device__ __forceinline__ double sv_abs(double x) { return fabs(x); }
__device__ __forceinline__ float sv_min(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double sv_min(double a, double b) { return fmin(a, b); }
__device__ __forceinline__ float sv_bcast0(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, x))); }
__device__ __forceinline__ double sv_bcast0(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <class F> struct SvVec;
template <> struct SvVec<float> { typedef float v2 __attribute__((ext_vector_type(2))); };
template <> struct SvVec<double> { typedef double v2 __attribute__((ext_vector_type(2))); };
template <class F> struct alignas(16) Sv4 { F x, y, z, w; };
template <class F> struct alignas(2 * sizeof(F)) Sv2 { F x, y; };
// the weights of a pair of likelihood terms.  F = double: with their square roots rho -- per term alpha = rho a / q, beta = rho b / q,
// gradient sum rho alpha, Hessian sum alpha alpha^T: one multiplication less than through t = R / q, t / q.  (F = float keeps
// {R0, R1}: the extra 1.1 KB of LDS per block would cost the float kernel its third block per CU, measured 36.1 -> 45.2 ms.)
template <class F> struct SvWt;
template <> struct SvWt<float> {
    typedef Sv2<float> T;
    static __device__ __forceinline__ T make(float r0, float r1, double, double) { return T{r0, r1}; }
};
template <> struct SvWt<double> {
    typedef Sv4<double> T;
    static __device__ __forceinline__ T make(double r0, double r1, double h0, double h1) { return T{r0, r1, h0, h1}; }
};
template <class F> __device__ __forceinline__ typename SvVec<F>::v2 sv_rho(const Sv2<F> &) { return typename SvVec<F>::v2{F(0), F(0)}; }
template <class F> __device__ __forceinline__ typename SvVec<F>::v2 sv_rho(const Sv4<F> &t) { return typename SvVec<F>::v2{t.z, t.w}; }
// the record of a last-level node -- 15 numbers (shared sums L, T0..T2, W00..W22, column sums, the point) -- as planes of
// 16-byte vectors indexed [plane][node]: a wave reads one plane with consecutive 16-byte addresses (round 2 held the record
// node-major, 64 bytes apart: four lanes per bank group, SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS = 0.79)
template <class F> struct SvPlanes;
template <> struct SvPlanes<float> {
    float4 q[4][WAVE];
    __device__ __forceinline__ void put_third(int, float, float, const float (&)[20]) {}       // (never taken: sv_third is for F = double)
    __device__ __forceinline__ void get_third(int, float &, float &, float (&)[20]) const {}
    __device__ __forceinline__ void put(int n, const float (&v)[16]) {
#pragma unroll
        for (int k = 0; k < 4; k++) q[k][n] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    }
    __device__ __forceinline__ void get(int n, float (&v)[16]) const {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float4 t = q[k][n];
            v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w;
        }
    }
};
template <> struct SvPlanes<double> {
    double2 q[8][WAVE];
    __device__ __forceinline__ void put(int n, const double (&v)[16]) {
#pragma unroll
        for (int k = 0; k < 8; k++) q[k][n] = make_double2(v[2 * k], v[2 * k + 1]);
    }
    __device__ __forceinline__ void get(int n, double (&v)[16]) const {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const double2 t = q[k][n];
            v[2 * k] = t.x; v[2 * k + 1] = t.y;
        }
    }
    // (the tight modes' record: the column sums as doubles, then twenty floats -- six planes of the eight)
    __device__ __forceinline__ void put_third(int n, double s1, double s2, const float (&f)[20]) {
        q[0][n] = make_double2(s1, s2);
#pragma unroll
        for (int k = 0; k < 5; k++) ((float4 *)&q[1 + k][n])[0] = make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]);
    }
    __device__ __forceinline__ void get_third(int n, double &s1, double &s2, float (&f)[20]) const {
        const double2 t = q[0][n];
        s1 = t.x; s2 = t.y;
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float4 u = ((const float4 *)&q[1 + k][n])[0];
            f[4 * k] = u.x; f[4 * k + 1] = u.y; f[4 * k + 2] = u.z; f[4 * k + 3] = u.w;
        }
    }
};

// (F = double only; the float instantiations' LDS is full to the last 16 bytes -- their holder sits in the padding before pb_pt)
template <class F> struct SvShp { __device__ __forceinline__ F get(int) const { return F(0); } __device__ __forceinline__ void set(F, F, F) {} };
template <> struct SvShp<double> {
    float f[4];            // (single precision: the tight modes' shared step is taken in it throughout, sv_child_eval_third)
    __device__ __forceinline__ double get(int k) const { return (double)f[k]; }
    __device__ __forceinline__ void set(double a, double b, double c) { f[0] = (float)a; f[1] = (float)b; f[2] = (float)c; }
};
template <int ML, class F, int NS>
struct SvWave {
    alignas(16) unsigned short pre[WAVE * NS + 8];      // rows of the prefix, a | b << 8 (NS intervals per lane: 128, or 256 in the wide instantiation)
    uint2 list0[N3_MAX_Q];                              // level 1 nodes (children of the prefix's last node)
    uint2 list[ML > 3 ? ML - 3 : 1][SV_CAP];            // level l = 2 .. ML-2: {packed parent node, ancestor slots (6 bits each) | slot << 24}
    uint2 listL[SvCapL<F>::v];                               // level ML-1 (the last-level nodes): longer, so that a round of the level above
                                                        // takes ~40 nodes instead of ~12 (with 64 entries and ~5 children per node its
                                                        // rounds ran at a fifth of the lanes)
    unsigned short kid[SV_KIDS];                        // candidates of the current round: last row's slot | parent lane << 8
    SvPlanes<F> par;                                    // per last-level node of the round: shared sums, column sums, point
    unsigned pcode[WAVE];                               // ... and the slots of its path rows (6 bits each) | usable << 31
    unsigned task_line;                                 // a prefix of the task had collinear rows (n3_core.hpp: N3Line)
    SvShp<F> shp;                                       // the round's shared point (w0, u1, u2) where the nodes' records have no room for it (sv_third)
    double pb_pt[3];                                    // the prefix bound's last point (w0, u1, u2): where the next prefix's bound starts (sv_prefix_beyond)
    Sv4<F> fXY[(N3_MAX_Q + 2) / 2];                     // group tile of the prefix, two terms per entry {a0, a1, b0, b1}
    typename SvWt<F>::T fRR[(N3_MAX_Q + 2) / 2];        // ... and their weights {R0, R1} (an odd last term is paired with weight 0); F = double:
    typename SvWt<F>::T fRL[ML / 2];                    // {R0, R1, sqrt R0, sqrt R1}.  fRL: the weights of the leaf rows, paired likewise
    uint2 qRec[SV_QCAP];                                // queue of records that need more Newton steps: {slots of the path rows (6 bits
                                                        // each: the node's code), last row's slot | offset in the task << 8} -- the
                                                        // rows are decoded by the lane that takes the entry (sv_drain), not by the
                                                        // whole wave at every push
    F qU1[SV_QCAP], qU2[SV_QCAP];                       // ... the iterate it continues from
#ifdef SV_WITNESS
    float qL0[SV_QCAP];                                 // (witness build) lambda^2 / sum r of the record's shared first evaluation
#endif
};
// The ratio-rank table in LDS covers differences up to 7 copies: all there is up to K = 7, and nearly all beyond (the compact
// alphabet of a K > 7 search holds a few far-apart rows); the rest is read from the full table in HBM, which api.hip keeps behind
// the ratio masks (`dynmask`) so that the wave needs no pointer of its own for it.  (The full 31 x 31 table in LDS costs the float
// instantiation its third block per CU: 36 -> 45 ms per 2^31 candidates.)
#define SV_RIDX_W 15
// (TAB = the double instantiation of up to 128 intervals: the wide one's LDS is full -- two blocks per CU to the last kilobyte.)
template <class F, int NS> struct SvTab { static constexpr bool v = sizeof(F) == 8 && NS == 2; };
template <int ML, class F, int NS>
struct SvLds {
    SvWave<ML, F, NS> w[SV_WAVES];
    unsigned long long smask[ML][N3_MAX_Q];             // static child masks of the ML last depths
    unsigned char lb[WAVE * NS], ub[WAVE * NS];
    unsigned char ridx[SV_RIDX_W * SV_RIDX_W + 3];       // the central part of the ratio-rank table (|dx|, |dy| <= 7), see sv_child_dyn
    unsigned char rowtab[N3_MAX_Q + 3];
    unsigned short row16[N3_MAX_Q];                     // slot -> a | b << 8
    // F = double (round 6): slot -> (a, b) as numbers.  A full evaluation took the leaf rows of its record as bytes -- a table look-up,
    // two field extractions and two conversions per row, every evaluation (sv_child_rows + the decode in sv_step: 7 vector
    // instructions per row, 42 of an evaluation's ~480); one 16-byte LDS read by slot replaces them.  (The float instantiation keeps
    // the bytes: 512 B more would cost it its third block per CU.)
    Sv2<F> rowF[SvTab<F, NS>::v ? N3_MAX_Q : 1];
};

// inclusive scan over the wave with DPP row shifts / broadcasts (no LDS traffic)
__device__ __forceinline__ int sv_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// sum over the wave (every lane gets it), DPP row shifts / broadcasts
__device__ __forceinline__ float sv_wave_sum_f32(float v) {
#define SV_DPP_ADD(ctrl, rowmask) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowmask, 0xf, false))
    SV_DPP_ADD(0x111, 0xf);   // row_shr:1
    SV_DPP_ADD(0x112, 0xf);   // row_shr:2
    SV_DPP_ADD(0x114, 0xf);   // row_shr:4
    SV_DPP_ADD(0x118, 0xf);   // row_shr:8
    SV_DPP_ADD(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    SV_DPP_ADD(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef SV_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Everything a wave carries through the expansion (wave-uniform unless noted).
template <int ML, class F, int NS>
struct SvCtx {
    SvLds<ML, F, NS> *S;
    SvWave<ML, F, NS> *W;
    const unsigned long long *dynmask;
    const u128 *cnt;                 // counting table (only read while a task skips to its first candidate)
    int Q;
    unsigned long long swm;
    int NT1, lane, D, G, GP, m;
    N3State par;
    SearchArgs A;
    SvSurvivor *surv;
    unsigned surv_cap;
    unsigned *surv_count;
    u128 base;                       // rank of the task's first candidate
    unsigned remaining, done;        // candidates of the task still to come / done (a task holds < 2^16)
    unsigned long long skip;         // leaves of the task's first prefix that precede its first candidate
    // likelihood data of the current prefix
    F S1p, S2p;                      // column sums of the prefix rows (weighted by the normal counts), / N
    F leafN[ML];                     // normal counts of the leaf rows / N
    F leafRf[ML];                    // tumour counts of the leaf rows
    F leafRho[ML];                   // ... and their square roots
    F rtot_f, rtot_over_rmin, inv_Rtot, conv_l2, fine_l2;
    double A_mu_tol;                 // ... the tolerance itself (witness records)
    F mu_c, mu_t2, inv_tau;          // option n3_mu_tol: 0.891 (1 - t)^3 (1 - t+) tol sqrt(Rmin) / Rtot (per prefix; 0: off), the l2 of that t, 1 / tau -- see sv_mu_limit
    double K0, screen_margin, thr;   // thr = running minimum + window, loaded per task; screen_margin: see sv_beyond
    F Tcmp;                          // the threshold of sv_beyond's comparison, in F (per task)
    F sqrt_ror;                      // sqrt(Rtot / Rmin) (per prefix)
    int no_dismiss;
    int second;                      // 1: children that need another evaluation take it IN PLACE, right after the shared one (sv_children): the tight
                                     // full-solve modes, where every child does -- the queue then only holds what needs a third
    int third_min;                   // ... and a further one in place while at least this many lanes of the trip need it (65: never)
    // the chain point (wave-uniform): where the next round's shared sums are evaluated, see sv_parent
    F wn0, wn1, wn2;
    int qcount;
    // full trips (round 6): a round of the last level takes only as many nodes as fill whole trips of 64 children; the nodes of its
    // partial last trip wait for the next round -- at the end of a list at the front of listL, behind which the next fill is written
    // (carryL of them).  final_path: this expansion is the prefix's last at every level above -- nothing follows, so nothing is held back.
    int carryL, final_path;
    // statistics (wave-uniform scalars)
    // (32 bits each: a task holds < 2^16 candidates.  Degenerate candidates and contenders are counted where they are listed --
    // rare paths --, `dismissed` follows on the host: every regular candidate ends dismissed or listed.  Round 2 kept nine 64-bit
    // counters and six ballots per round of children: scalar registers the kernel does not have, they lived in VGPR lanes.)
#ifdef SV_PROF
    unsigned long long pt[7];        // cycles: 0 group tile, 1 parent phase, 2 children phase (less its drains), 3 drain, 4 prefix successor, 5 whole wave,
                                     // 6 the last level's own expansion (list read, child masks, scan, kid list) -- the upper levels are the rest
#endif
    unsigned n_par, n_prefix;        // likelihood terms of the shared sums (phase P), prefixes walked
    unsigned n_child, n_dit;         // shared first evaluations (children) / full evaluations (queue)
#ifdef SV_WITNESS
    unsigned long long wrel;         // rank of the task's first candidate - first rank of the call (a call walks < 2^56)
    double tau;
#endif
};

// ---- the witness build (-DSV_WITNESS: a second object of this very source, build.py; theta_search_witness) -----------------------
// What did the sieve LEAVE a candidate at?  Every 2^wit_shift-th candidate of the call writes a record when it is finished --
// converged, dismissed by its bound, or handed to the finish kernel: the mixture at the point it was left at (after the last
// Newton step), the value and the decrement its last evaluation found, the decrement of the shared first evaluation, and how
// many evaluations it took.  Every decision is made by the code the timed kernel runs (the records are written beside it; the
// tests compare the two builds' counters on the same ranges), so the records are what the bench's unit of work comes to.
#ifdef SV_WITNESS
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_witness(const SvCtx<ML, F, NS> &c, unsigned off, unsigned status, unsigned evals, float l2_first, F l2_last,
                                           F val2, F s1, F s2, F u1, F u2, F mu_lim = F(0)) {
    if (!c.A.wit) return;
    const unsigned long long rel = c.wrel + off;
    if (rel & ((1ull << c.A.wit_shift) - 1ull)) return;
    const unsigned long long idx = rel >> c.A.wit_shift;
    if (idx >= c.A.wit_cap) return;
    SvWitness *w = c.A.wit + idx;
    // nu = (1 - s1 u1 - s2 u2, s1 u1, s2 u2); nu -> mu is M3's closed form (Optimizer.py:318-330), as the finish kernel applies it
    const double d1 = (double)u1, d2 = (double)u2;
    const double u0 = (1.0 - (double)s1 * d1 - (double)s2 * d2) / c.tau, us = u0 + d1 + d2;
    w->mu[0] = u0 / us;
    w->mu[1] = d1 / us;
    w->mu[2] = d2 / us;
    w->nll = c.K0 - 0.6931471805599453 * (double)val2;
    w->l2_last = (float)l2_last;
    w->l2_first = l2_first;
    w->evaluations = (unsigned short)evals;
    w->status = (unsigned short)status;
    // the certificate's bound on the distance in mu (sv_mu_limit: converged means l2 <= limit, and the bound is tol x 0.9 l2 / limit)
    w->mu_bound = (c.mu_c > F(0) && mu_lim > F(0)) ? (float)(0.9 * (double)c.A_mu_tol * (double)l2_last / (double)mu_lim) : 0.0f;
}
#define SV_WIT(...) __VA_ARGS__
#else
#define SV_WIT(...)
#endif


// The leaf rows of a record as a full evaluation takes them.  F = float: two rows {a, b, a', b'} per dword, decoded term by term.
// F = double: the slots themselves -- the path code (6 bits per row) and the last row's slot --, looked up in SvLds::rowF.
template <int ML, bool TAB> struct SvRowsT;
template <int ML> struct SvRowsT<ML, false> {
    unsigned rw[ML / 2];
};
template <int ML> struct SvRowsT<ML, true> {
    unsigned code, slot;
};

// ---- the tolerance ON MU as a certificate (option "n3_mu_tol"; round 6) ---------------------------------------------------------
// An evaluation at u has the decrement lambda (l2 = lambda^2 / Rtot) and the tangent Hessian H.  With t = lambda / sqrt(Rmin) <= 0.1
// (f / Rmin is self-concordant) one full Newton step ends at lambda+^2 <= lambda^4 / (Rmin (1 - t)^4); the minimiser u* then lies within
// lambda+ / (1 - t+) of the new point in the norm of H(u+) >= (1 - t)^2 H(u), whose smaller eigenvalue is sigma = 2 det / (tr + sqrt(tr^2 - 4 det)) >= det / tr:
//      |u+ - u*|_2 <= l2 Rtot / ((1 - t)^3 (1 - t+) sqrt(Rmin sigma)).
// nu -> mu is M3's closed form (Optimizer.py:318-330): mu_j = u_j / U, U = u0 + u1 + u2, u0 = (1 - s1 u1 - s2 u2) / tau, so
// d mu_j = (du_j - mu_j k.du) / U with k = (1 - s1 / tau, 1 - s2 / tau), and |d mu|_inf <= |du|_2 (1.5 + max(1, |mu1| + |mu2|) |k|_2) / U.
// Hence the step from this evaluation ends within `tol` of the optimum in every component of mu if
//      l2 <= (0.9 / 1.01) (1 - t)^3 (1 - t+) tol sqrt(Rmin) / Rtot x sqrt(sigma) U / (1.5 + max(1, M) |k|)      (a 10 % margin; t taken at the
//      largest value an evaluation that passes n3_conv_l2 can have on the prefix: 0.718 at t = 0.1, 0.988 on the bench's data)
// -- the returned limit (c.mu_c holds the first factor).  Away from the simplex (a nu_j < -0.05, or U <= 0: the reference accepts
// no nu out of [0, 1], Optimizer.py:150-160, so it reports no mixture of the candidate's own there) there is no limit: +inf.
// tests/test_certified_tolerance_cpu.py checks the chain on random problems.
// (Single precision throughout: the limit carries a 10 % margin and the chain has no cancellation.  The limit is taken with det / tr for
// sigma.)
template <int ML, class F, int NS>
__device__ __forceinline__ F sv_mu_limit(const SvCtx<ML, F, NS> &c, F H11, F H22, F det, F s1, F s2, F u1, F u2) {
    const float fs1 = (float)s1, fs2 = (float)s2, fu1 = (float)u1, fu2 = (float)u2, it = (float)c.inv_tau;
    const float n1 = fs1 * fu1, n2 = fs2 * fu2, n0 = 1.0f - n1 - n2;
    const float k1 = __builtin_fmaf(-fs1, it, 1.0f), k2 = __builtin_fmaf(-fs2, it, 1.0f);
    const float U = __builtin_fmaf(n0, it, fu1 + fu2);
    // (The smaller eigenvalue through its lower bound det / tr -- sigma = det / the larger one, and that one is below the trace: within
    // a factor 2, and nearly exact where the Hessian is ill-conditioned, which is where the limit binds.  The whole limit then takes ONE
    // reciprocal and ONE square root: mu_c U^2 sqrt(det / (tr den^2)).  Round 6's first form took the eigenvalue itself -- two square
    // roots, three reciprocals: 4 % of the certified leg.  The Hessian's entries are sums of at most Rtot K^2, den is O(1): single
    // precision's range holds tr den^2 and det.)
    const float tr = (float)(H11 + H22);
    const float au = fabsf(fu1) + fabsf(fu2);
    const float kn = __builtin_amdgcn_sqrtf(__builtin_fmaf(k1, k1, k2 * k2));
    const float den = __builtin_fmaf(fmaxf(U, au), kn, 1.5f * U);                                   // U (1.5 + max(1, M) |k|)
    const float lim = (float)c.mu_c * (U * U) * __builtin_amdgcn_sqrtf((float)det * __builtin_amdgcn_rcpf(tr * (den * den)));
    // (the constant in mu_c assumes t <= the prefix's largest: an evaluation beyond it -- a caller's own, looser n3_conv_l2 -- is not the last)
    return (n0 >= -0.05f && n1 >= -0.05f && n2 >= -0.05f && U > 0.0f) ? (F)fminf(lim, (float)c.mu_t2) : F(__builtin_inff());
}

// One evaluation of value, gradient and Hessian at (u1, u2) over the group tile and the record's rows, two terms at a time
// (F = float: packed instructions), and the Newton step.  The likelihood in the scaled variables of n3_core.hpp:
// q_i = 1 + (x_i - s1) u1 + (y_i - s2) u2, NLL = K0 - sum R_i ln q_i.  Returns 0 = stepped (u1, u2 hold the new iterate;
// val2 = sum R log2 q and l2 = lambda^2 / Rtot at the OLD one), 1 = stepped and converged (l2 < conv), 2 = outside the domain
// (u1, u2 halved towards 0), 3 = no usable step (ill-conditioned Hessian, NaN).
template <int ML, class F, int NS>
__device__ __forceinline__ int sv_step(const SvCtx<ML, F, NS> &c, const SvRowsT<ML, SvTab<F, NS>::v> &rows, F s1, F s2, F &u1, F &u2, F &val2, F &l2, F &la, F &mlim) {
    typedef typename SvVec<F>::v2 v2;
    // (The sums START from the first pair of leaf rows -- products instead of multiply-adds onto zeros: fourteen FP64 accumulators set to
    // zero were 28 vector moves per evaluation, 5 % of it -- and take the other leaf rows, then the group tile.)
    v2 g1, g2, h11, h12, h22, lg, lga;
    const v2 vs1 = {s1, s1}, vs2 = {s2, s2}, vu1 = {u1, u1}, vu2 = {u2, u2}, one = {F(1), F(1)};
    // per term, with rho = sqrt R:  alpha = rho a / q, beta = rho b / q;  gradient sum rho alpha, Hessian sum alpha alpha^T
    // (one multiplication less than through t = R / q, t / q).  A term outside the domain (q <= 0, or so close that its
    // single-precision image is 0) leaves a NaN or an infinity in the sum of logarithms: no running minimum of q is kept.
    auto body = [&](auto first, v2 x, v2 y, v2 R, v2 rho) {
        constexpr bool FIRST = decltype(first)::value;
        v2 a = x - vs1, b = y - vs2;
        v2 q = __builtin_elementwise_fma(a, vu1, __builtin_elementwise_fma(b, vu2, one));
        F wx, wy, lx, ly;
#ifdef SV_NO_LOGS      // (A/B build, timing only: what the logarithms of a private evaluation cost; its values are meaningless)
        wx = sv_rcp(q.x); wy = sv_rcp(q.y); lx = ly = F(0);
#else
        sv_rcp_lg2(q.x, wx, lx);
        sv_rcp_lg2(q.y, wy, ly);
#endif
        const v2 w = {wx, wy}, l = {lx, ly};
        lg = FIRST ? R * l : __builtin_elementwise_fma(R, l, lg);
#ifndef SV_NO_LGA      // (A/B build: what the error bound's accumulation costs -- tools/ab_build.sh nolga -DSV_NO_LGA; never shipped)
        if constexpr (sizeof(F) == 8) {                // (error bound of the f32 logarithms, sv_beyond)
            const v2 la2 = {sv_abs(l.x), sv_abs(l.y)};
            lga = FIRST ? R * la2 : __builtin_elementwise_fma(R, la2, lga);
        }
#else
        if constexpr (FIRST) lga = v2{F(0), F(0)};
#endif
        if constexpr (sizeof(F) == 8) {
            v2 gw = rho * w;
            v2 al = gw * a, be = gw * b;
            g1 = FIRST ? rho * al : __builtin_elementwise_fma(rho, al, g1);
            g2 = FIRST ? rho * be : __builtin_elementwise_fma(rho, be, g2);
            h11 = FIRST ? al * al : __builtin_elementwise_fma(al, al, h11);
            h12 = FIRST ? al * be : __builtin_elementwise_fma(al, be, h12);
            h22 = FIRST ? be * be : __builtin_elementwise_fma(be, be, h22);
        } else {
            if constexpr (FIRST) lga = v2{F(0), F(0)};
            v2 t = R * w;
            g1 = FIRST ? t * a : __builtin_elementwise_fma(t, a, g1);
            g2 = FIRST ? t * b : __builtin_elementwise_fma(t, b, g2);
            v2 tw = t * w;
            v2 ta = tw * a, tb = tw * b;
            h11 = FIRST ? ta * a : __builtin_elementwise_fma(ta, a, h11);
            h12 = FIRST ? ta * b : __builtin_elementwise_fma(ta, b, h12);
            h22 = FIRST ? tb * b : __builtin_elementwise_fma(tb, b, h22);
        }
    };
    auto leaf = [&](auto first, int j) {
        const typename SvWt<F>::T rr = c.W->fRL[j];
        if constexpr (SvTab<F, NS>::v) {
            const Sv2<F> ra = c.S->rowF[(rows.code >> (12 * j)) & 63u];
            const Sv2<F> rb = c.S->rowF[j < ML / 2 - 1 ? (rows.code >> (12 * j + 6)) & 63u : rows.slot];
            body(first, v2{ra.x, rb.x}, v2{ra.y, rb.y}, v2{rr.x, rr.y}, sv_rho<F>(rr));
        } else {
            const unsigned d = rows.rw[j];      // bytes {a, b, a', b'}
            body(first, v2{(F)(d & 0xffu), (F)((d >> 16) & 0xffu)}, v2{(F)((d >> 8) & 0xffu), (F)(d >> 24)}, v2{rr.x, rr.y}, sv_rho<F>(rr));
        }
    };
    leaf(std::true_type{}, 0);
#pragma unroll
    for (int j = 1; j < ML / 2; j++) leaf(std::false_type{}, j);
    const Sv4<F> *fXY = c.W->fXY;
    const typename SvWt<F>::T *fRR = c.W->fRR;
SV_UNROLL(SV_UNR)
    for (int p = 0; p < c.GP; p++) {
        const Sv4<F> xy = fXY[p];
        const typename SvWt<F>::T rr = fRR[p];
        body(std::false_type{}, v2{xy.x, xy.y}, v2{xy.z, xy.w}, v2{rr.x, rr.y}, sv_rho<F>(rr));
    }
    val2 = lg.x + lg.y;
    if (!(sv_abs(val2) < F(__builtin_inff()))) {
        u1 *= F(0.5);
        u2 *= F(0.5);
        return 2;
    }
    const F G1 = g1.x + g1.y, G2 = g2.x + g2.y;
    const F H11 = h11.x + h11.y, H12 = h12.x + h12.y, H22 = h22.x + h22.y;
    const F hh = H11 * H22;
    const F det = sv_fma(-H12, H12, hh);
    if (!(det > (F)N3_COND_MIN * hh)) return 3;
    const F idet = sv_rcp(det);
    const F d1 = (H22 * G1 - H12 * G2) * idet;
    const F d2 = (H11 * G2 - H12 * G1) * idet;
    l2 = (G1 * d1 + G2 * d2) * c.inv_Rtot;
    la = lga.x + lga.y;
    if (!(l2 == l2) || !(sv_abs(d1) + sv_abs(d2) < F(1e30))) return 3;
    F step = F(1);
    if (ballot64(l2 > F(0.09))) {      // (damped phase: rare; a real branch -- if-converted, its square root and reciprocal ran for every evaluation)
        F l2l = l2;
        asm volatile("" : "+v"(l2l));
        if (l2 > F(0.09)) step = sv_rcp(F(1) + sv_sqrt(l2l));
    }
    // (option n3_mu_tol, wave-uniform: the limit the certificate on mu puts on l2 at THIS point, before the step)
    F conv = c.conv_l2;
    mlim = F(0);
    if (c.mu_c > F(0)) {
        mlim = sv_mu_limit<ML, F, NS>(c, H11, H22, det, s1, s2, u1, u2);
        conv = sv_min(conv, mlim);
    }
    u1 = sv_fma(step, d1, u1);
    u2 = sv_fma(step, d2, u2);
    // "converged" = the coarse threshold on lambda^2 / sum r AND lambda^2 < Rmin / 4 (quadratic convergence is only granted
    // once the decrement is small against the smallest term weight)
    return (l2 < conv && l2 * c.rtot_over_rmin < F(0.25)) ? 1 : 0;
}

// Is a rigorous LOWER BOUND of the candidate's optimum beyond the threshold (running minimum + window)?  From one evaluation
// (value sum val2 = sum R log2 q and decrement l2 = lambda^2 / Rtot at the same point): NLL is self-concordant with parameter
// 2 / sqrt(Rmin), hence with t = lambda / sqrt(Rmin) < 1/2
//      min NLL >= NLL(u) - lambda^2 / (2 (1 - t)) >= NLL(u) - (lambda^2 / 2) (1 + t + 2 t^2)         (1/(1-t) <= 1 + t + 2 t^2 on [0, 1/2])
// less what the computed value may be off by:
//   F = float   sums in single precision: 2e-5 Rtot + 1 (screen_margin; |error| <= Rtot (|ln q| 2^-23 + 2^-22) stays far below)
//   F = double  the sums are exact to ~1e-15; only the logarithms are single precision (v_log_f32 of (float) q: |error| <=
//               2^-24 / ln 2 from the conversion + one ulp of the result), so the value is off by at most
//               8.7e-8 Rtot + 1.3e-7 sum R |log2 q| (`la`, accumulated next to the value) -- a few units instead of 133 on the
//               bench's data: the contender list of the FP64 mode holds genuine near-ties only.
// The test  K0 - ln2 val2 - gap - margin > thr  is made as  ln2 val2 + gap (+ margin) < T  in F, with T = K0 - thr (- margin)
// rounded DOWN into F by more than the comparison's own rounding (sv_set_threshold) -- no conversions in the hot path.
// false when the bound does not apply (t >= 1/2).  sl = sqrt(l2), which the caller has anyway.
template <int ML, class F, int NS>
__device__ __forceinline__ bool sv_beyond(const SvCtx<ML, F, NS> &c, F val2, F l2, F sl, F la) {
    const F t = sl * c.sqrt_ror;
    const F gap = F(1.05 * 0.5) * (l2 * c.rtot_f) * sv_fma(t, sv_fma(F(2), t, F(1)), F(1));
    F lhs = sv_fma(F(0.6931471805599453), val2, gap);
    if constexpr (sizeof(F) == 8) lhs += 0.6931471805599453 * (8.7e-8 * c.rtot_f + 1.3e-7 * la) + 1e-3;
    return t < F(0.5) && lhs < c.Tcmp;
}
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_set_threshold(SvCtx<ML, F, NS> &c, double thr) {
    c.thr = thr;
    if constexpr (sizeof(F) == 8) {
        c.Tcmp = c.K0 - thr;                            // (the margin depends on the evaluation: added to the left side)
    } else {
        // float: lhs carries <= 3 roundings of relative 2^-24 on magnitudes <= |T| + ~1000 wherever the comparison could flip
        const double T = c.K0 - thr - c.screen_margin;
        c.Tcmp = (float)(T - 4e-7 * (fabs(T) + 1000.0) - 0.01);
    }
}

// column sums / N of a record: (s1, s2); false if a tumour column is all zero (degenerate: the reference's Chat is NaN)
template <int ML, class F, int NS>
__device__ __forceinline__ bool sv_sums(const SvCtx<ML, F, NS> &c, const SvRowsT<ML, SvTab<F, NS>::v> &rows, F &s1, F &s2) {
    F a = c.S1p, b = c.S2p;
#pragma unroll
    for (int j = 0; j < ML / 2; j++) {
        if constexpr (SvTab<F, NS>::v) {
            const Sv2<F> ra = c.S->rowF[(rows.code >> (12 * j)) & 63u];
            const Sv2<F> rb = c.S->rowF[j < ML / 2 - 1 ? (rows.code >> (12 * j + 6)) & 63u : rows.slot];
            a = sv_fma(ra.x, c.leafN[2 * j], a);
            b = sv_fma(ra.y, c.leafN[2 * j], b);
            a = sv_fma(rb.x, c.leafN[2 * j + 1], a);
            b = sv_fma(rb.y, c.leafN[2 * j + 1], b);
        } else {
            const unsigned d = rows.rw[j];
            a = sv_fma((F)(d & 0xffu), c.leafN[2 * j], a);
            b = sv_fma((F)((d >> 8) & 0xffu), c.leafN[2 * j], b);
            a = sv_fma((F)((d >> 16) & 0xffu), c.leafN[2 * j + 1], a);
            b = sv_fma((F)(d >> 24), c.leafN[2 * j + 1], b);
        }
    }
    s1 = a;
    s2 = b;
    return a > F(0) && b > F(0);
}

// A contender (or a record the sieve cannot handle): rows and rank to the device list; the finish kernel takes it from there.
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_survivor(const SvCtx<ML, F, NS> &c, const unsigned (&rw)[ML / 2], unsigned off, F u1, F u2) {
    const unsigned idx = atomicAdd(c.surv_count, 1u);
    atomicAdd(&c.A.ctr->sieve_survivors, 1ull);
    if (idx >= c.surv_cap) return;            // the host sees the count and redoes the slice
    SvSurvivor *s = c.surv + idx;
    const u128 rk = c.base + off;
    s->rank_lo = (uint64_t)rk;
    s->rank_hi = (uint64_t)(rk >> 64);
    s->u1 = (double)u1;                       // (round 5: the finish kernel starts its FP64 Newton from the sieve's iterate, not from the simplex centre)
    s->u2 = (double)u2;
    unsigned short *dst = (unsigned short *)s->rows;
    for (int i = 0; i < c.D; i++) dst[i] = c.W->pre[i];
#pragma unroll
    for (int j = 0; j < ML / 2; j++) {
        dst[c.D + 2 * j] = (unsigned short)(rw[j] & 0xffffu);
        dst[c.D + 2 * j + 1] = (unsigned short)(rw[j] >> 16);
    }
}

// leaf rows of a child as the queue / the contender list hold them: two rows {a, b, a', b'} per dword
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_child_rows(const SvCtx<ML, F, NS> &c, unsigned code, unsigned slot, unsigned (&rw)[ML / 2]) {
#pragma unroll
    for (int j = 0; j < ML / 2; j++) {
        rw[j] = c.S->row16[(code >> (12 * j)) & 63u];
        if (j < ML / 2 - 1) rw[j] |= (unsigned)c.S->row16[(code >> (12 * j + 6)) & 63u] << 16;
    }
    rw[ML / 2 - 1] |= (unsigned)c.S->row16[slot] << 16;
}

template <int ML, class F, int NS>
__device__ __forceinline__ void sv_rows_load(const SvCtx<ML, F, NS> &c, unsigned code, unsigned slot, SvRowsT<ML, SvTab<F, NS>::v> &rows) {
    if constexpr (SvTab<F, NS>::v) {
        rows.code = code;
        rows.slot = slot;
    } else {
        sv_child_rows<ML, F, NS>(c, code, slot, rows.rw);
    }
}
// ... and a contender's rows for the list (rare path)
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_survivor_rows(const SvCtx<ML, F, NS> &c, const SvRowsT<ML, SvTab<F, NS>::v> &rows, unsigned off, F u1, F u2) {
    if constexpr (SvTab<F, NS>::v) {
        // (a contender is rare: its table addresses and its 128-bit rank are computed HERE -- laundered, or they are hoisted out of
        // the branch and of the evaluation loop, ten vector instructions per trip for nothing)
        unsigned code = rows.code;
        asm volatile("" : "+v"(code), "+v"(off));
        unsigned rw[ML / 2];
        sv_child_rows<ML, F, NS>(c, code, rows.slot, rw);
        sv_survivor<ML, F, NS>(c, rw, off, u1, u2);
    } else {
        sv_survivor<ML, F, NS>(c, rows.rw, off, u1, u2);
    }
}

// Further Newton steps for the queued records: until dismissed, converged (a contender) or given up.  PERSISTENT LANES: a lane
// whose record is finished takes the next queue entry at once (ballot + prefix count of the idle lanes), so the wave iterates
// as long as there is work for most of its lanes -- round 2 took the queue 64 at a time in lock step, and with a third of the
// records needing a second or third step every batch ran three iterations at a fraction of its lanes.
//
// FULL WAVE-STEPS ONLY (round 4).  A wave-step costs the same with 5 live lanes as with 64, and a queue of ~110 entries of which
// one in ten needs a second or third evaluation ended every drain with steps at 50, 5, 1 live lanes: 60 % of the lane slots of
// the queue phase did the work.  A drain in the middle of a prefix (FINAL = false: the queue is about to overflow) now stops
// as soon as the queue is handed out and fewer than SV_QTHR lanes are still live: those lanes put their record back -- the
// entry as it was, with the iterate they have reached -- and the next drain starts on a full wave again.  The drain at the
// end of a prefix (FINAL: the group tile changes) empties the queue as before.  SV_QTHR <= 64 leaves room for the 64 pushes
// of the trip that called.
#ifndef SV_QTHR
#define SV_QTHR 56
#endif
template <int ML, class F, int NS, bool FINAL>
__device__ __forceinline__ void sv_drain(SvCtx<ML, F, NS> &c) {
#ifdef SV_PROF
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    int next = 0;                                   // (wave-uniform) queue entries handed out so far
    int left = 0;                                   // (wave-uniform) entries put back
    bool live = false;
    SvRowsT<ML, SvTab<F, NS>::v> rows;
    sv_rows_load<ML, F, NS>(c, 0u, 0u, rows);
    F u1 = F(0), u2 = F(0), s1 = F(1), s2 = F(1);
    unsigned qy = 0u, code = 0u;                    // the entry's words: last row's slot | offset in the task << 8 | evaluations so far << 24, path slots
    int iters = 0;                                  // evaluations of the record so far, the shared one included (kept across put-backs)
    SV_WIT(float l0 = 0.0f;)
    while (true) {
        const unsigned long long idle = ballot64(!live);
        if (next < c.qcount && idle) {
            const int idx = next + mbcnt(idle);
            if (!live && idx < c.qcount) {
                const uint2 qr = c.W->qRec[idx];
                sv_rows_load<ML, F, NS>(c, qr.x, qr.y & 0xffu, rows);
                u1 = c.W->qU1[idx];
                u2 = c.W->qU2[idx];
                code = qr.x;
                qy = qr.y & 0xffffffu;
                SV_WIT(l0 = c.W->qL0[idx];)
                sv_sums<ML, F, NS>(c, rows, s1, s2);
                if (!(u1 == u1)) {                    // (no usable first point: from the simplex centre)
                    u1 = F(1.0 / 3.0) * sv_rcp(s1);
                    u2 = F(1.0 / 3.0) * sv_rcp(s2);
                }
                iters = (int)(qr.y >> 24);
                live = true;
            }
            const int room = c.qcount - next, nid = __builtin_popcountll(idle);
            next += nid < room ? nid : room;
        }
        const unsigned long long lm = ballot64(live);
        if (!lm) break;
        if constexpr (!FINAL) {
            if (next >= c.qcount && __builtin_popcountll(lm) < SV_QTHR) {
                wave_lds_sync();                         // (every entry has been read)
                if (live) {
                    const int pos = mbcnt(lm);
                    c.W->qRec[pos] = make_uint2(code, qy | ((unsigned)iters << 24));
                    c.W->qU1[pos] = u1;
                    c.W->qU2[pos] = u2;
                    SV_WIT(c.W->qL0[pos] = l0;)
                }
                left = __builtin_popcountll(lm);
                wave_lds_sync();
                break;
            }
        }
        c.n_dit += (unsigned)__builtin_popcountll(lm);
#ifdef SV_PROF
        SV_CNT(c.pt[3] += 1);
#endif
        bool fin = false, surv = false;
        if (live) {
            F val2 = F(0), l2 = F(0), la = F(0), mlim = F(0);
            const int st = sv_step<ML, F, NS>(c, rows, s1, s2, u1, u2, val2, l2, la, mlim);
            iters++;
            SV_WIT(unsigned wst = 0u;)
            if (st == 3 || iters >= 40) {
                surv = fin = true;               // ill-conditioned / stuck: the finish kernel solves it in FP64
                SV_WIT(wst = 6u;)
            } else if (st != 2) {
                // the bound finishes a candidate as soon as it applies (not in the full-solve mode, which iterates every
                // candidate to the coarse tolerance first); a converged candidate it does not finish is a contender -- once
                // the decrement is below fine_l2 (FP64: the gap of the bound is then a fraction of a unit; float: at once,
                // the margin of the single-precision sums dominates anyway).  The finish kernel decides exactly.
                const bool beyond = sv_beyond<ML, F, NS>(c, val2, l2, sv_sqrt(l2), la);
                if (beyond && (!c.no_dismiss || st == 1)) {
                    fin = true;
                    SV_WIT(wst = st == 1 ? 2u : 4u;)
                } else if (st == 1 && l2 < c.fine_l2) {
                    surv = fin = true;
                    SV_WIT(wst = 5u;)
                }
            }
            SV_WIT(if (fin) sv_witness<ML, F, NS>(c, qy >> 8, wst, (unsigned)iters, l0, l2, val2, s1, s2, u1, u2, mlim);)
        }
        if (surv) sv_survivor_rows<ML, F, NS>(c, rows, qy >> 8, iters >= 40 ? F(__builtin_nanf("")) : u1, u2);
        if (fin) {
            live = false;
        }
    }
    c.qcount = left;
#ifdef SV_PROF
    SV_CNT(c.pt[4] += 1);
    SV_CYC(c.pt[3] += __builtin_amdgcn_s_memtime() - pt0);
#endif
}

// ---- first evaluation, shared between the children of one last-level node -------------------------------------------
// In homogeneous coordinates w = (w0, u1, u2) a likelihood term is q_i = w0 + x_i u1 + y_i u2 -- it does not depend on the
// candidate's normalisation z = (1, s1, s2) (column sums / N), which only enters through the slice z.w = 1 the candidate's
// mixture lives on.  Candidates that differ in their LAST row only (the children of one node of the last expanded level)
// therefore share, at a common point w, the value / gradient / Hessian sums of ALL their other terms:
//      L = sum R log2 q,   T = sum (R/q) (1, x, y),   W = sum (R/q^2) (1, x, y)(1, x, y)^T         (10 numbers)
// Phase P: the lane that owns the node computes them once (group tile of the prefix + the node's own path rows) at the
// lane's chain point.  Phase C: one lane per CHILD adds the child's last row, restricts gradient and Hessian to the tangent
// space of the child's slice (d0 = -s1 d1 - s2 d2), and has the Newton decrement and the lower bound of the child's optimum
// for ~1/3 of the work of a full evaluation.  NLL = K0 - ln2 (L - Rtot log2(z.w)): scale invariant, so w needs no
// normalisation.  Children the bound cannot finish go to the queue and continue with full evaluations at their own iterate.
// Record of a node (SvPlanes, 16 numbers): 0 L, 1..3 T0 T1 T2, 4..9 W00 W01 W02 W11 W12 W22, 10 11 S1 S2, 12..14 w0 u1 u2,
// 15 sum R |log2 q| (F = double: the error bound of the single-precision logarithms, sv_beyond).
#ifndef SV_LEAN_FIRST
#define SV_LEAN_FIRST 1   // tight full-solve modes: the shared evaluation without logarithms, value and bound (the second one has them)
#endif
#ifndef SV_THIRD
#define SV_THIRD 1        // ... and with the cubic correction of its step (third-order sums in the nodes' records; F = double)
#endif
template <int ML, class F, int NS>
__device__ __forceinline__ bool sv_third(const SvCtx<ML, F, NS> &c) { return SV_THIRD && SV_LEAN_FIRST && sizeof(F) == 8 && c.second; }
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_shared_point(const SvCtx<ML, F, NS> &c, F S1, F S2, F &w0, F &u1, F &u2) {
    const F S1c = sv_bcast0(S1), S2c = sv_bcast0(S2);
    const bool c_ok = S1c > F(0) && S2c > F(0);
    const F c1 = F(1.0 / 3.0) * sv_rcp(c_ok ? S1c : F(1)), c2 = F(1.0 / 3.0) * sv_rcp(c_ok ? S2c : F(1));
    const F b0 = sv_bcast0(c.wn0), b1 = sv_bcast0(c.wn1), b2 = sv_bcast0(c.wn2);
    const bool have = b0 == b0;
    w0 = have ? sv_fma(F(1.0 - SV_PULL), b0, F(SV_PULL / 3.0)) : F(1.0 / 3.0);
    u1 = have ? sv_fma(F(1.0 - SV_PULL), b1, F(SV_PULL) * c1) : c1;
    u2 = have ? sv_fma(F(1.0 - SV_PULL), b2, F(SV_PULL) * c2) : c2;
}

// The tight modes' nodes (sv_third; F = double): the shared sums only shape the children's starting points (sv_child_eval_third), so they
// are taken in SINGLE PRECISION, two terms per packed instruction -- T = sum R z / q, W = sum R z z^T / q^2 and the third-order sums
// V_abc = sum R z_a z_b z_c / q^3 (z = (1, x, y); order 000 001 002 011 012 022 111 112 122 222) the cubic correction is made of: 34
// vector instructions per pair of terms where the FP64 sums with V in floats behind them took ~95.  Accumulation over ~19 terms leaves
// the step ~2e-5 of itself off, 4e-7 of the point: below what the correction leaves.  No value: a term outside the domain shows in the
// smallest q.  The column sums stay doubles (they are every later evaluation's).  Contraction off, every multiply-add written out:
// the witness build must take the same roundings.
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_parent_third(SvCtx<ML, F, NS> &c, bool take, unsigned code) {
#pragma clang fp contract(off)
    typedef float f2 __attribute__((ext_vector_type(2)));
    float px[ML - 1], py[ML - 1];
    F S1 = c.S1p, S2 = c.S2p;
#pragma unroll
    for (int j = 0; j < ML - 1; j++) {
        const unsigned r16 = c.S->row16[(code >> (6 * j)) & 63u];
        px[j] = (float)(r16 & 0xffu);
        py[j] = (float)(r16 >> 8);
        S1 = sv_fma((F)(r16 & 0xffu), c.leafN[j], S1);
        S2 = sv_fma((F)(r16 >> 8), c.leafN[j], S2);
    }
    F w0d, u1d, u2d;
    sv_shared_point<ML, F, NS>(c, S1, S2, w0d, u1d, u2d);
    const bool sums_ok = S1 > F(0) && S2 > F(0);
    if (c.lane == 0) c.W->shp.set(w0d, u1d, u2d);             // (read by the children after the syncs below)
    const float w0 = (float)w0d, u1 = (float)u1d, u2 = (float)u2d;
    const f2 vw0 = {w0, w0}, vu1 = {u1, u1}, vu2 = {u2, u2}, zero = {0.0f, 0.0f};
    f2 A[19];                                                  // T0 T1 T2 W00 W01 W02 W11 W12 W22 V000 .. V222
#pragma unroll
    for (int k = 0; k < 19; k++) A[k] = zero;
    float qmin = __builtin_inff();
    auto body = [&](f2 x, f2 y, f2 R) {
        const f2 q = __builtin_elementwise_fma(x, vu1, __builtin_elementwise_fma(y, vu2, vw0));
        qmin = __builtin_fminf(qmin, __builtin_fminf(q.x, q.y));
        const f2 w = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
        const f2 t = R * w, tw = t * w, twx = tw * x, twy = tw * y;
        A[0] += t;
        A[1] = __builtin_elementwise_fma(t, x, A[1]);
        A[2] = __builtin_elementwise_fma(t, y, A[2]);
        A[3] += tw; A[4] += twx; A[5] += twy;
        A[6] = __builtin_elementwise_fma(twx, x, A[6]);
        A[7] = __builtin_elementwise_fma(twx, y, A[7]);
        A[8] = __builtin_elementwise_fma(twy, y, A[8]);
        const f2 h = tw * w, hx = h * x, hy = h * y, hxx = hx * x, hxy = hx * y, hyy = hy * y;
        A[9] += h; A[10] += hx; A[11] += hy; A[12] += hxx; A[13] += hxy; A[14] += hyy;
        A[15] = __builtin_elementwise_fma(hxx, x, A[15]);
        A[16] = __builtin_elementwise_fma(hxx, y, A[16]);
        A[17] = __builtin_elementwise_fma(hxy, y, A[17]);
        A[18] = __builtin_elementwise_fma(hyy, y, A[18]);
    };
    // the group tile's part, once per round: lane p takes pair p, the partial sums (and the smallest q) meet in LDS -- the record
    // planes, free until the records are written below --, twenty lanes add them up, every lane starts its path rows from the totals
    float tot[20];
    {
        float *scr = (float *)&c.W->par;
        constexpr int TOT = 704;                               // (33 pairs x 20 floats before it; the planes hold 2048)
        if (c.lane < c.GP) {
            const Sv4<F> xy = c.W->fXY[c.lane];
            const typename SvWt<F>::T rr = c.W->fRR[c.lane];
            body(f2{(float)xy.x, (float)xy.y}, f2{(float)xy.z, (float)xy.w}, f2{(float)rr.x, (float)rr.y});
            float4 *dst = (float4 *)(scr + 20 * c.lane);
#pragma unroll
            for (int k = 0; k < 4; k++) dst[k] = make_float4(A[4 * k].x + A[4 * k].y, A[4 * k + 1].x + A[4 * k + 1].y, A[4 * k + 2].x + A[4 * k + 2].y, A[4 * k + 3].x + A[4 * k + 3].y);
            dst[4] = make_float4(A[16].x + A[16].y, A[17].x + A[17].y, A[18].x + A[18].y, qmin);
        }
        wave_lds_sync();
        if (c.lane < 20) {
            float t = c.lane == 19 ? __builtin_inff() : 0.0f;
            for (int p = 0; p < c.GP; p++) {
                const float v = scr[20 * p + c.lane];
                t = c.lane == 19 ? __builtin_fminf(t, v) : t + v;
            }
            scr[TOT + c.lane] = t;
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 5; k++) {
            const float4 t4 = ((const float4 *)(scr + TOT))[k];
            tot[4 * k] = t4.x; tot[4 * k + 1] = t4.y; tot[4 * k + 2] = t4.z; tot[4 * k + 3] = t4.w;
        }
        wave_lds_sync();                                       // (the planes are rewritten next)
    }
    if (take) {
#pragma unroll
        for (int k = 0; k < 19; k++) A[k] = f2{tot[k], 0.0f};
        qmin = tot[19];
        // the ML - 1 path rows: pairs, an odd one with a copy of itself of weight 0
#pragma unroll
        for (int j = 0; j + 1 < ML - 1; j += 2)
            body(f2{px[j], px[j + 1]}, f2{py[j], py[j + 1]}, f2{(float)c.leafRf[j], (float)c.leafRf[j + 1]});
        if ((ML - 1) & 1) body(f2{px[ML - 2], px[ML - 2]}, f2{py[ML - 2], py[ML - 2]}, f2{(float)c.leafRf[ML - 2], 0.0f});
        float rf[20];
#pragma unroll
        for (int k = 0; k < 19; k++) rf[k] = A[k].x + A[k].y;
        rf[19] = 0.0f;
        const bool usable = sums_ok && qmin > 0.0f && fabsf(rf[0]) < __builtin_inff();
        c.W->par.put_third(c.lane, S1, S2, rf);
        c.W->pcode[c.lane] = code | (usable ? 0x80000000u : 0u);
    }
}

template <int ML, class F, int NS>
__device__ __forceinline__ void sv_parent(SvCtx<ML, F, NS> &c, bool take, unsigned code) {
    if constexpr (sizeof(F) == 8) {
        if (sv_third<ML, F, NS>(c)) {
            sv_parent_third<ML, F, NS>(c, take, code);
            return;
        }
    }
    typedef typename SvVec<F>::v2 v2;
    // path rows D .. D+ML-2 of the node (its ancestors in the expanded levels and itself)
    F px[ML - 1], py[ML - 1];
    F S1 = c.S1p, S2 = c.S2p;
#pragma unroll
    for (int j = 0; j < ML - 1; j++) {
        const unsigned r16 = c.S->row16[(code >> (6 * j)) & 63u];
        px[j] = (F)(r16 & 0xffu);
        py[j] = (F)(r16 >> 8);
        S1 = sv_fma(px[j], c.leafN[j], S1);
        S2 = sv_fma(py[j], c.leafN[j], S2);
    }
    F w0, u1, u2;
    sv_shared_point<ML, F, NS>(c, S1, S2, w0, u1, u2);         // (the same w for every node of the round)
    const bool sums_ok = S1 > F(0) && S2 > F(0);
    v2 L = {F(0), F(0)}, T0 = L, T1 = L, T2 = L, W00 = L, W01 = L, W02 = L, W11 = L, W12 = L, W22 = L, LA = L;
    const v2 vw0 = {w0, w0}, vu1 = {u1, u1}, vu2 = {u2, u2};
    // (rho = sqrt R: gamma = rho / q, T = sum rho gamma (1, x, y), W = sum gamma^2 (1, x, y)(1, x, y)^T; a term outside the
    // domain shows as a NaN / an infinity in L, see sv_step)
    auto body = [&](v2 x, v2 y, v2 R, v2 rho) {
        v2 q = __builtin_elementwise_fma(x, vu1, __builtin_elementwise_fma(y, vu2, vw0));
        F wx, wy, lx, ly;
        sv_rcp_lg2(q.x, wx, lx);
        sv_rcp_lg2(q.y, wy, ly);
        const v2 w = {wx, wy}, l = {lx, ly};
        L = __builtin_elementwise_fma(R, l, L);
        if constexpr (sizeof(F) == 8) LA = __builtin_elementwise_fma(R, v2{sv_abs(l.x), sv_abs(l.y)}, LA);
        if constexpr (sizeof(F) == 8) {
            v2 ga = rho * w;
            T0 = __builtin_elementwise_fma(rho, ga, T0);
            v2 gx = ga * x, gy = ga * y;
            T1 = __builtin_elementwise_fma(rho, gx, T1);
            T2 = __builtin_elementwise_fma(rho, gy, T2);
            W00 = __builtin_elementwise_fma(ga, ga, W00);
            W01 = __builtin_elementwise_fma(ga, gx, W01);
            W02 = __builtin_elementwise_fma(ga, gy, W02);
            W11 = __builtin_elementwise_fma(gx, gx, W11);
            W12 = __builtin_elementwise_fma(gx, gy, W12);
            W22 = __builtin_elementwise_fma(gy, gy, W22);
        } else {
            v2 t = R * w;
            T0 += t;
            T1 = __builtin_elementwise_fma(t, x, T1);
            T2 = __builtin_elementwise_fma(t, y, T2);
            v2 tw = t * w;
            W00 += tw;
            v2 twx = tw * x, twy = tw * y;
            W01 += twx;
            W02 += twy;
            W11 = __builtin_elementwise_fma(twx, x, W11);
            W12 = __builtin_elementwise_fma(twx, y, W12);
            W22 = __builtin_elementwise_fma(twy, y, W22);
        }
    };
    // ---- the group tile's part of the sums, ONCE PER ROUND (round 4): every node of the round evaluates at the same point, so
    // the ~13 group terms of the prefix -- two thirds of a node's terms -- are the same numbers in all 64 lanes.  Lane p takes
    // pair p of the tile; the partial sums meet in LDS (the record planes, free until the nodes' records are written below),
    // twelve lanes add them up, and every lane starts its own path rows from the totals.
    F tot[12];
    {
        F *scr = (F *)&c.W->par;
        constexpr int TOT = 512;
        if (c.lane < c.GP) {
            const Sv4<F> xy = c.W->fXY[c.lane];
            const typename SvWt<F>::T rr = c.W->fRR[c.lane];
            body(v2{xy.x, xy.y}, v2{xy.z, xy.w}, v2{rr.x, rr.y}, sv_rho<F>(rr));
            const F part[12] = {L.x + L.y, T0.x + T0.y, T1.x + T1.y, T2.x + T2.y, W00.x + W00.y, W01.x + W01.y, W02.x + W02.y, W11.x + W11.y,
                                W12.x + W12.y, W22.x + W22.y, LA.x + LA.y, F(0)};
            Sv2<F> *dst = (Sv2<F> *)(scr + 12 * c.lane);
#pragma unroll
            for (int k = 0; k < 6; k++) dst[k] = Sv2<F>{part[2 * k], part[2 * k + 1]};
        }
        wave_lds_sync();
        if (c.lane < 12) {
            F t = F(0);
            for (int p = 0; p < c.GP; p++) t += scr[12 * p + c.lane];
            scr[TOT + c.lane] = t;
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const Sv2<F> t2 = ((const Sv2<F> *)(scr + TOT))[k];
            tot[2 * k] = t2.x;
            tot[2 * k + 1] = t2.y;
        }
        wave_lds_sync();                                   // (the planes are rewritten next)
    }
    if (take) {
        L = v2{tot[0], F(0)}; T0 = v2{tot[1], F(0)}; T1 = v2{tot[2], F(0)}; T2 = v2{tot[3], F(0)};
        W00 = v2{tot[4], F(0)}; W01 = v2{tot[5], F(0)}; W02 = v2{tot[6], F(0)}; W11 = v2{tot[7], F(0)};
        W12 = v2{tot[8], F(0)}; W22 = v2{tot[9], F(0)}; LA = v2{tot[10], F(0)};
        // the ML - 1 path rows: pairs, an odd one with a copy of itself of weight 0
#pragma unroll
        for (int j = 0; j + 1 < ML - 1; j += 2)
            body(v2{px[j], px[j + 1]}, v2{py[j], py[j + 1]}, v2{c.leafRf[j], c.leafRf[j + 1]}, v2{c.leafRho[j], c.leafRho[j + 1]});
        if ((ML - 1) & 1) body(v2{px[ML - 2], px[ML - 2]}, v2{py[ML - 2], py[ML - 2]}, v2{c.leafRf[ML - 2], F(0)}, v2{c.leafRho[ML - 2], F(0)});
        const F Lsum = L.x + L.y;
        const bool usable = sums_ok && sv_abs(Lsum) < F(__builtin_inff());
        const F rec[16] = {Lsum, T0.x + T0.y, T1.x + T1.y, T2.x + T2.y, W00.x + W00.y, W01.x + W01.y, W02.x + W02.y, W11.x + W11.y,
                           W12.x + W12.y, W22.x + W22.y, S1, S2, w0, u1, u2, LA.x + LA.y};
        c.W->par.put(c.lane, rec);
        c.W->pcode[c.lane] = code | (usable ? 0x80000000u : 0u);
    }
}

#ifndef SV_THIRD_MIN
#define SV_THIRD_MIN 48   // lanes that must still need an evaluation after the in-place second for a THIRD to be taken in place too (SvCtx.third_min; 65: never)
#endif
// What the shared first evaluation of one child comes to (sv_child_eval): all a lane carries from the arithmetic to the
// bookkeeping, so that the arithmetic of SEVERAL children per lane can be one straight-line block.
template <class F>
struct SvChild {
    bool act, regular, ev, push, surv;
    unsigned slot, code, off;
    F qu1, qu2;                // where the child continues in the queue
    F n1, n2;                  // its stepped mixture, and the stepped point itself (u1, u2; w0 = 1 - n1 - n2): the lane's chain point, valid if `chain`
    F c1, c2;
    bool chain;
    F s1, s2;                  // the child's column sums / N (sv_second: a second evaluation in place needs them)
#ifdef SV_WITNESS
    bool wdone;                // finished by this evaluation (converged and valued / dismissed by the bound)
    bool wconv;
    F wl2, wval2, ws1, ws2, wmlim;
#endif
};

// Phase C arithmetic for the child k of the round, BRANCH-FREE: every lane computes everything (on harmless inputs where the
// child does not exist or has no usable shared point) and the outcome is a handful of selects.  A lane's evaluation is one
// chain of ~100 dependent operations; with two or three waves per SIMD that chain's latency, not the issue rate, was what the
// children phase cost (44 % of the search kernel's wave cycles, profiles/r3).  Without branches the scheduler interleaves the
// chains of the SV_CPL children a lane takes per trip.
// The tight modes' shared step (sv_third; F = double), in SINGLE PRECISION throughout: it shapes a starting point -- the child is
// iterated and valued by its private evaluations in FP64 -- and the cancellation in G = T_j - s_j T_0 (G / T ~ sqrt l2 ~ 1e-2) leaves
// the step ~5e-6 of itself off, 1e-7 of the point: two orders below what the cubic correction leaves.  Against the FP64 form with the
// correction in single precision: ~90 vector instructions instead of ~165 per child, six 16-byte LDS reads instead of eight.
//   The CUBIC correction (Chebyshev's method; round 6): with D = (-s1 d1 - s2 d2, d1, d2) the Newton direction in w, the third
// derivative of sum R ln q along it is 2 V[D, D] (V = the node's third-order sums + the child's own last row), and the step that
// cancels the error's quadratic term as well is d + H^-1 c, c_j = V[D, D]_j - s_j V[D, D]_0: the decrement the FIRST private
// evaluation finds falls from ~l2^2 to ~l2^3 -- below the certificates' limits for nine candidates in ten instead of five
// (profiles/r6/NOTES.md section 10).  Only with a full step, and dropped where it is not small against the Newton step itself (far
// from the optimum the cubic model says nothing).  No value, no bound, no convergence here: every child takes a private evaluation.
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_child_eval_third(const SvCtx<ML, F, NS> &c, int lo, int k, int nrec, SvChild<F> &o) {
    // (every multiply-add written out, none left to the optimizer: the witness build is a second compilation of this source, and the
    // two must take the same roundings -- tests/test_gpu_round5.py compares their counters to the unit)
#pragma clang fp contract(off)
    const F Nl = c.leafN[ML - 1];
    const float Rl = (float)c.leafRf[ML - 1];
    o.act = k < nrec;
    const unsigned kd = o.act ? c.W->kid[lo + k] : 0u;
    o.slot = kd & 0xffu;
    const unsigned pl = kd >> 8;
    F S1, S2;
    float f[20];
    c.W->par.get_third(pl, S1, S2, f);
    o.code = c.W->pcode[pl];
    const unsigned r16 = c.S->row16[o.slot];
    const float x = (float)(r16 & 0xffu), y = (float)(r16 >> 8);
    const F s1 = sv_fma((F)(r16 & 0xffu), Nl, S1), s2 = sv_fma((F)(r16 >> 8), Nl, S2);       // (the column sums as every later evaluation takes them)
    o.regular = s1 > F(0) && s2 > F(0);
    o.off = c.done + (unsigned)k;
    const float w0 = (float)c.W->shp.get(0), u1 = (float)c.W->shp.get(1), u2 = (float)c.W->shp.get(2);
    const float fs1 = (float)s1, fs2 = (float)s2;
    const float q = __builtin_fmaf(x, u1, __builtin_fmaf(y, u2, w0));
    o.ev = o.act && o.regular && (o.code >> 31) && q > 0.0f;
    const float w = __builtin_amdgcn_rcpf(q);
    const float t = Rl * w, tw = t * w, twx = tw * x, twy = tw * y;
    const float T0 = f[0] + t, T1 = __builtin_fmaf(t, x, f[1]), T2 = __builtin_fmaf(t, y, f[2]);
    const float W00 = f[3] + tw, W01 = f[4] + twx, W02 = f[5] + twy;
    const float W11 = __builtin_fmaf(twx, x, f[6]), W12 = __builtin_fmaf(twx, y, f[7]), W22 = __builtin_fmaf(twy, y, f[8]);
    const float G1 = __builtin_fmaf(-fs1, T0, T1), G2 = __builtin_fmaf(-fs2, T0, T2);
    const float A1 = __builtin_fmaf(-fs1, W00, W01), A2 = __builtin_fmaf(-fs2, W00, W02);
    const float H11 = __builtin_fmaf(-fs1, A1, __builtin_fmaf(-fs1, W01, W11));
    const float H12 = __builtin_fmaf(-fs2, A1, __builtin_fmaf(-fs1, W02, W12));
    const float H22 = __builtin_fmaf(-fs2, A2, __builtin_fmaf(-fs2, W02, W22));
    const float hh = H11 * H22, det = __builtin_fmaf(-H12, H12, hh);
    const float zw = __builtin_fmaf(fs1, u1, __builtin_fmaf(fs2, u2, w0));
    const bool cond_ok = det > (float)N3_COND_MIN * hh && zw > 0.0f;
    const float idet = __builtin_amdgcn_rcpf(det);
    float d1 = __builtin_fmaf(H22, G1, -(H12 * G2)) * idet, d2 = __builtin_fmaf(H11, G2, -(H12 * G1)) * idet;
    const float l2 = __builtin_fmaf(G1, d1, G2 * d2) * (float)c.inv_Rtot;
    const bool num_ok = l2 == l2 && fabsf(d1) + fabsf(d2) < 1e30f;
    {
        const float *V = f + 9;      // 000 001 002 011 012 022 111 112 122 222
        const float D0 = __builtin_fmaf(-fs1, d1, -(fs2 * d2));
        const float e = __builtin_fmaf(x, d1, __builtin_fmaf(y, d2, D0));
        const float he = tw * w * e * e;
        const float p00 = D0 * D0, p01 = 2.0f * D0 * d1, p02 = 2.0f * D0 * d2, p11 = d1 * d1, p12 = 2.0f * d1 * d2, p22 = d2 * d2;
        const float o0 = __builtin_fmaf(V[0], p00, __builtin_fmaf(V[1], p01, __builtin_fmaf(V[2], p02, __builtin_fmaf(V[3], p11, __builtin_fmaf(V[4], p12, __builtin_fmaf(V[5], p22, he))))));
        const float o1 = __builtin_fmaf(V[1], p00, __builtin_fmaf(V[3], p01, __builtin_fmaf(V[4], p02, __builtin_fmaf(V[6], p11, __builtin_fmaf(V[7], p12, __builtin_fmaf(V[8], p22, he * x))))));
        const float o2 = __builtin_fmaf(V[2], p00, __builtin_fmaf(V[4], p01, __builtin_fmaf(V[5], p02, __builtin_fmaf(V[7], p11, __builtin_fmaf(V[8], p12, __builtin_fmaf(V[9], p22, he * y))))));
        const float c1 = __builtin_fmaf(-fs1, o0, o1), c2 = __builtin_fmaf(-fs2, o0, o2);
        const float k1 = __builtin_fmaf(H22, c1, -(H12 * c2)) * idet, k2 = __builtin_fmaf(H11, c2, -(H12 * c1)) * idet;
        const bool small = fabsf(k1) + fabsf(k2) < 0.5f * (fabsf(d1) + fabsf(d2)) && l2 <= 0.09f;
        d1 += small ? k1 : 0.0f;
        d2 += small ? k2 : 0.0f;
    }
    float step = 1.0f;
    if (ballot64(l2 > 0.09f)) {                                                                 // (damped phase: rare, the branch is wave-uniform)
        float l2l = l2;
        asm volatile("" : "+v"(l2l));
        step = l2 > 0.09f ? __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(l2l)) : 1.0f;
    }
    const float sc = __builtin_amdgcn_rcpf(zw);
    const float v1 = __builtin_fmaf(step, d1, u1) * sc, v2 = __builtin_fmaf(step, d2, u2) * sc;
    const bool good = o.ev && cond_ok && num_ok;
    const float n1 = fs1 * v1, n2 = fs2 * v2;
    o.n1 = (F)n1;
    o.n2 = (F)n2;
    o.c1 = (F)v1;
    o.c2 = (F)v2;
    o.chain = good && fabsf(n1) + fabsf(n2) < 1e6f;
    o.surv = false;
    o.push = o.act && o.regular;
    o.qu1 = good ? (F)v1 : F(__builtin_nanf(""));
    o.qu2 = (F)v2;
    o.s1 = s1;
    o.s2 = s2;
#ifdef SV_WITNESS
    o.wdone = false;
    o.wmlim = F(0);
    o.wconv = false;
    o.wl2 = (F)l2;
    o.wval2 = F(0);
    o.ws1 = s1;
    o.ws2 = s2;
#endif
}

template <int ML, class F, int NS>
__device__ __forceinline__ void sv_child_eval(const SvCtx<ML, F, NS> &c, int lo, int k, int nrec, SvChild<F> &o) {
    if constexpr (sizeof(F) == 8) {
        if (sv_third<ML, F, NS>(c)) {
            sv_child_eval_third<ML, F, NS>(c, lo, k, nrec, o);
            return;
        }
    }
    const F Rl = c.leafRf[ML - 1], Nl = c.leafN[ML - 1];
    o.act = k < nrec;
    const unsigned kd = o.act ? c.W->kid[lo + k] : 0u;
    o.slot = kd & 0xffu;
    const unsigned pl = kd >> 8;
    F P[16];
    c.W->par.get(pl, P);
    o.code = c.W->pcode[pl];
    const unsigned r16 = c.S->row16[o.slot];
    const F x = (F)(r16 & 0xffu), y = (F)(r16 >> 8);
    const F s1 = sv_fma(x, Nl, P[10]), s2 = sv_fma(y, Nl, P[11]);
    o.regular = s1 > F(0) && s2 > F(0);
    o.off = c.done + (unsigned)k;
    const F w0 = P[12], u1 = P[13], u2 = P[14];
    const F q = sv_fma(x, u1, sv_fma(y, u2, w0));
    o.ev = o.act && o.regular && (o.code >> 31) && q > F(0);
    // (a lane without a usable point, or with ill-conditioned sums, computes on whatever it has: infinities and NaNs cost nothing
    // and every decision below is behind `good` = ev && cond_ok && num_ok -- no selects in the arithmetic)
    // The tight full-solve modes (c.second) take a second evaluation of every child anyway and value it there: the shared one is
    // for the STEP alone -- no logarithms, no value, no bound (SV_LEAN_FIRST; wave-uniform branches).
    const bool lean = SV_LEAN_FIRST && c.second;
    F w, lq = F(0);
    if (lean) w = sv_rcp(q); else sv_rcp_lg2(q, w, lq);
    const F t = Rl * w, tw = t * w, twx = tw * x, twy = tw * y;
    const F T0 = P[1] + t, T1 = sv_fma(t, x, P[2]), T2 = sv_fma(t, y, P[3]);
    const F W00 = P[4] + tw, W01 = P[5] + twx, W02 = P[6] + twy;
    const F W11 = sv_fma(twx, x, P[7]), W12 = sv_fma(twx, y, P[8]), W22 = sv_fma(twy, y, P[9]);
    // tangent space of the child's slice z.w = const: d = (-s1 d1 - s2 d2, d1, d2)
    const F G1 = sv_fma(-s1, T0, T1), G2 = sv_fma(-s2, T0, T2);
    const F A1 = sv_fma(-s1, W00, W01), A2 = sv_fma(-s2, W00, W02);      // W0j - s_j W00
    const F H11 = sv_fma(-s1, A1, sv_fma(-s1, W01, W11));
    const F H12 = sv_fma(-s2, A1, sv_fma(-s1, W02, W12));
    const F H22 = sv_fma(-s2, A2, sv_fma(-s2, W02, W22));
    const F hh = H11 * H22, det = sv_fma(-H12, H12, hh);
    const F zw = sv_fma(s1, u1, sv_fma(s2, u2, w0));
    const bool cond_ok = det > (F)N3_COND_MIN * hh && zw > F(0);          // else: ill-conditioned for these sums
    const F idet = sv_rcp(det);
    const F d1 = (H22 * G1 - H12 * G2) * idet, d2 = (H11 * G2 - H12 * G1) * idet;
    const F l2 = (G1 * d1 + G2 * d2) * c.inv_Rtot;
    F sc, lz = F(0);
    if (lean) sc = sv_rcp(zw); else sv_rcp_lg2(zw, sc, lz);
    const bool num_ok = l2 == l2 && sv_abs(d1) + sv_abs(d2) < F(1e30);
    F step = F(1);
    if (ballot64(l2 > F(0.09))) {                                                              // (damped phase: rare, the branch is wave-uniform)
        F l2l = l2;
        asm volatile("" : "+v"(l2l));                                                           // (... and stays one: not if-converted)
        step = l2 > F(0.09) ? sv_rcp(F(1) + sv_sqrt(l2l)) : F(1);
    }
    // the stepped point on the child's own slice (z.d = 0, so z.w stays): mixture n_j = s_j u_j / z.w
    const F v1 = sv_fma(step, d1, u1) * sc, v2 = sv_fma(step, d2, u2) * sc;
    const bool good = o.ev && cond_ok && num_ok;
    o.n1 = s1 * v1;
    o.n2 = s2 * v2;
    o.c1 = v1;
    o.c2 = v2;
    o.chain = good && sv_abs(o.n1) + sv_abs(o.n2) < F(1e6);
    // (same decisions as in sv_drain)
    bool conv = false, done = false;
    F val2 = F(0);
    if (!lean) {
        const F L = sv_fma(Rl, lq, P[0]);
        val2 = sv_fma(-c.rtot_f, lz, L);
        F la = F(0);
        if constexpr (sizeof(F) == 8) la = sv_fma(c.rtot_f, sv_abs(lz), sv_fma(Rl, sv_abs(lq), P[15]));
        F cv = c.conv_l2;
        if (c.mu_c > F(0)) {
            // (n3_mu_tol: the child's own coordinates are the shared point's divided by z.w -- H scales with its square, u with its inverse)
            const F ml = sv_mu_limit<ML, F, NS>(c, H11 * zw * zw, H22 * zw * zw, det * (zw * zw) * (zw * zw), s1, s2, u1 * sc, u2 * sc);
            cv = sv_min(cv, ml);
            SV_WIT(o.wmlim = ml;)
        }
        conv = l2 < cv && l2 * c.rtot_over_rmin < F(0.25);
        const bool beyond = sv_beyond<ML, F, NS>(c, val2, l2, sv_sqrt(l2), la);
        done = good && beyond && (!c.no_dismiss || conv);         // the bound (search) / converged and valued beyond the window (full solve)
    }
    o.surv = good && !done && conv && l2 < c.fine_l2;
    // queued: another step from the stepped point -- or, without a usable shared point / with ill-conditioned sums, from the
    // simplex centre (NaN marks that: sv_drain has the record's column sums anyway)
    o.push = o.act && o.regular && !done && !o.surv;
    o.qu1 = good ? v1 : F(__builtin_nanf(""));
    o.qu2 = v2;
    o.s1 = s1;
    o.s2 = s2;
#ifdef SV_WITNESS
    o.wdone = done;
    if (!(c.mu_c > F(0)) || lean) o.wmlim = F(0);
    o.wconv = conv;
    o.wl2 = l2;
    o.wval2 = val2;
    o.ws1 = s1;
    o.ws2 = s2;
#endif
}

#ifndef SV_CPL
#define SV_CPL 1          // children a lane evaluates per trip (2 measured 3 % slower: the phase is issue bound, not latency bound, and two sets of live values spill)
#endif

// The candidates [0, total) of the round (less the task window), SV_CPL children per lane and trip.
template <int ML, class F, int NS>
__device__ __forceinline__ void sv_children(SvCtx<ML, F, NS> &c, int total) {
    const unsigned long long sk = c.skip < (unsigned long long)total ? c.skip : (unsigned long long)total;
    c.skip -= sk;
    const int lo = (int)sk;
    const unsigned long long room = (unsigned long long)(total - lo);
    const int nrec = (int)(room < (unsigned long long)c.remaining ? room : (unsigned long long)c.remaining);
    if (nrec <= 0) return;
#ifdef SV_PROF
    SV_CNT(c.pt[6] += (unsigned)nrec);
#endif
    for (int k0 = 0; k0 < nrec; k0 += WAVE * SV_CPL) {
#ifdef SV_PROF
        SV_CNT(c.pt[2] += 1);
#endif
        SvChild<F> ch[SV_CPL];
#pragma unroll
        for (int e = 0; e < SV_CPL; e++) sv_child_eval<ML, F, NS>(c, lo, k0 + e * WAVE + c.lane, nrec, ch[e]);
#pragma unroll
        for (int e = 0; e < SV_CPL; e++) {
            const SvChild<F> &o = ch[e];
            if (k0 + e * WAVE >= nrec) break;             // (wave-uniform)
            if (o.act && !o.regular) {
                degenerate_append(c.A.ctr, c.A.deg, c.A.deg_cap, c.base + o.off);
                atomicAdd(&c.A.ctr->degenerate, 1ull);
            }
            c.n_child += (unsigned)__builtin_popcountll(ballot64(o.ev));
            // The chain point of the NEXT round: the mean of the stepped points of the children of this round's last FULL trip
            // (single precision does: it is a starting point).  w is nearly the same for all candidates of a neighbourhood; the
            // mean takes the siblings' own shifts out, and 64 samples do it better than fewer or older ones -- one lane's last
            // child: 1.416 evaluations per candidate, a partial last trip 1.364, the first trip 1.447, the whole round 1.325,
            // the last full trip 1.310 (profiles/r4/NOTES.md).
            if (k0 + WAVE * SV_CPL <= nrec && k0 + 2 * WAVE * SV_CPL > nrec) {
                const float a0 = sv_wave_sum_f32(o.chain ? (float)(F(1) - o.n1 - o.n2) : 0.0f);
                const float a1 = sv_wave_sum_f32(o.chain ? (float)o.c1 : 0.0f), a2 = sv_wave_sum_f32(o.chain ? (float)o.c2 : 0.0f);
                const int an = __builtin_popcountll(ballot64(o.chain));
                if (an > 0) {
                    const float inv = __builtin_amdgcn_rcpf((float)an);
                    c.wn0 = (F)(a0 * inv);
                    c.wn1 = (F)(a1 * inv);
                    c.wn2 = (F)(a2 * inv);
                }
            }
#ifdef SV_WITNESS
            if (o.wdone || o.surv)
                sv_witness<ML, F, NS>(c, o.off, o.surv ? 5u : (o.wconv ? 1u : 3u), 1u, (float)o.wl2, o.wl2, o.wval2, o.ws1, o.ws2, o.c1, o.c2, o.wmlim);
#endif
            unsigned long long pm = ballot64(o.push);
            const unsigned long long sm = ballot64(o.surv);
            if (sm) {                                     // (rare: a contender straight from the shared evaluation)
                if (o.surv) {
                    unsigned rw[ML / 2];
                    sv_child_rows<ML, F, NS>(c, o.code, o.slot, rw);
                    sv_survivor<ML, F, NS>(c, rw, o.off, o.c1, o.c2);
                }
            }
            bool push = o.push;
            F qu1 = o.qu1, qu2 = o.qu2;
            unsigned evals = o.ev ? 1u : 0u;
            for (int pass = 0; c.second && pm && (pass == 0 || (pass < 3 && __builtin_popcountll(pm) >= c.third_min)); pass++) {
                // The tight full-solve modes: (nearly) every child needs a second evaluation -- taken HERE, in lock step, by the lane that
                // has the child at hand, instead of through the queue (push, pop, decode, column sums, a wave-step that may run half
                // empty): the queue is left with the ~10 % that need a third (and where most of the trip's lanes do -- SV_THIRD_MIN --
                // that one is taken here as well).  Same arithmetic, same decisions as sv_drain.
                SvRowsT<ML, SvTab<F, NS>::v> rows;
                sv_rows_load<ML, F, NS>(c, o.code, o.slot, rows);
                F u1 = qu1, u2 = qu2;
                if (ballot64(!(u1 == u1))) {            // (no usable shared point: from the simplex centre -- rare, behind a wave-uniform branch)
                    F fs1 = o.s1, fs2 = o.s2;
                    asm volatile("" : "+v"(fs1), "+v"(fs2));          // (the reciprocals stay in here: hoisted, they ran once per trip)
                    if (!(u1 == u1)) {
                        u1 = F(1.0 / 3.0) * sv_rcp(fs1);
                        u2 = F(1.0 / 3.0) * sv_rcp(fs2);
                    }
                }
                F val2 = F(0), l2 = F(0), la = F(0), mlim = F(0);
                const int st = sv_step<ML, F, NS>(c, rows, o.s1, o.s2, u1, u2, val2, l2, la, mlim);
                c.n_dit += (unsigned)__builtin_popcountll(pm);
                if (push) {
                    evals++;
                    bool fin = false, surv = false;
                    SV_WIT(unsigned wst = 0u;)
                    if (st == 3) {
                        surv = fin = true;
                        SV_WIT(wst = 6u;)
                    } else if (st != 2) {
                        const bool beyond = sv_beyond<ML, F, NS>(c, val2, l2, sv_sqrt(l2), la);
                        if (beyond && st == 1) {
                            fin = true;
                            SV_WIT(wst = 2u;)
                        } else if (st == 1 && l2 < c.fine_l2) {
                            surv = fin = true;
                            SV_WIT(wst = 5u;)
                        }
                    }
                    SV_WIT(if (fin) sv_witness<ML, F, NS>(c, o.off, wst, evals, o.qu1 == o.qu1 ? (float)o.wl2 : __builtin_nanf(""), l2, val2, o.s1, o.s2, u1, u2, mlim);)
                    if (surv) sv_survivor_rows<ML, F, NS>(c, rows, o.off, u1, u2);
                    if (fin) push = false;
                    qu1 = u1;
                    qu2 = u2;
                }
                pm = ballot64(push);
            }
            if (pm) {
                if (c.qcount + __builtin_popcountll(pm) > SV_QCAP) sv_drain<ML, F, NS, false>(c);
                if (push) {
                    const int pos = c.qcount + mbcnt(pm);
                    c.W->qRec[pos] = make_uint2(o.code, o.slot | (o.off << 8) | (evals << 24));   // (top byte: evaluations so far)
                    c.W->qU1[pos] = qu1;
                    c.W->qU2[pos] = qu2;
                    SV_WIT(c.W->qL0[pos] = o.qu1 == o.qu1 ? (float)o.wl2 : __builtin_nanf("");)
                }
                c.qcount += __builtin_popcountll(pm);
                wave_lds_sync();
            }
        }
    }
    c.done += (unsigned)nrec;
    c.remaining -= (unsigned)nrec;
}

// n3_child_dyn (n3_core.hpp) on the sieve's tables: the dynamic part of the edge test for a child that passed the masks
template <int ML, class F, int NS>
__device__ __forceinline__ bool sv_child_dyn(const SvCtx<ML, F, NS> &c, const N3State &par, int slot, N3State &out) {
    const unsigned rw = c.S->rowtab[slot];
    const int a = rw & 15, b = rw >> 4;
    int lo = par.lo, hi = par.hi;
    const int dx = a - par.a, dy = b - par.b;
    if (dx != 0 && dy != 0) {
        int t;
        if ((unsigned)(dx + SV_RIDX_W / 2) < (unsigned)SV_RIDX_W && (unsigned)(dy + SV_RIDX_W / 2) < (unsigned)SV_RIDX_W)
            t = c.S->ridx[(dy + SV_RIDX_W / 2) * SV_RIDX_W + (dx + SV_RIDX_W / 2)];
        else      // (only in searches with copy numbers above 7)
            t = ((const unsigned char *)(c.dynmask + (size_t)c.Q * c.NT1 * c.NT1))[(dy + N3_MAX_COPY) * N3_RIDX_W + (dx + N3_MAX_COPY)];
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

template <int ML, class F, int NS>
__device__ __forceinline__ unsigned long long sv_child_mask(const SvCtx<ML, F, NS> &c, const N3State &node, int l) {
    unsigned long long mk = c.S->smask[l][node.slot] & c.dynmask[((size_t)node.slot * c.NT1 + node.lo) * c.NT1 + (node.hi - 1)];
    return node.sw ? (mk & c.swm) : mk;
}

// Expand the nodes of leaf level LVL (LVL = 0: the prefix's last node; else the level's list [0 .. n_in)) in rank order.
template <int ML, int LVL, class F, int NS>
__device__ __forceinline__ void sv_expand(SvCtx<ML, F, NS> &c, int n_in) {
    constexpr bool last = (LVL == ML - 1);
    const int cap = last ? SV_KIDS : (LVL == ML - 2 ? SvCapL<F>::v : (LVL == 0 ? N3_MAX_Q : SV_CAP));
    int pos = 0;
    while (pos < n_in) {
#ifdef SV_PROF
        const unsigned long long pi = __builtin_amdgcn_s_memtime();
#endif
        const int i = pos + c.lane;
        bool live = i < n_in;
        N3State node = c.par;
        unsigned code = 0;                                 // slots of rows D .. D+LVL-1, 6 bits each
        uint2 e = make_uint2(0u, 0u);
        if (LVL > 0) {
            if (live) {
                e = (LVL == 1) ? c.W->list0[i] : (LVL == ML - 1) ? c.W->listL[i] : c.W->list[LVL >= 2 && LVL < ML - 1 ? LVL - 2 : 0][i];
                const N3State pst = n3_unpack(e.x);
                const unsigned slot = e.y >> 24;
                sv_child_dyn<ML, F, NS>(c, pst, (int)slot, node);
                code = (e.y & 0xffffffu) | (slot << (6 * (LVL > 0 ? LVL - 1 : 0)));
            }
        } else {
            live = c.lane == 0;
        }
        // A task starts `skip` leaves into its first prefix.  Whole subtrees that lie before that point are dropped by their
        // exact sizes from the counting table instead of being expanded and thrown away (rare path: first prefix of a task).
        if (LVL > 0 && c.skip > 0) {
            unsigned long long sz = 0;
            if (live) {
                const size_t ci = ((((size_t)(c.D + LVL - 1) * c.Q + node.slot) * 2 + node.sw) * c.NT1 + node.lo) * c.NT1 + (node.hi - 1);
                const u128 tv = c.cnt[ci];
                sz = (tv >> 64) ? ~0ull : (unsigned long long)tv;
            }
            int drop = 0;
            const int nl = n_in - pos < WAVE ? n_in - pos : WAVE;
            for (int l = 0; l < nl; l++) {
                const unsigned long long s_l = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(sz >> 32), l) << 32) |
                                               (unsigned)__builtin_amdgcn_readlane((int)sz, l);
                if (c.skip < s_l) break;
                c.skip -= s_l;
                drop++;
            }
            if (drop) {
                pos += drop;
                continue;
            }
        }
        unsigned long long mk = live ? sv_child_mask(c, node, LVL) : 0ull;
        const int cnt = __builtin_popcountll(mk);
        const int incl = sv_incl_scan(cnt);
        // (LVL = ML - 2: the list of last-level nodes starts behind the nodes carried over from the previous fill)
        const int carry_in = (LVL == ML - 2 && !last) ? c.carryL : 0;
        int t = __builtin_popcountll(ballot64(live && incl <= cap - carry_in));   // nodes whose children all fit (a prefix of the lanes)
        int total = __builtin_amdgcn_readlane(incl, t - 1);
        const int off = incl - cnt;
        int carry_out = 0;
        if constexpr (last) {
#if SV_FULL_TRIPS
            // (The tight modes, where a child costs a private evaluation or more: 78.5 -> 76.4 ms per 2^31 on the certified leg.  Where most
            // children are finished by the shared evaluation the fewer nodes per round cost the parent phase more than the trips gain:
            // the coarse FP64 leg 45.2 -> 46.1 ms -- not taken there.)
            // Whole trips only: with `full` = the largest multiple of 64 children the round's nodes reach, the nodes beyond it are left --
            // to the next round of this list, or (the list's last round) carried to the next fill.  A node has < 64 children, so a
            // round with >= 64 always keeps its first node; one with fewer is taken as it is when more nodes follow in the list (64
            // nodes of at most one child each), and carried whole when they do not.
            const bool more = pos + t < n_in;
            const int full = total & ~63;
            if (c.second && full != total && (more ? full >= 64 : !c.final_path)) {
                const int tt = full >= 64 ? __builtin_popcountll(ballot64(live && incl <= full)) : 0;
                if (!more) {
                    carry_out = t - tt;
                    wave_lds_sync();                                  // (every lane has read its entry)
                    if (live && c.lane >= tt && c.lane < t) c.W->listL[c.lane - tt] = e;
                }
                t = tt;
                total = tt ? __builtin_amdgcn_readlane(incl, tt - 1) : 0;
            }
            c.carryL = carry_out;
            if (t == 0) break;                                        // (everything carried: the list is done)
#endif
        }
        const bool take = live && c.lane < t;
        if constexpr (last) {
            if (take) {
                // (the two 32-bit halves of the mask one after the other: half the instructions of a 64-bit ctz / clear per child)
                unsigned short *dst = c.W->kid + off;
                const unsigned tag = (unsigned)c.lane << 8;
                unsigned mlo = (unsigned)mk, mhi = (unsigned)(mk >> 32);
                while (mlo) {
                    const unsigned s = (unsigned)__builtin_ctz(mlo);
                    mlo &= mlo - 1;
                    *dst++ = (unsigned short)(s | tag);
                }
                while (mhi) {
                    const unsigned s = (unsigned)__builtin_ctz(mhi) + 32u;
                    mhi &= mhi - 1;
                    *dst++ = (unsigned short)(s | tag);
                }
            }
#ifdef SV_PROF
            const unsigned long long pa = __builtin_amdgcn_s_memtime();
            SV_CYC(c.pt[6] += pa - pi);
            SV_CNT(c.pt[0] += 1);
#endif
            sv_parent<ML, F, NS>(c, take && cnt > 0, code);
            {   // likelihood terms of the round's shared sums: ML - 1 path rows per node, the group tile once
                const unsigned nn = (unsigned)__builtin_popcountll(ballot64(take && cnt > 0));
                c.n_par += nn * (unsigned)(ML - 1) + (nn ? (unsigned)c.G : 0u);
            }
#ifdef SV_PROF
            SV_CNT(c.pt[1] += (unsigned)__builtin_popcountll(ballot64(take && cnt > 0)));
#endif
            wave_lds_sync();
#ifdef SV_PROF
            const unsigned long long pb = __builtin_amdgcn_s_memtime(), dr0 = c.pt[3];
            SV_CYC(c.pt[1] += pb - pa);
#endif
            sv_children<ML, F, NS>(c, total);
            wave_lds_sync();
#ifdef SV_PROF
            SV_CYC(c.pt[2] += (__builtin_amdgcn_s_memtime() - pb) - (c.pt[3] - dr0));
#endif
        } else {
#ifdef SV_PROF
            SV_CNT(c.pt[5] += 1);
#endif
            const unsigned ps = n3_pack(node);
            if (take) {
                uint2 *dst = ((LVL == 0) ? c.W->list0 : (LVL == ML - 2) ? c.W->listL : c.W->list[LVL >= 1 && LVL < ML - 2 ? LVL - 1 : 0]) + off + carry_in;
                unsigned mlo = (unsigned)mk, mhi = (unsigned)(mk >> 32);
                while (mlo) {
                    const unsigned s = (unsigned)__builtin_ctz(mlo);
                    mlo &= mlo - 1;
                    *dst++ = make_uint2(ps, code | (s << 24));
                }
                while (mhi) {
                    const unsigned s = (unsigned)__builtin_ctz(mhi) + 32u;
                    mhi &= mhi - 1;
                    *dst++ = make_uint2(ps, code | (s << 24));
                }
            }
            wave_lds_sync();
            {
                const int fp = c.final_path;
                c.final_path = fp && pos + t >= n_in;
                sv_expand<ML, LVL + 1, F, NS>(c, total + carry_in);
                c.final_path = fp;
            }
            wave_lds_sync();
        }
        if (c.remaining == 0) return;
        pos += t + carry_out;
    }
}

// ---- a whole PREFIX finished by one bound (search mode; option "n3_prefix_bound") ----------------------------------------
// For a candidate with the prefix's rows fixed, let every LEAF interval l fit perfectly: replace its term q_l = w.c_l by a free
// t_l > 0.  NLL = K0 - sum' R_i ln q_i - sum_l R_l ln t_l + Rtot ln(z'.w + sum_l N_l t_l)  (sum' = the prefix's group terms, z' =
// the prefix's column sums) is minimised over the t_l in closed form -- t_l = R_l B / (Rtot N_l), B = z'.w Rtot / R', R' = sum' R_i --
// and what is left is the likelihood of the PREFIX ALONE plus a constant of the problem:
//      min over the leaf rows  >=  min_w [K0 - sum' R_i ln q_i + R' ln(z''.w)] + R' ln(1 - Nrem) + R' ln(Rtot / R') - sum_l R_l ln(R_l / (Rtot N_l))
// (z'' = z' / (1 - Nrem), Nrem = sum_l N_l).  The bracket is an 18-to-13-term problem of the kind the kernel solves all the time:
// a few Newton steps from the chain point, lane g on group term g, sums over the wave, and its self-concordance lower bound
// (exact FP64 logarithms here: once per prefix).  A prefix whose bound lies beyond the window of the running minimum holds no
// finalist, no suspect (a rejected candidate's fallback value is above its minimum) and -- column sums positive -- no
// degenerate candidate: its ~13 000 leaves are counted as dismissed and the wave moves on.  One evaluation decides for a prefix
// that cannot be pruned (its value at the chain point is already within the window): ~0.5 % of such a prefix's work.
__device__ __forceinline__ double sv_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
template <class F>
__device__ __noinline__ bool sv_prefix_beyond(const Sv4<F> *fXY, const typename SvWt<F>::T *fRR, int G, double s1n, double s2n, double Rmin_pre,
                                              const double *r_leaf, const double *rN_leaf, int nleaf, double inv_N, double Rtot, double K0, double thr,
                                              double w0c, double u1c, double u2c, double *pt) {
    const int lane = threadIdx.x & 63;
    double Rrem = 0.0, Nrem = 0.0, Csum = 0.0;
    for (int l = 0; l < nleaf; l++) {
        const double Rl = r_leaf[l], Nl = rN_leaf[l] * inv_N;
        Rrem += Rl;
        Nrem += Nl;
        if (Rl > 0.0) Csum += Rl * log(Rl / (Rtot * Nl));
    }
    const double Rp = Rtot - Rrem, om = 1.0 - Nrem;
    if (!(Rp > 0.0) || !(om > 0.0) || G > WAVE) return false;
    const double Cconst = Rp * log(om) + Rp * log(Rtot / Rp) - Csum;
    const double s1 = s1n / om, s2 = s2n / om;
    // lane g: group term g of the tile (pairs {x0, x1, y0, y1}, weights {R0, R1, ...})
    const bool active = lane < G;
    double x = 0.0, y = 0.0, R = 0.0;
    if (active) {
        const F *xy = (const F *)&fXY[lane >> 1];
        const F *rr = (const F *)&fRR[lane >> 1];
        x = (double)xy[lane & 1];
        y = (double)xy[2 + (lane & 1)];
        R = (double)rr[lane & 1];
    }
    const double a = x - s1, b = y - s2;
    double u1 = (1.0 / 3.0) / s1, u2 = (1.0 / 3.0) / s2;
    bool from_chain = false;
    // start: where the previous prefix's bound ended (neighbouring prefixes differ in their last rows: one or two steps do), else
    // the sieve's chain point, else the prefix's simplex centre
    if (pt[0] == pt[0]) {
        w0c = pt[0];
        u1c = pt[1];
        u2c = pt[2];
    }
    if (w0c == w0c) {
        const double zw = w0c + s1 * u1c + s2 * u2c;
        if (zw > 0.0 && zw < 1e300) {
            u1 = u1c / zw;
            u2 = u2c / zw;
            from_chain = true;
        }
    }
    for (int it = 0; it < 12; it++) {
        const double q = 1.0 + a * u1 + b * u2;
        if (ballot64(active && !(q > 0.0))) {
            if (from_chain) {                       // (the chain point is outside this prefix's domain: from its simplex centre)
                u1 = (1.0 / 3.0) / s1;
                u2 = (1.0 / 3.0) / s2;
                from_chain = false;
            } else {
                u1 *= 0.5;
                u2 *= 0.5;
            }
            continue;
        }
        const double qq = active ? q : 1.0, w = 1.0 / qq, t = R * w, tw = t * w;
        const double lq = log(qq);
        const double val = sv_wave_sum_f64(R * lq);
        // (the value at ANY point is above the prefix's minimum: within the window already -> no bound can finish the prefix)
        if (!(K0 - val + Cconst > thr)) return false;
        // F = float: the tile holds the group weights rounded to single precision (relative 2^-24): the value is that of a problem
        // whose weights are off by as much -- at most 1.3e-7 sum R |ln q| away
        double wmargin = 0.0;
        if constexpr (sizeof(F) == 4) wmargin = 1.3e-7 * sv_wave_sum_f64(R * fabs(lq));
        const double g1 = sv_wave_sum_f64(t * a), g2 = sv_wave_sum_f64(t * b);
        const double h11 = sv_wave_sum_f64(tw * a * a), h12 = sv_wave_sum_f64(tw * a * b), h22 = sv_wave_sum_f64(tw * b * b);
        const double hh = h11 * h22, det = hh - h12 * h12;
        if (!(det > N3_COND_MIN * hh)) return false;
        const double d1 = (h22 * g1 - h12 * g2) / det, d2 = (h11 * g2 - h12 * g1) / det;
        const double lam2 = g1 * d1 + g2 * d2;
        if (!(lam2 == lam2) || !(fabs(d1) + fabs(d2) < 1e30)) return false;
        const double tt = sqrt(fmax(lam2, 0.0) / Rmin_pre);
        if (lane == 0) {                                 // (a point of the prefix's domain: the next prefix starts from it)
            pt[0] = 1.0 - s1 * u1 - s2 * u2;
            pt[1] = u1;
            pt[2] = u2;
        }
        if (tt < 0.25) {
            // min >= value - (lambda^2 / 2)(1 + t + 2 t^2)   (self-concordance, t = lambda / sqrt(Rmin) < 1/2; 5 % on top like sv_beyond)
            const double lb = K0 - val - 0.525 * lam2 * (1.0 + tt + 2.0 * tt * tt) + Cconst;
            if (lb - wmargin - 1e-3 - 1e-12 * fabs(K0) > thr) return true;
            if (lam2 < 1e-3) return false;          // (converged: the prefix's minimum is within the window)
        }
        const double step = tt > 0.25 ? 1.0 / (1.0 + tt) : 1.0;
        u1 += step * d1;
        u2 += step * d2;
    }
    return false;
}

// The next task of the launch, for the whole wave.  Out of line on purpose: with the fetch inlined into the kernel's task loop
// (hipcc 7.2, -O3) the wave never left the loop -- one task per wave through the same counter, or a grid-stride loop without
// the counter, both ran; bisected on the GPU, profiles/r4/NOTES.md.
__device__ __noinline__ int sv_next_task(unsigned *ctr) {
    int t = 0;
    if ((threadIdx.x & 63) == 0) t = (int)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(t);
}

#ifdef SV_WITNESS
#define n3_sieve_kernel n3_sieve_witness_kernel
#endif
__host__ __device__ __forceinline__ bool sv_second_mode(const N3Dev &P) { return P.no_dismiss && P.conv_l2 < 1e-6 && !P.no_second; }
// SEC = false: the instantiation of the launches that take no second evaluation in place (known on the host: sv_second_mode) -- the
// tight modes' code (lean shared evaluation, third-order sums, cubic correction) folds away instead of riding along behind
// wave-uniform branches (the coarse FP64 leg: 2 % with it; the float instantiations -- the shipped search -- likewise).
template <int ML, class F, int NS, bool SEC>
__global__ __launch_bounds__(64 * SV_WAVES, sizeof(F) == 4 ? SV_OCC : SV_OCC64) void n3_sieve_kernel(N3Dev Pg, SearchArgs A, const N3Task *tasks, const unsigned *stbuf,
                                                                          int ntasks, SvSurvivor *surv, unsigned surv_cap,
                                                                          unsigned *surv_count) {
    __shared__ SvLds<ML, F, NS> S;
    const int m = Pg.m, D = m - ML, Q = Pg.Q;
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        S.lb[i] = Pg.lb[i];
        S.ub[i] = Pg.ub[i];
    }
    for (int i = threadIdx.x; i < SV_RIDX_W * SV_RIDX_W; i += blockDim.x)
        S.ridx[i] = Pg.ridx[(i / SV_RIDX_W - SV_RIDX_W / 2 + N3_MAX_COPY) * N3_RIDX_W + (i % SV_RIDX_W - SV_RIDX_W / 2 + N3_MAX_COPY)];
    for (int i = threadIdx.x; i < Q; i += blockDim.x) {
        const unsigned rw = Pg.rowtab[i];
        S.rowtab[i] = (unsigned char)rw;
        S.row16[i] = (unsigned short)((rw & 15u) | ((rw >> 4) << 8));
        if constexpr (SvTab<F, NS>::v) S.rowF[i] = Sv2<F>{(F)(rw & 15u), (F)(rw >> 4)};
    }
    for (int i = threadIdx.x; i < ML * N3_MAX_Q; i += blockDim.x) (&S.smask[0][0])[i] = Pg.smask[(size_t)D * N3_MAX_Q + i];
    __syncthreads();
    N3Dev P = Pg;
    P.lb = S.lb;
    P.ub = S.ub;
    P.rowtab = S.rowtab;               // (P.ridx stays the full table in HBM: the prefix successor reads it once per prefix)

    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // PERSISTENT WAVES: a wave takes the next task of the launch whenever it has finished one (one counter per launch, behind the
    // launch's contender counter).  With one task per wave and four waves per block, a block held its registers and LDS until
    // its slowest task was through, and the last blocks of a launch ran on a nearly empty chip: 10 % of the wave slots' time.
    unsigned *const task_next = surv_count + SV_TASKCTR_OFF;
    if (lane == 0) S.w[wv].pb_pt[0] = __builtin_nan("");       // (the prefix bound's point lives across the wave's tasks: w fits the data's ratios, wherever in the space)
    for (;;) {
    const int task = sv_next_task(task_next);
    if (task >= ntasks) break;                         // whole wave leaves together; no block barrier below
    wave_lds_sync();                                   // (the wave's LDS areas are rewritten for the new task)
    const N3Task tk = tasks[task];
    unsigned st[NS];                                   // lane l: the packed prefix nodes of depths l, 64 + l, ...
#pragma unroll
    for (int j = 0; j < NS; j++) st[j] = WAVE * j + lane < D ? stbuf[(size_t)task * N3_STB + WAVE * j + lane] : 0u;

    SvCtx<ML, F, NS> c;
    c.S = &S;
    c.W = &S.w[wv];
    c.dynmask = Pg.dynmask;
    c.cnt = Pg.cnt;
    c.Q = Pg.Q;
    c.swm = Pg.swmask;
    c.NT1 = Pg.NT + 1;
    c.lane = lane;
    c.D = D;
    c.m = m;
    c.A = A;
    c.surv = surv;
    c.surv_cap = surv_cap;
    c.surv_count = surv_count;
    c.base = ((u128)tk.base_hi << 64) | tk.base_lo;
    c.remaining = (unsigned)tk.count;
    c.skip = tk.skip;
    c.done = 0;
    c.K0 = Pg.K0;
    c.screen_margin = 2e-5 * Pg.Rtot + 1.0;            // f32 sums: |error| <= Rtot (|ln q| 2^-23 + 2^-22) stays far below this
    c.rtot_f = (F)Pg.Rtot;
    c.inv_Rtot = (F)(1.0 / Pg.Rtot);
    c.inv_tau = (F)(1.0 / (double)Pg.tau);
    c.mu_c = F(0);
    c.A_mu_tol = Pg.mu_tol;
    c.conv_l2 = (F)Pg.conv_l2;
    c.fine_l2 = sizeof(F) == 8 ? (F)fmin(Pg.conv_l2, 1e-8) : c.conv_l2;
    c.no_dismiss = Pg.no_dismiss;
    c.second = SEC;                                    // (the launch's mode itself: n3_launch_sieve instantiates both)
    // a tolerance only a third evaluation meets (the tight leg, 1e-12): that one in place too where most lanes of a trip need it
    // (measured: tight leg 128.5 -> 121.0 ms per 2^31; at the certified tolerance, where one lane in ten needs a third, +2 %: not taken)
    // (round 6: also under n3_mu_tol -- the limit on mu sends one lane in four to a third evaluation, not one in ten: 96.2 -> 95.0 ms)
    c.third_min = (Pg.conv_l2 < 1e-10 || Pg.mu_tol > 0.0) ? SV_THIRD_MIN : 65;
#ifdef SV_THIRD_FORCE      // (A/B build: the third evaluation in place from this many lanes on, whatever the tolerance)
    c.third_min = SV_THIRD_FORCE;
#endif
    c.wn0 = F(__builtin_nanf(""));                   // (no chain point yet: the simplex centre)
    c.wn1 = c.wn2 = F(0);
    c.qcount = 0;
    c.n_par = c.n_prefix = 0;
    c.n_child = c.n_dit = 0;
#ifdef SV_WITNESS
    c.wrel = (unsigned long long)(c.base - (((u128)A.wit_begin_hi << 64) | A.wit_begin_lo));
    c.tau = (double)Pg.tau;
#endif
#ifdef SV_PROF
    for (int i = 0; i < 7; i++) c.pt[i] = 0;
    const unsigned long long pw0 = __builtin_amdgcn_s_memtime();
#endif
    const double inv_N = 1.0 / Pg.N;
    double leafR[ML];
#pragma unroll
    for (int l = 0; l < ML; l++) {
        leafR[l] = Pg.r[D + l];                        // (uniform address: scalar loads)
        c.leafRf[l] = (F)leafR[l];
        c.leafRho[l] = (F)sqrt(leafR[l]);
        c.leafN[l] = (F)(Pg.rN[D + l] * inv_N);
    }
    if (lane == 0) {
#pragma unroll
        for (int l = 0; l < ML; l += 2) c.W->fRL[l >> 1] = SvWt<F>::make((F)leafR[l], (F)leafR[l + 1], sqrt(leafR[l]), sqrt(leafR[l + 1]));
    }
    unsigned long long n_terms = 0, n_pterms = 0, n_pruned = 0;

    // Rank-deficient candidates (rows on one line, n3_core.hpp: N3Line) are the host's to list: testing every child here --
    // even behind a wave-uniform flag, in a second instantiation of the expansion or out of line -- cost 1-5 % of the kernel
    // through the registers and the layout of its hot loops.  A wave only notes whether one of its prefixes is collinear (a few
    // scalar instructions in the group-tile loop); api.hip materialises such tasks and scans them (list_deficient).
    // (The note lives in LDS: one more scalar held across the expansion cost 2 % of the kernel -- it runs out of scalar registers.)
    if (lane == 0) c.W->task_line = 0u;
    // the device-wide running minimum: only the finish kernel lowers it, between sieve launches -- one load per task
    sv_set_threshold<ML, F, NS>(c, order_unbits(load_agent_u64(&A.ctr->best_bits)) + A.window);
    while (c.remaining > 0) {
        // ---- group tile of the prefix: intervals with the same row collapse into one likelihood term {a, b, sum r}
#ifdef SV_PROF
        const unsigned long long pg0 = __builtin_amdgcn_s_memtime();
#endif
        int G = 0;
        double S1p = 0.0, S2p = 0.0, Rmin = __builtin_inf();
        {
            // One pass instead of one round of wave reductions per distinct row (13 rounds of six dependent shuffles: 18 000 cycles per
            // prefix, half of what a prefix finished by its bound costs): every lane ADDS its intervals' counts to the bin of their row
            // in LDS (ds_add_f64; integer-valued doubles below 2^53: exact in any order), then lane l looks at the bins l, 64 + l, ...
            // and writes the tile entries of the rows that occur -- in row-code order.  The bins live in the record planes, which
            // are free between prefixes.
            bool inp[NS];
            unsigned myrow[NS];                                          // a | b << 4
#pragma unroll
            for (int j = 0; j < NS; j++) {
                inp[j] = WAVE * j + lane < D;
                myrow[j] = st[j] >> 24;
                if (inp[j]) c.W->pre[WAVE * j + lane] = (unsigned short)((myrow[j] & 15u) | ((myrow[j] >> 4) << 8));
            }
            {   // Do the prefix rows (x_i, y_i) lie on one line?  (Rare.)  Then some children may be rank-deficient (n3_core.hpp:
                // N3Line): the task is noted for the host.  Lane-parallel: the first row, the first row that differs, one cross
                // product per lane.
                const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)myrow[0], 0);
                unsigned p1 = p0;
                bool differs = false;
#pragma unroll
                for (int j = NS - 1; j >= 0; j--) {                      // (the first row that differs: lowest j, lowest lane)
                    const unsigned long long df = ballot64(inp[j] && myrow[j] != p0);
                    if (df) {
                        p1 = (unsigned)__builtin_amdgcn_readlane((int)myrow[j], __builtin_ctzll(df));
                        differs = true;
                    }
                }
                bool on_line = true;
                if (differs) {
                    const int a0 = (int)(p0 & 15u), b0 = (int)(p0 >> 4), da = (int)(p1 & 15u) - a0, db = (int)(p1 >> 4) - b0;
                    unsigned long long off = 0ull;
#pragma unroll
                    for (int j = 0; j < NS; j++) {
                        const int cr = da * ((int)(myrow[j] >> 4) - b0) - db * ((int)(myrow[j] & 15u) - a0);
                        off |= ballot64(inp[j] && cr != 0);
                    }
                    on_line = !off;
                }
                if (on_line && lane == 0) c.W->task_line = 1u;
            }
            double *binR = (double *)&c.W->par, *binN = binR + 256;       // 2 x 256 doubles = 4 KB (the float planes' size)
#pragma unroll
            for (int k = 0; k < 4; k++) ((double2 *)binR)[lane + WAVE * k] = make_double2(0.0, 0.0);
            wave_lds_sync();
#pragma unroll
            for (int j = 0; j < NS; j++)
                if (inp[j]) {
                    atomicAdd(&binR[myrow[j]], Pg.r[WAVE * j + lane]);
                    atomicAdd(&binN[myrow[j]], Pg.rN[WAVE * j + lane]);
                }
            wave_lds_sync();
            double s1l = 0.0, s2l = 0.0, rml = __builtin_inf();
            for (int k = 0; k < 4; k++) {
                const unsigned code = (unsigned)(lane + WAVE * k);
                const double Ns = binN[code], Rs = binR[code];
                const bool has = Ns > 0.0;                                // (normal counts are >= 1: a row that occurs has Ns > 0)
                const unsigned long long hm = ballot64(has);
                if (!hm) continue;
                if (has) {
                    const int idx = G + mbcnt(hm);
                    const F a = (F)(code & 15u), b = (F)(code >> 4);
                    F *xy = (F *)&c.W->fXY[idx >> 1];
                    F *rr = (F *)&c.W->fRR[idx >> 1];
                    xy[idx & 1] = a;
                    xy[2 + (idx & 1)] = b;
                    rr[idx & 1] = (F)Rs;
                    if constexpr (sizeof(F) == 8) rr[2 + (idx & 1)] = sqrt(Rs);
                    s1l += (double)(code & 15u) * Ns;
                    s2l += (double)(code >> 4) * Ns;
                    if (Rs > 0.0) rml = fmin(rml, Rs);
                }
                G += __builtin_popcountll(hm);
            }
            wave_lds_sync();
            if ((G & 1) && lane == 0) {          // an odd last term is paired with a copy of itself of weight 0
                F *xy = (F *)&c.W->fXY[G >> 1];
                F *rr = (F *)&c.W->fRR[G >> 1];
                xy[1] = xy[0];
                xy[3] = xy[2];
                rr[1] = F(0);
                if constexpr (sizeof(F) == 8) rr[3] = F(0);
            }
            S1p = sv_wave_sum_f64(s1l);
            S2p = sv_wave_sum_f64(s2l);
            Rmin = wave_min(rml);
        }
        const double Rmin_pre = Rmin < __builtin_inf() ? Rmin : 1.0;      // ... of the prefix's group terms alone (sv_prefix_beyond)
#pragma unroll
        for (int l = 0; l < ML; l++)
            if (leafR[l] > 0.0) Rmin = fmin(Rmin, leafR[l]);
        if (!(Rmin < __builtin_inf())) Rmin = 1.0;
        c.G = G;
        c.GP = (G + 1) >> 1;
        c.S1p = (F)(S1p * inv_N);
        c.S2p = (F)(S2p * inv_N);
        c.rtot_over_rmin = (F)(Pg.Rtot / Rmin);
        c.sqrt_ror = (F)sqrt(Pg.Rtot / Rmin);
        {
            // (sv_mu_limit's constant: 0.9 / 1.01 x (1 - t)^3 (1 - t+) for the LARGEST t = lambda / sqrt(Rmin) an evaluation that passes
            // "n3_conv_l2" can have on this prefix -- 0.004 on the bench's data, where the worst case t = 0.1 gave 0.718: the limit is
            // 1.38 times larger and binds less often)
            const double tm = fmin(sqrt(Pg.conv_l2 * Pg.Rtot / Rmin), 0.1), tp = (tm / (1.0 - tm)) * (tm / (1.0 - tm));
            const double fchain = (1.0 - tm) * (1.0 - tm) * (1.0 - tm) * (1.0 - tp);
            c.mu_c = (Pg.mu_tol > 0.0 && c.no_dismiss) ? (F)((0.9 / 1.01) * fchain * Pg.mu_tol * sqrt(Rmin) / Pg.Rtot) : F(0);
            c.mu_t2 = (F)(tm * tm * Rmin / Pg.Rtot);                 // ... and the decrement that t corresponds to
        }
        wave_lds_sync();
        const unsigned it0 = c.n_dit, par0 = c.n_par;
        c.par = n3_unpack(n3_lane_state<NS>(st, D - 1));
        c.n_prefix++;
        bool pruned = false;
        if (Pg.prefix_bound && !c.no_dismiss && c.thr < 1e300 && S1p > 0.0 && S2p > 0.0 &&
            sv_prefix_beyond<F>(c.W->fXY, c.W->fRR, G, S1p * inv_N, S2p * inv_N, Rmin_pre, Pg.r + D, Pg.rN + D, ML, inv_N, Pg.Rtot, c.K0, c.thr,
                                (double)c.wn0, (double)c.wn1, (double)c.wn2, c.W->pb_pt)) {
            // the leaves below the prefix, from the counting table; the task's share of them is done
            const N3State &pn = c.par;
            const u128 tv = c.cnt[((((size_t)(D - 1) * c.Q + pn.slot) * 2 + pn.sw) * c.NT1 + pn.lo) * c.NT1 + (pn.hi - 1)];
            const unsigned long long sz = (tv >> 64) ? ~0ull : (unsigned long long)tv;
            if (sz > c.skip) {
                const unsigned long long avail = sz - c.skip;
                const unsigned here = avail < (unsigned long long)c.remaining ? (unsigned)avail : c.remaining;
                c.done += here;
                c.remaining -= here;
                n_pruned += here;
                pruned = true;
            }
        }
#ifdef SV_PROF
        SV_CYC(c.pt[0] += __builtin_amdgcn_s_memtime() - pg0);
#endif
        if (!pruned) {
            c.carryL = 0;
            c.final_path = 1;
            sv_expand<ML, 0, F, NS>(c, 1);
            if (c.qcount) sv_drain<ML, F, NS, true>(c);           // the tile changes with the prefix: the queue is emptied first
        }
        n_terms += (unsigned long long)(c.n_dit - it0) * (unsigned)(G + ML);          // full evaluations: every term of the candidate
        n_pterms += (unsigned long long)(c.n_par - par0);                              // shared sums of the rounds: path rows per node + the group tile per round
        c.skip = 0;                                    // only the first prefix of a task starts mid-way
        if (c.remaining == 0) break;
#ifdef SV_PROF
        const unsigned long long pn0 = __builtin_amdgcn_s_memtime();
        const bool more = n3_next_prefix<NS>(P, st, D, lane);
        SV_CYC(c.pt[4] += __builtin_amdgcn_s_memtime() - pn0);
        if (!more) break;
#else
        if (!n3_next_prefix<NS>(P, st, D, lane)) break;
#endif
        wave_lds_sync();                               // the prefix rows in LDS are rewritten next
    }
    if (lane == 0 && c.W->task_line) {
        const unsigned idx = atomicAdd(&A.ctr->line_count, 1u);
        if (idx < A.line_cap) {
            A.line[2 * idx] = (unsigned long long)c.base;
            A.line[2 * idx + 1] = (unsigned long long)(c.base >> 64);
        }
    }
    if (lane == 0) {
        SearchCounters *sc = stat_slot(A);       // (one of 64 copies: a quarter of a million waves adding to one cache line serialise)
        atomicAdd(&sc->evaluated, (unsigned long long)c.done);
        atomicAdd(&sc->iterations, (unsigned long long)c.n_child + c.n_dit);
        atomicAdd(&sc->terms, n_terms);
        atomicAdd(&sc->sieve_pterms, n_pterms);
        atomicAdd(&sc->sieve_children, (unsigned long long)c.n_child);
        if (n_pruned) atomicAdd(&sc->sieve_pruned, n_pruned);
#ifdef SV_PROF
        SV_CYC(c.pt[5] = __builtin_amdgcn_s_memtime() - pw0);
        for (int i = 0; i < 7; i++) atomicAdd(&sc->prof[i], c.pt[i]);
#else
        atomicAdd(&sc->prof[0], (unsigned long long)c.n_par);
#endif
        atomicAdd(&sc->prof[7], (unsigned long long)c.n_prefix);
    }
    }
}

#ifndef SV_WITNESS      // (the witness object holds the sieve kernel and its launcher only)
// ------------------------------------------------------------------------------------------------------------------
// finish: one lane per contender
// ------------------------------------------------------------------------------------------------------------------
__device__ __noinline__ int sv_reference_outcome(int m, double tau, const double *r, const double *rN, const unsigned char *rows, double nu[3]) {
    N3RefSystem sys;
    sys.m = m;
    sys.tau = tau;
    sys.r = r;
    sys.rN = rN;
    sys.c = rows;
    sys.init();
    return n3_ref_outcome(sys, nu);      // 1 own iterate (nu), 2 the nu = 1/3 fallback, 0 None
}

extern __shared__ unsigned short fin_rows[];     // [256][m | 1]: the block's contender rows, a | b << 8 (dynamic LDS)
__global__ __launch_bounds__(256) void n3_finish_kernel(N3Dev P, SearchArgs A, const SvSurvivor *surv, unsigned surv_cap,
                                                        const unsigned *surv_count, unsigned *accepted_count) {
    __shared__ double rr[N3_MAX_M_WIDE], rn[N3_MAX_M_WIDE];
    __shared__ double2 ltab[128];                    // smx_log's table: m exact logarithms per contender at ~15 instructions each
    const int m = P.m;
    unsigned n = *surv_count;
    if (n > surv_cap) n = surv_cap;
    if (blockIdx.x * blockDim.x >= n) return;        // (block-uniform: nothing to do for this block)
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        rr[i] = P.r[i];
        rn[i] = P.rN[i];
    }
    smx_log_stage(ltab);
    // The rows of the block's 256 contenders go to LDS once, loaded by consecutive threads from consecutive addresses (a record's
    // 2 m bytes are contiguous).  Round 2 read them from global memory in every pass over the intervals -- a dozen passes of
    // byte loads 272 bytes apart between lanes: with millions of contenders (a step that runs into a region better than the
    // minimum it was given) the kernel took 4 ns per contender, ten times what its arithmetic needs.
    const int stride = m | 1;                        // (odd: the lanes' rows start on different banks)
    {
        const unsigned first = blockIdx.x * blockDim.x;
        const unsigned nrec = n - first < 256u ? n - first : 256u;
        const unsigned total = nrec * (unsigned)m;
        const unsigned magic = (unsigned)((0x100000000ull + (unsigned)m - 1) / (unsigned)m);
        for (unsigned d = threadIdx.x; d < total; d += 256u) {
            const unsigned rec = __umulhi(d, magic), i = d - rec * (unsigned)m;          // d / m (exact: d (m - 1) < 2^32)
            fin_rows[rec * (unsigned)stride + i] = ((const unsigned short *)surv[first + rec].rows)[i];
        }
    }
    __syncthreads();
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const SvSurvivor *sv = surv + idx;
    const unsigned char *rows = sv->rows;            // [m][2] bytes {a, b} (global: for the reference's procedure below)
    const unsigned short *myrows = fin_rows + threadIdx.x * stride;
    const u128 rank = ((u128)sv->rank_hi << 64) | sv->rank_lo;
    const double tau = (double)P.tau, inv_Rtot = 1.0 / P.Rtot;
    double S1 = 0.0, S2 = 0.0, Rmin = __builtin_inf();
    for (int i = 0; i < m; i++) {
        const unsigned rw = myrows[i];
        S1 = __builtin_fma((double)(rw & 0xffu), rn[i], S1);
        S2 = __builtin_fma((double)(rw >> 8), rn[i], S2);
        if (rr[i] > 0.0) Rmin = fmin(Rmin, rr[i]);
    }
    if (!(Rmin < __builtin_inf())) Rmin = 1.0;
    if (S1 == 0.0 || S2 == 0.0) return;              // (the sieve lists degenerate candidates elsewhere)
    const double s1 = S1 / P.N, s2 = S2 / P.N;
    auto terms = [&](auto &&body) {
        for (int i = 0; i < m; i++) {
            const unsigned rw = myrows[i];
            body((double)(rw & 0xffu), (double)(rw >> 8), rr[i]);
        }
    };
    // Newton from the simplex centre (interior for every candidate), to lambda^2 / Rtot < 1e-12
    // Newton from the iterate the sieve left the contender at (converged to its tolerance there: one or two steps to 1e-12 instead of
    // six to eight from the simplex centre -- a step that runs into a region better than its hint lists millions of contenders);
    // the centre where there is none, or where it is not a point of the domain
    N3Newton T;
    T.u1 = T.p1 = (1.0 / 3.0) / s1;
    T.u2 = T.p2 = (1.0 / 3.0) / s2;
    if (sv->u1 == sv->u1 && sv->u2 == sv->u2 && fabs(sv->u1) + fabs(sv->u2) < 1e30) {
        bool inside = true;
        for (int i = 0; i < m; i++) {
            const unsigned rw = myrows[i];
            inside = inside && (rr[i] <= 0.0 || __builtin_fma((double)(rw & 0xffu) - s1, sv->u1, __builtin_fma((double)(rw >> 8) - s2, sv->u2, 1.0)) > 0.0);
        }
        if (inside) {
            T.u1 = T.p1 = sv->u1;
            T.u2 = T.p2 = sv->u2;
        }
    }
    T.iters = 0;
    T.status = 0;
    T.singular = false;
    while (T.status == 0) n3_newton_step(terms, s1, s2, inv_Rtot, T, 1e-12);
    atomicAdd(&A.ctr->finish_iterations, (unsigned long long)T.iters);
    double best = order_unbits(load_agent_u64(&A.ctr->best_bits));
    double u1 = T.status == 1 ? T.u1 : T.p1, u2 = T.status == 1 ? T.u2 : T.p2;
    bool conv = T.status == 1, accept = false;
    if (conv) {
        const double n1 = s1 * u1, n2 = s2 * u2, n0 = 1.0 - n1 - n2;
        accept = (n0 >= 0.0 && n0 <= 1.0 && n1 >= 0.0 && n1 <= 1.0 && n2 >= 0.0 && n2 <= 1.0);
    }
    double h11 = 0.0, h12 = 0.0, h22 = 0.0, acc = 0.0, g1 = 0.0, g2 = 0.0;
    bool outside = false;
    terms([&](double x, double y, double R) {
        const double a = x - s1, b = y - s2;
        const double q = __builtin_fma(a, u1, __builtin_fma(b, u2, 1.0));
        outside |= !(q > 0.0);
        const double t = R / q, tw = t / q;
        g1 = __builtin_fma(t, a, g1);
        g2 = __builtin_fma(t, b, g2);
        h11 = __builtin_fma(tw * a, a, h11);
        h12 = __builtin_fma(tw * a, b, h12);
        h22 = __builtin_fma(tw * b, b, h22);
    });
    if (conv && !accept && T.singular) {             // rank-deficient: the minimiser is a line; intersect it with the simplex
        N3Hess T2;
        T2.u1 = u1; T2.u2 = u2; T2.h11 = h11; T2.h12 = h12; T2.h22 = h22;
        accept = n3_admissible(T2, s1, s2);
        u1 = T2.u1;
        u2 = T2.u2;
    }
    terms([&](double x, double y, double R) {
        const double q = __builtin_fma(x - s1, u1, __builtin_fma(y - s2, u2, 1.0));
        acc = __builtin_fma(R, smx_log(q, ltab), acc);
    });
    double nll = P.K0 - acc;
    if (accept) {
        atomicAdd(&A.ctr->accepted, 1ull);
        atomicAdd(accepted_count, 1u);               // (per slice: a slice redone by the fused kernel takes that kernel's count)
        if (!(nll <= best + A.window)) return;       // its own minimum is beyond the window: whatever the reference reports is too
        // The minimum lies in the simplex -- but does the reference find it?  (n3.hip, n3_cold_path: same decision.)
        double nu[3];
        const int outcome = sv_reference_outcome(m, tau, rr, rn, rows, nu);
        if (outcome == 0) return;                    // the reference returns None for it: neither a finalist nor a suspect
        u1 = nu[1] / s1;
        u2 = nu[2] / s2;
        acc = 0.0;
        terms([&](double x, double y, double R) {
            const double q = __builtin_fma(x - s1, u1, __builtin_fma(y - s2, u2, 1.0));
            acc = __builtin_fma(R, smx_log(q, ltab), acc);
        });
        nll = P.K0 - acc;
        best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
        if (nll <= best + A.window) {
            const double u0 = (1.0 - s1 * u1 - s2 * u2) / tau, usum = u0 + u1 + u2;    // closed form of M3 (Optimizer.py:318-330)
            tie_append(A.ctr, A.list, A.list_cap, rank, nll, u0 / usum, u1 / usum, u2 / usum);
            if (nll < best) atomicMin(&A.ctr->best_bits, order_bits(nll));
        }
        return;
    }
    // Rejected: a LOWER BOUND of anything the reference could report for it inside the simplex.  Converged outside the
    // simplex: the unconstrained minimum plus the self-concordance gain Rmin w(d / sqrt(Rmin)), w(t) = t - ln(1 + t), d = the
    // Hessian-norm distance from the optimum to the simplex.  Not converged: Frank-Wolfe from the last feasible iterate.
    double lbnd;
    if (conv) {
        const double vx[3] = {0.0, 1.0 / s1, 0.0}, vy[3] = {0.0, 0.0, 1.0 / s2};
        double d2 = __builtin_inf();
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int f = (e + 1) % 3;
            const double ex = vx[f] - vx[e], ey = vy[f] - vy[e];
            const double px = u1 - vx[e], py = u2 - vy[e];
            const double hex = h11 * ex + h12 * ey, hey = h12 * ex + h22 * ey;
            double t = (px * hex + py * hey) / (ex * hex + ey * hey);
            t = fmin(fmax(t, 0.0), 1.0);
            const double rx = px - t * ex, ry = py - t * ey;
            d2 = fmin(d2, rx * (h11 * rx + h12 * ry) + ry * (h12 * rx + h22 * ry));
        }
        const double tt = sqrt(fmax(d2, 0.0) / Rmin);
        const double gain = 0.98 * Rmin * (tt - log1p(tt));
        lbnd = nll + (gain == gain ? gain : 0.0);
    } else {
        const double e0 = g1 * u1 + g2 * u2, e1 = e0 - g1 / s1, e2 = e0 - g2 / s2;
        const double fw = fmin(e0, fmin(e1, e2));
        lbnd = (outside || !(fw == fw) || !(nll == nll)) ? -__builtin_inf() : nll + fw;
    }
    if (!(lbnd == lbnd)) lbnd = -__builtin_inf();
    best = fmin(best, order_unbits(load_agent_u64(&A.ctr->best_bits)));
    if (lbnd <= best + A.window) suspect_append(A.ctr, A.sus, A.sus_cap, rank, lbnd, nll);
    if (lbnd < order_unbits(load_agent_u64(&A.ctr->rej_bits))) {
        const unsigned long long old = atomicMin(&A.ctr->rej_bits, order_bits(lbnd));
        if (old > order_bits(lbnd)) {  // we hold the minimum (racy pair, diagnostic only)
            A.ctr->rej_rank_lo = (unsigned long long)rank;
            A.ctr->rej_rank_hi = (unsigned long long)(rank >> 64);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------
// P.L must be the burst depth (n3_sieve_levels) -- for the task kernel as well (tasks are cut at depth m - P.L).
int n3_sieve_levels(const N3Dev &P) {
    if (P.m < 8) return 0;                              // small searches stay on the fused kernel
    // six expanded rows: a prefix then holds thousands of leaves, so that what a wave does once per prefix (group tile, a chain
    // of dependent table reads down the first levels) is amortised; four for short matrices
    int want = P.m >= 10 ? 6 : 4;
    if (const char *e = getenv("THETA_SIEVE_LEVELS")) {
        const int v = atoi(e);
        if (v == 4 || v == 6) want = v;
    }
    return want;
}

#endif   // !SV_WITNESS

#ifdef SV_WITNESS
#define n3_launch_sieve n3_launch_sieve_witness
#endif
void n3_launch_sieve(const N3Dev &P, const SearchArgs &A, const N3Task *tasks, const unsigned *stbuf, int ntasks, SvSurvivor *surv,
                     unsigned surv_cap, unsigned *surv_count, hipStream_t st) {
#ifndef SV_WITNESS
    if (A.wit) {           // a witnessed call (theta_search_witness): the same source compiled with -DSV_WITNESS
        n3_launch_sieve_witness(P, A, tasks, stbuf, ntasks, surv, surv_cap, surv_count, st);
        return;
    }
#endif
    // as many blocks as the chip holds at once (or as there are tasks for): the waves fetch their tasks themselves
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    const int need = (ntasks + SV_WAVES - 1) / SV_WAVES, resident = n_cu * (P.force64 ? SV_OCC64 : SV_OCC);
    dim3 grid(need < resident ? need : resident), block(64 * SV_WAVES);
    // (NS = prefix intervals per lane: 2 up to 128 intervals -- the instantiation everything is tuned for --, 4 up to 256: BASELINE
    // config 5's shape, m = 200; wider prefix tables in LDS, two blocks per CU)
#define SV_LAUNCH(MLV, FT, NSV, SECV) hipLaunchKernelGGL((n3_sieve_kernel<MLV, FT, NSV, SECV>), grid, block, 0, st, P, A, tasks, stbuf, ntasks, surv, surv_cap, surv_count)
#define SV_LAUNCH2(MLV, FT, NSV) do { if (sec) SV_LAUNCH(MLV, FT, NSV, true); else SV_LAUNCH(MLV, FT, NSV, false); } while (0)
    const bool wide = P.m - P.L > 2 * WAVE, sec = sv_second_mode(P);
    if (P.force64) {       // FP64 throughout (n3_force_f64): the same kernel on doubles
        if (P.L <= 4) { if (wide) SV_LAUNCH2(4, double, 4); else SV_LAUNCH2(4, double, 2); }
        else { if (wide) SV_LAUNCH2(6, double, 4); else SV_LAUNCH2(6, double, 2); }
    } else {
        if (P.L <= 4) { if (wide) SV_LAUNCH2(4, float, 4); else SV_LAUNCH2(4, float, 2); }
        else { if (wide) SV_LAUNCH2(6, float, 4); else SV_LAUNCH2(6, float, 2); }
    }
#undef SV_LAUNCH2
#undef SV_LAUNCH
}

#ifndef SV_WITNESS
void n3_launch_finish(const N3Dev &P, const SearchArgs &A, const SvSurvivor *surv, unsigned surv_cap, const unsigned *surv_count,
                      unsigned *accepted_count, hipStream_t st) {
    // (a grid for a full list: blocks beyond the count leave at once)
    const size_t lds = (size_t)256 * (size_t)(P.m | 1) * sizeof(unsigned short);
    (void)hipFuncSetAttribute((const void *)n3_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(n3_finish_kernel, dim3((surv_cap + 255) / 256), dim3(256), lds, st, P, A, surv, surv_cap, surv_count, accepted_count);
}
#endif   // !SV_WITNESS
