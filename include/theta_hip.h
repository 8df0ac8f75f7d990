/*
 * theta_hip.h -- C ABI of libtheta_hip.so: the MI355X (gfx950) implementation of THetA's
 * combinatorial likelihood search.
 *
 * Every entry point replaces one operator of the reference's hot path (citations are
 * file:line into the reference's python/ directory).  Plain pointers and sizes only; the caller
 * owns every buffer; the library keeps no pointer after a call returns.  All functions return
 * THETA_OK (0) or a positive error code; theta_last_error() gives the message of the last failure
 * on the calling thread.
 *
 * Conventions
 *   n        number of populations, 2 or 3 (column 0 of C is the normal genome, == tau)
 *   m        number of intervals of the search (rows of C), 2 <= m <= THETA_MAX_M (n=3: <= 256 on the sieve path -- search, FP64 mode,
 *            materialised generator, theta_solve_batch and theta_search_values alike; the FUSED kernel's own per-candidate dump
 *            holds 64, beyond that theta_search_values reports the reference's outcome per candidate)
 *   r, rN    tumour / normal read counts AFTER the reference's sort_r (DataTools.py:95-118),
 *            int64, rN[i] > 0
 *   lb, ub   per-interval copy-number bounds as given to Enumerator(...) (Enumerator.py:39);
 *            the library applies _check_bound_order (Enumerator.py:90-113) itself
 *   rank     position of a candidate in the reference's enumeration order
 *            (Enumerator.generate_next_C, Enumerator.py:74-87), 0-based, as 128 bits
 *            {lo, hi} little-endian pair of uint64
 *   C (u8)   a candidate is stored without its constant column: m*(n-1) bytes, row-major
 *            [interval][tumour column]
 */
#ifndef THETA_HIP_H
#define THETA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define THETA_MAX_M 256          /* intervals in one search                                   */
#define THETA_MAX_COPY 15        /* largest copy number an entry of C may take                */

enum {
    THETA_OK = 0,
    THETA_ERR_ARG = 1,           /* bad argument (shape, range, null pointer)                 */
    THETA_ERR_NO_CANDIDATES = 2, /* bounds admit no matrix (RunTHetA.py:217-219 exits here)   */
    THETA_ERR_HIP = 3,           /* HIP runtime failure (no device, launch error, OOM)        */
    THETA_ERR_OVERFLOW = 4,      /* n=2 count beyond 64 bits; a rank range no search finishes   */
    THETA_ERR_CAPACITY = 5       /* output capacity too small; *n_out holds the needed size   */
};

typedef struct theta_ctx theta_ctx;         /* one per process, bound to one GPU              */
typedef struct theta_problem theta_problem; /* one search instance resident in HBM            */

/* ---- context ------------------------------------------------------------------------------ */
int theta_create(int device_id, theta_ctx **out);
/* number of GPUs this process can see (0 and THETA_ERR_HIP when there is none): what `do_optimization(..., max_processes)`
 * (RunTHetA.py:124-171) shards a search over -- one worker process per GPU instead of the reference's forked CPU workers */
int theta_device_count(int *n_out);
void theta_destroy(theta_ctx *ctx);
const char *theta_last_error(void);
/* name[cap] receives the device name; cu = compute units; hbm_bytes = total device memory.    */
int theta_device_info(theta_ctx *ctx, char *name, int cap, int *cu, uint64_t *hbm_bytes);
/* Platform check of the n=3 parity claim for RANK-DEFICIENT candidates: what the reference reports for those hangs on the last
 * bit of numpy's `x ** 2` (Optimizer.py:303-311), i.e. of libm's pow(x, 2.0) under its interpreter; the kernels restate glibc >= 2.28
 * on x86-64 with FMA (csrc/refpow.hpp).  mismatches = how many of n seeded arguments THIS host's pow(x, 2.0) squares differently
 * (0 on the platform the parity was established on; no GPU needed). */
int theta_refpow_check(int n, int *mismatches);
/* hipDeviceSynchronize on the context's GPU (every entry point below already returns with its work finished). */
int theta_synchronize(theta_ctx *ctx);

/* ---- search instance ---------------------------------------------------------------------- */
/*
 * Replaces the construction `Enumerator(n,m,k,tau,lb,ub,multi_event)` + `Optimizer(r,rN,m,n,tau,
 * upper_bound=max_normal)` of RunTHetA.py:134-135 / 181-182.  Uploads r, rN and the bounds, and
 * builds the rank<->candidate counting tables in HBM (exact: the n=2 table is the cumulative form
 * of TimeEstimate.count_number_matrices_2, TimeEstimate.py:91-111; the n=3 table counts the
 * matrices Enumerator._generate_next_C_3 really yields, Enumerator.py:172-214).
 * max_normal is only enforced for n=2, like the reference (Optimizer.py:107-110).
 * n=3: copy numbers (bounds) up to THETA_MAX_COPY.  Ranks exist as long as at most 64 distinct valid rows (a, b) lie within the
 * bounds of some interval (always so up to 7; beyond, the reference's own bounds heuristic -- ub = max(k, y + 1),
 * DataTools.py:64-66 -- produces narrow windows that fit).  With more rows (full bounds [0, 9]: 72) the problem is MIX-ONLY:
 * created all the same, theta_problem_count reports 2^128 - 1, every entry point that takes ranks (theta_search*, theta_enumerate,
 * theta_bnb) answers THETA_ERR_ARG with a message, and theta_mix_search searches the space whole.  A space of 2^128 matrices or more is accepted: its count saturates at
 * 2^128 - 1 (theta_problem_count: "that many or more") and rank ranges below that are searched like any other.
 */
int theta_problem_create(theta_ctx *ctx, int n, int m, int tau, const int64_t *r, const int64_t *rN,
                         const int32_t *lb, const int32_t *ub, double max_normal,
                         theta_problem **out);
void theta_problem_destroy(theta_problem *p);
/* log2 of a LOWER BOUND of the number of matrices of an n=3 space (host arithmetic, no GPU, a millisecond): the matrices whose rows
 * keep a + b from decreasing -- all of them matrices Enumerator._generate_next_C_3 yields (Enumerator.py:172-264: the ratio window
 * of such a matrix always holds the ratio 1); exponentially many in m.  theta_problem_create uses it to tell a space of 2^128 matrices or more WITHOUT its counting table (m = 200, k = 7:
 * 2 GB, 0.4 s), which then waits for the first call that takes ranks; -inf when the bound is 0. */
int theta_count_lower_bound(int m, int tau, const int32_t *lb, const int32_t *ub, double *log2_count);
/* Number of candidates generate_next_C() would yield (excludes the Q1 duplicate first matrix). */
int theta_problem_count(theta_problem *p, uint64_t count[2]);

/* Search statistics, all counters are per call. */
typedef struct theta_search_stats {
    uint64_t evaluated;      /* candidates enumerated and solved                                */
    uint64_t accepted;       /* candidates with an admissible optimum (Optimizer.solve != None);
                                n=3 fused search: among the candidates that were not `dismissed`   */
    uint64_t degenerate;     /* rank-deficient candidates: listed, not solved (theta_search_degenerate) */
    uint64_t iterations;     /* solver iterations summed over candidates                        */
    uint64_t terms;          /* likelihood terms (interval groups) summed over candidates       */
    uint64_t list_overflow;  /* records dropped because the device tie list was full            */
    uint64_t flops;          /* FP64 operations executed by the solver (counted in-kernel)      */
    uint64_t flops_f32;      /* FP32 operations of the n=3 packed coarse pass and screen        */
    uint64_t dismissed;      /* n=3: candidates finished after one evaluation because a rigorous lower
                                bound of their optimum lies beyond the window of the running minimum */
    double best_nll;         /* smallest accepted NLL seen by the kernel (fused arithmetic)     */
    double rejected_bound;   /* smallest lower bound on the NLL of any REJECTED candidate       */
    uint64_t rejected_rank[2];
    double kernel_ms;        /* duration of the search kernel, HIP events on its stream         */
    double setup_ms;         /* duration of the unranking / task set-up kernels                 */
    uint64_t phase_cycles[8];/* shader cycles per kernel phase summed over waves (diagnostic):
                                0 group tile, 1 leaf scan, 2 solver iterations, 3 values+tracking,
                                4 prefix successor, 5 whole wave                               */
    uint64_t survivors;      /* n=3 fast path: contenders the sieve kernel handed to the finish kernel */
    uint64_t fallback_candidates; /* n=3 fast path: candidates of slices redone by the fused kernel (contender list full) */
    uint64_t redo_flops;     /* ... FP64 operations the fused kernel executed on those slices -- NOT part of `flops`:       */
    uint64_t redo_flops_f32; /* ... every candidate is counted once in evaluated / dismissed / iterations / terms / flops,  */
    double redo_kernel_ms;   /* ... and kernel_ms is the sieve + finish kernels' only; the redo's time is here              */
    uint64_t kernel_launches;/* launches of the search kernel behind kernel_ms (n=3 sieve: one per slice of the range)      */
    uint64_t pruned;         /* n=3 sieve: candidates of prefixes finished by the prefix bound ("n3_prefix_bound"): counted in
                                evaluated and dismissed, no evaluation of their own in iterations / flops                   */
} theta_search_stats;

/*
 * Fused enumerate + solve + arg-min over the candidates with rank in [rank_begin, rank_end).
 * Replaces the loop of do_optimization_single (RunTHetA.py:191-208): Enumerator.generate_next_C
 * (Enumerator.py:74-87), Optimizer.solve (Optimizer.py:68-88) and the running minimum.
 *
 * Returns, in increasing rank order, every accepted candidate whose NLL is within `window` of the
 * smallest NLL found in the range (the host replays the reference's sequential isClose rule,
 * Misc.py:36-47, on this list).  nll/mu come from the fused arithmetic (group-aggregated sums);
 * use theta_solve_batch on the returned C for values in the reference's own summation order.
 *
 * One call takes at most 2^31 candidates at n=3 (2^40 at n=2; THETA_ERR_ARG beyond): a longer range is walked in
 * pieces by the caller, each piece started from the minimum found so far (theta_problem_hint) and the pieces' lists merged
 * within `window` of the overall minimum -- theta_amd.Problem.search does exactly that, one piece at a time, and refuses
 * ranges of more than 2^25 pieces (THETA_ERR_OVERFLOW): a 70-interval space with bounds [0, 2] holds 2.5e34 matrices.
 *
 *   cap        capacity (records) of the output arrays
 *   nll[cap], mu[cap*n], rank[cap*2], C[cap*m*(n-1)]
 *   n_out      number of records written (or needed, with THETA_ERR_CAPACITY)
 *   stats      may be NULL
 */
int theta_search(theta_problem *p, const uint64_t rank_begin[2], const uint64_t rank_end[2],
                 double window, int cap, double *nll, double *mu, uint64_t *rank, uint8_t *C,
                 int *n_out, theta_search_stats *stats);

/*
 * theta_search over SEVERAL rank ranges in one pass of the kernels: ranges[nranges * 4] = {begin lo, begin hi, count lo, count hi} in
 * rank order, disjoint (what theta_bnb returns), at most 2^31 candidates together.  Same outputs, same side lists
 * (theta_search_suspects / _degenerate), same hint and options as theta_search -- the candidates of the ranges are searched as if
 * they were one range with the gaps cut out.  n = 3 on the sieve path (m >= 8).  A slice of the call whose contender list
 * overflows ends the call with THETA_ERR_CAPACITY and *n_out = 0: search the ranges one by one then (theta_search has a ladder).
 */
int theta_search_ranges(theta_problem *p, int nranges, const uint64_t *ranges, double window, int cap, double *nll, double *mu,
                        uint64_t *rank, uint8_t *C, int *n_out, theta_search_stats *stats);

/*
 * "Suspects" of the last theta_search call on this problem (n=3): candidates the search REJECTED (likelihood
 * optimum outside the simplex) whose lower bound lies within `window` of the minimum.  The reference does report
 * such a candidate -- at nu = (1/3,1/3,1/3), where its BFGS fallback stalls (Optimizer.py:150-160, 255-265) -- so
 * feed C to theta_solve_batch (ok = 2 entries carry that value) and let the ones within the window join the
 * finalists; theta_boundary_min bounds what an off-path scipy run could report instead.  rank[cap*2],
 * lbound[cap] (lower bound of the NLL), C[cap*m*(n-1)].  n_out = number available; the device list holds 2^20, call with cap = -1
 * to learn how many more were dropped (a range whose own minimum is poor can have millions: pass a hint, below).
 */
int theta_search_suspects(theta_problem *p, int cap, uint64_t *rank, double *lbound, uint8_t *C, int *n_out);

/*
 * n=3 RANK-DEFICIENT candidates of the last theta_search call, in rank order: matrices whose columns (tau, x, y) are linearly
 * dependent, i.e. whose rows (x_i, y_i) lie on one line -- two equal tumour columns, x + y = const, a constant column, an
 * ALL-ZERO tumour column.  The search kernels do not solve them: what the reference reports for such a matrix is not its
 * optimum.  The bordered Jacobian of Optimizer.py:288-301 is exactly singular, MINPACK's hybrj may stop unconverged at a nu
 * inside [0,1]^3 whose components do not sum to one, _solve_n3plus takes it (Optimizer.py:150-153), M3 makes a mu with a
 * negative entry of it and L3 reports NaN (or a finite value below the matrix's true minimum); with an all-zero column the
 * arithmetic is NaN from normalize_C on (Optimizer.py:167-174), fsolve returns its start, M3's second fsolve call lands on a
 * unit vector plus rounding residue and L3 turns that into a finite NLL or NaN.  The reference's driver takes whatever comes
 * out -- a NaN likelihood is "close" to anything (Misc.py:44-46) and is appended to `best` wherever it stands.  Feed C to
 * theta_solve_batch, which reproduces the reference's outcome for each, and replay ALL of them with the finalists
 * (theta_amd/search.py: degenerate_records).  rank[cap*2], C[cap*m*2]; cap = -1 queries how many did not fit the device
 * list (2^20 per call: search a shorter range then, as theta_amd.Problem.search does).
 */
int theta_search_degenerate(theta_problem *p, int cap, uint64_t *rank, uint8_t *C, int *n_out);

/*
 * Run-time switches of a search instance (no reference counterpart).  name / value:
 *   "n3_no_dismiss"  1: no candidate is finished by the lower bound of its optimum after one evaluation -- every one is
 *                    iterated to the coarse tolerance and valued ("passed through the full solve")
 *   "n3_force_f64"   1: every evaluation in FP64 -- the sieve kernel's double instantiation (default: packed FP32 evaluations,
 *                    FP64 for contenders); with "n3_no_dismiss" the full solve at the reference's precision
 *   "n3_conv_l2"     coarse-pass threshold on the squared Newton decrement / sum r (default 1e-4).  A candidate is left ONE full
 *                    Newton step beyond the evaluation that finds it; with t = lambda / sqrt(Rmin) <= 0.1 that step ends at a
 *                    decrement <= 1.53 (sum r / Rmin) l2^2 (self-concordance), so a tolerance T on the point a candidate is left at
 *                    is met by the value sqrt(T Rmin / (1.53 sum r)) (bench.py: certified_conv_l2)
 *   "n3_mu_tol"      > 0 (with "n3_no_dismiss"): the tolerance as a CERTIFICATE ON MU.  An evaluation at u (decrement lambda, tangent Hessian H,
 *                    t = lambda / sqrt(Rmin) <= 0.1) counts as converged only if, besides "n3_conv_l2", the point one full Newton step further
 *                    is certified within this distance of the candidate's optimum in every component of mu: self-concordance bounds the
 *                    step's decrement and the drift of H, H's smaller eigenvalue (through its lower bound det / tr) turns the Hessian norm into a distance in u, and
 *                    d mu / d u is bounded by (1.5 + max(1, |mu1| + |mu2|) |k|) / U (tests/test_certified_tolerance_cpu.py restates the chain and
 *                    checks it on 3 000 random problems; a 10 % margin on top).  Points away from the simplex (a nu_j < -0.05) -- where the
 *                    reference reports no mixture of the candidate's own -- are left to "n3_conv_l2" alone.  0 (default): the decrement alone decides
 *   "n3_warm_blend"  weight of the previous optimum in a chunk's first warm start
 *   "n3_sieve"       1 (default): the two-kernel path (sieve + finish, n3_sieve.hip) where it applies; 0: the fused
 *                    kernel of n3.hip throughout (also used for theta_search_values and m < 8; m <= 64)
 *   "n3_contender_cap"  contenders a slice of the sieve may list before it counts as overflowed and is redone (0 = the
 *                    list's real capacity, 2^24); small values let tests walk the redo ladder
 *   "n3_nan_sweep"   1: after the search every candidate of the range also goes through the reference's own per-candidate procedure
 *                    (as theta_solve_batch runs it, device resident, 2e8-5e8 candidates/s), and the ones the reference reports with a
 *                    NaN likelihood -- about one full-rank matrix in a million, decided by where MINPACK's iteration stops --, and the ones
 *                    it reports at or below the search's minimum + window, join
 *                    the list of theta_search_degenerate.  0 (default).  theta_amd.search sets it for spaces up to 2^33 matrices
 *   "n3_auto_f64"    1 (default): a problem on which the packed-FP32 screen lists more than 0.5 % of a call's candidates as contenders
 *                    (its margin, 2e-5 sum r + 1, is coarse against the spread of the NLL within the range: 200 intervals of which a
 *                    range varies the last dozen) runs the double instantiation from the next call on; 0: never (same finalists)
 *   "n3_prefix_bound" 1 (default): the n=3 search finishes a whole prefix (the first m - 6 rows: ~13 000 candidates at m = 50, K = 6) when
 *                    the lower bound of its relaxed problem -- every leaf interval fitted perfectly: the likelihood of the prefix alone plus
 *                    a constant -- lies beyond the window of the running minimum; 0: every candidate gets its own evaluation (same
 *                    finalists, suspects and degenerate lists).  Never applies under "n3_no_dismiss"
 *   "n3_second"      1 (default): in the tight full-solve modes ("n3_no_dismiss" with "n3_conv_l2" < 1e-6), where every candidate needs a second
 *                    evaluation, it is taken in place right after the shared one, by the lane that holds the child, instead of through
 *                    the queue; 0: through the queue (same lists; an A/B switch).  With 1 the shared first step of these modes is also taken
 *                    in single precision with a cubic correction (it only shapes the start of the private FP64 evaluations) and the last
 *                    level's rounds come in whole trips of 64 children: 2.07 instead of 2.49 evaluations per candidate on the bench's data
 *   "n2_no_dismiss"  1: the n=2 search solves every candidate; 0 (default): a candidate whose rigorous lower bound -- one evaluation
 *                    at a chain point, self-concordance -- lies beyond the window of the running minimum is done (same finalists)
 *   "n3_per_task"    candidates per wave task (0 = automatic), "n2_per_thread" candidates per thread (0 = automatic)
 *   "mix_shard_world", "mix_shard_rank"   theta_mix_search on G ranks: rank g keeps the boxes dealt to it where they become as small as 8 leaves a side
 *                    (set the world first); default 1 / 0: every box
 *   "mix_beam"       boxes per level of a THETA_MIX_DIVE (8 .. 1024, default 512)
 *   "mix_max_steps"  steps the walk over the intervals of one (leaf, corner) may take before theta_mix_search gives up (default 2^22)
 *   "mix_max_ms"     theta_mix_search gives up (THETA_ERR_CAPACITY) once it has spent this much wall time with work left (0, the default: never)
 *   "mix_max_boxes"  theta_mix_search gives up (THETA_ERR_CAPACITY) once this many boxes have been bounded and more are waiting
 *                    (0, the default: never -- a flat likelihood is walked to the end, however long it takes)
 * The THETA_N3_* environment variables of the same names only set the defaults at theta_problem_create.
 */
int theta_problem_set_option(theta_problem *p, const char *name, double value);

/*
 * One-shot hint for the next theta_search on this problem: an NLL some candidate is already known to reach (e.g. the
 * minimum of a previously searched rank range or shard).  Starting the running minimum there keeps the tie list and
 * the suspect list short; it never changes the result as long as the hint is attainable.
 */
int theta_problem_hint(theta_problem *p, double nll_upper_bound);

/*
 * Exact minimum of the n=3 NLL over the BOUNDARY of the simplex (some nu_j = 0) for B materialised candidates:
 * the smallest value the reference could report for a candidate whose optimum lies outside the simplex.
 */
int theta_boundary_min(theta_ctx *ctx, int m, int tau, const int64_t *r, const int64_t *rN, int B, const uint8_t *C,
                       double *bound);

/*
 * Per-candidate dump of the fused kernel over ranks rank_begin .. rank_begin+count-1: the
 * reference's --GET_VALUES developer aid (RunTHetA.py:210-215, FileIO.py:114).  nll[count], mu[count*n].
 * n=2: NaN where Optimizer.solve returns None.  n=3: the MINIMUM of each candidate's likelihood where it lies in the
 * simplex, else NaN -- a diagnostic of the fused arithmetic; what the reference REPORTS for an n=3 candidate (its own
 * optimum, the nu = 1/3 fallback, or None) comes from theta_solve_batch (the --GET_VALUES file is written from that).
 * n=3 with more than 64 intervals (the fused kernel holds one per lane): the dump IS the reference's report for every candidate --
 * the generator's matrices through theta_solve_batch's kernel, chunk by chunk in HBM; NaN where the reference reports nothing.
 */
int theta_search_values(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, double *nll,
                        double *mu, theta_search_stats *stats);

/*
 * WITNESS of the n=3 search kernel (no reference counterpart): what the sieve kernel leaves a candidate at.  Runs the search of
 * theta_search over [rank_begin, rank_end) under the instance's current options (hint included) with the WITNESS build of the
 * sieve kernel -- the same source compiled with one more macro, every decision made by the same code -- and returns one record
 * for every 2^every_log2-th candidate of the range: record i belongs to rank rank_begin + (i << every_log2).  With
 * "n3_no_dismiss" (the bench's full-solve legs) every regular candidate has a record with status 1, 2, 5 or 6; this is what
 * "passed through the full solve" (Optimizer.solve per candidate, Optimizer.py:128-165; RunTHetA.py:191-208) comes to per
 * candidate, checked against the oracle by tests/test_gpu_round5.py.  The finalists are theta_search's to return; `stats` are
 * the call's counters (equal to theta_search's on the same range and options).  out[cap]; n_out = records written (or needed,
 * with THETA_ERR_CAPACITY).  n = 3 on the sieve path only (m >= 8): THETA_ERR_ARG otherwise.
 */
typedef struct theta_witness {
    double mu[3];          /* mixture at the point the candidate was LEFT at (after its last Newton step; nu -> mu as Optimizer.M3) */
    double nll;            /* NLL the kernel computed at the candidate's LAST EVALUATION (single-precision logarithms)             */
    float l2_last;         /* squared Newton decrement / sum r found by that evaluation                                            */
    float l2_first;        /* ... by the shared first evaluation (NaN: the candidate had no usable shared point)                   */
    uint16_t evaluations;  /* evaluations of value + gradient + Hessian the candidate took, the shared one included              */
    uint16_t status;       /* 0 no record (rank-deficient candidate; prefix finished by its bound; slice redone by the fused kernel),
                              1 converged at the shared evaluation, 2 converged in the queue, 3 / 4 finished by the lower bound
                              (search mode) at the shared evaluation / in the queue, 5 contender (handed to the finish kernel),
                              6 handed to the finish kernel unsolved (ill-conditioned, or 40 evaluations)                          */
    float mu_bound;        /* option "n3_mu_tol": what the certificate bounds the distance IN MU between that point and the candidate's
                              optimum by (0: no certificate asked for, or the point lies outside the simplex)                      */
} theta_witness;
int theta_search_witness(theta_problem *p, const uint64_t rank_begin[2], const uint64_t rank_end[2], double window,
                         int every_log2, uint64_t cap, theta_witness *out, uint64_t *n_out, theta_search_stats *stats);

/*
 * BRANCH AND BOUND above the search's prefix (n = 3): which rank ranges of the WHOLE space can hold a matrix whose NLL optimum is
 * at most `threshold`?  Replaces -- for spaces no linear walk finishes (BASELINE configs 3 and 4: 4e27 / 2.6e38 matrices) -- the
 * enumeration loop of do_optimization_single (RunTHetA.py:173-220) down to the rows the fused search takes over.  The tree
 * Enumerator._generate_next_C_3 walks (Enumerator.py:172-214: same rows, same edges, same ratio windows, sizes from the counting
 * table) is expanded level by level; a node whose relaxed likelihood -- its fixed rows alone, every later interval fitted perfectly
 * -- has a lower bound beyond `threshold` is dropped with everything below it; the survivors of the emit depth (m - 6 rows) come
 * back as rank ranges [begin, begin + count), in rank order, adjacent ones joined: theta_search on them (with `threshold` minus the
 * window as hint) finds every matrix of the space whose optimum is within the window.  `threshold` must be ATTAINABLE + window: an
 * NLL some matrix of the space is known to reach (e.g. the minimum over the ranges of a beam run) plus the collection window.
 *   beam             0: exact (above).  W > 0: a dive instead -- nothing is pruned, every level keeps its W smallest bounds; the
 *                    ranges (at most W) are where to look for an attainable NLL first
 *   follow_collinear 1: nodes whose rows so far lie on one line are kept whatever their bound -- their subtrees hold the
 *                    rank-deficient matrices, which the reference values off their optimum (theta_search_degenerate lists them when the
 *                    ranges are searched).  Feasible for small spaces only: the collinear prefixes of 44 rows are ~1e21 at m = 50
 *   max_nodes        0: no limit; else the walk stops with THETA_ERR_OVERFLOW once that many nodes were expanded (stats are filled)
 *   ranges[cap * 4]  {begin lo, begin hi, count lo, count hi}; n_out = ranges written (or needed, with THETA_ERR_CAPACITY)
 * What it cannot give: matrices the reference reports BELOW their own optimum or with a NaN likelihood (rank-deficient ones
 * without follow_collinear; about one full-rank matrix in a million, Misc.py:44-46) -- they are not in the ranges unless their
 * optimum is.
 */
typedef struct theta_bnb_stats {
    uint64_t nodes_expanded;     /* frontier nodes expanded (one wave each)                                  */
    uint64_t children_bounded;   /* children given a bound (one Newton solve of <= 64 terms each)            */
    uint64_t newton_iterations;  /* ... iterations of those solves                                           */
    uint64_t children_pruned;    /* ... dropped: bound beyond the threshold                                  */
    uint64_t children_collinear; /* children kept because their rows are collinear (follow_collinear)        */
    uint64_t children_unbounded; /* children kept without an established bound (ill-conditioned, not converged) */
    uint64_t ranges_raw, ranges; /* surviving nodes emitted / rank ranges after joining adjacent ones        */
    uint64_t launches, max_frontier, chunk;
    double leaves;               /* matrices in the ranges                                                   */
    double kernel_ms, wall_ms;
    int emit_depth, complete;
    uint64_t frontier[THETA_MAX_M + 1];   /* nodes kept at depth d                                           */
} theta_bnb_stats;
int theta_bnb(theta_problem *p, double threshold, uint64_t beam, int follow_collinear, uint64_t max_nodes, uint64_t cap,
              uint64_t *ranges, uint64_t *n_out, theta_bnb_stats *stats);

/*
 * BRANCH AND BOUND OVER THE MIXTURE SPACE (n = 3): every matrix -- of the rows, bounds and edge rule of Enumerator._generate_next_C_3
 * (Enumerator.py:172-214, 248-298) -- whose NLL can be at most `threshold` for SOME mixture, found without walking the
 * ranks: BASELINE configs 3 and 4 (4e27 / 2.6e38 matrices) in a fraction of a second.  Replaces, for such spaces, the loop of
 * do_optimization_single (RunTHetA.py:173-220).  With v = s mu the reference's objective (Optimizer.py:236-244) is
 * sum_i [rN_i c_i.v - r_i ln(rN_i c_i.v)] + const at its best scale s: separable over the intervals for a fixed v, so a box of
 * mixtures bounds EVERY matrix at once from m x (rows) one-dimensional problems.  A tree of boxes over v >= 0 keeps those within
 * the threshold; the leaves (width leaf_rel, relative to the mean read-depth ratio per copy: 2e-4 is a good value) are walked depth
 * first over the intervals with the threshold as budget.  C[cap * m * 2] receives the matrices ({a, b} per interval) in the
 * reference's enumeration order, without duplicates: a SUPERSET of the matrices within the threshold (the symmetry rule and the
 * ratio window are the caller's to check; theta_solve_batch gives each one's value as the reference reports it).  The threshold
 * must be an attainable NLL + the collection window.
 *
 * mode, a sum of:
 *   THETA_MIX_PROPOSE     no list -- for the `cap` leaves of smallest bound, the matrix that fits the leaf's centre best (per interval
 *                         the row minimising its term): candidates for a better attainable NLL, to be valued by the caller before a
 *                         finer call with a lower threshold.
 *   THETA_MIX_DIVE        (with THETA_MIX_PROPOSE) no threshold: every level of the tree keeps its "mix_beam" boxes of smallest bound
 *                         (option, default 512), down to the leaf size -- a few thousand boxes in all; the proposals of its leaves
 *                         give the attainable NLL the thresholded search starts from.
 *   THETA_MIX_LINES       also the RANK-DEFICIENT matrices the reference can report within the threshold at a mixture with negative
 *                         entries (Optimizer.py:148-165: hybrj on a singular Jacobian; 318-330: M3's mu is never range-checked; L3 is
 *                         finite wherever the products c_i.mu keep one sign).  The rows of such a matrix lie on one line of the
 *                         alphabet's grid, c.v = alpha + t beta along it, and the value is the same separable objective at some
 *                         (alpha, beta) of either sign: one more tree per line, over (alpha, beta), rows restricted to the line.
 *                         stats->min_bound_lines is then a lower bound of everything finite the reference can report for a
 *                         rank-deficient matrix with two distinct rows or more that the list does not hold (+inf: no box of a line
 *                         came within the threshold).  NaN outcomes have no bound.
 *   THETA_MIX_LINES_ONLY  the lines' trees alone.
 * A sharded search (options "mix_shard_world" / "mix_shard_rank" of theta_problem_set_option): the boxes that have just become as small
 * as 8 leaves a side are dealt out by their position, each rank lists the matrices of its own; the union over the ranks is the unsharded list.
 *
 * THETA_ERR_CAPACITY with *n_out > cap: the list holds *n_out matrices, come again with that capacity.  THETA_ERR_CAPACITY with
 * *n_out == 0: the threshold leaves more boxes or matrices than the device holds (option "mix_max_boxes" bounds the work before).
 * What it cannot give: matrices the reference reports with a NaN likelihood (some rank-deficient ones; about one full-rank matrix
 * in a million).
 */
#define THETA_MIX_PROPOSE 1
#define THETA_MIX_LINES 2
#define THETA_MIX_LINES_ONLY 4
#define THETA_MIX_DIVE 8
typedef struct theta_mix_stats {
    uint64_t boxes_tested, levels, max_boxes, leaves, listed, matrices;
    uint64_t lines, line_leaves, syncs;   /* lines searched, leaves of lines, host synchronisations of the walk                    */
    double kernel_ms, wall_ms;
    double min_bound;            /* the smallest bound among the leaves of the whole alphabet (a lower bound of the space's minimum up to the leaf size) */
    double min_bound_lines;      /* ... among the leaves of the lines */
} theta_mix_stats;
int theta_mix_search(theta_problem *p, double threshold, double leaf_rel, int mode, uint64_t cap, uint8_t *C, uint64_t *n_out,
                     theta_mix_stats *stats);

/*
 * Materialised generator: writes candidates rank_begin .. rank_begin+count-1 in the reference's
 * order.  Replaces repeated Enumerator.generate_next_C() (Enumerator.py:74-87, 119-152, 172-214).
 * out[count * m * (n-1)].
 */
int theta_enumerate(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, uint8_t *out);

/*
 * Same candidates, written to DEVICE memory the caller owns (count * m * (n-1) bytes at d_out, a HIP device pointer
 * on the context's GPU), for consumers that stay on the GPU (theta_score_masked, a caller's own kernels).
 * kernel_ms (may be NULL) receives the HIP-event duration of the enumeration kernels.  No reference counterpart
 * beyond generate_next_C itself (Enumerator.py:74-87): this is the generator without the PCIe copy.
 * d_out must be 4-byte aligned (THETA_ERR_ARG otherwise).
 */
int theta_enumerate_device(theta_problem *p, const uint64_t rank_begin[2], uint64_t count, void *d_out, double *kernel_ms);

/*
 * Per-candidate solve in the reference's own arithmetic order (per-interval sums, the brenth
 * iteration for n=2): replaces Optimizer.solve(C) (Optimizer.py:68-165) for a batch of B
 * materialised candidates C[B*m*(n-1)].
 *   ok[B]      1 = solution, 0 = the reference's `None`.  n=3: the reference's own decision procedure -- MINPACK's hybrj
 *              (behind scipy's fsolve, Optimizer.py:148) restated on the Lagrangian system in the reference's operation
 *              order; 1 = its iterate lies in [0,1]^3 and is reported, 2 = it does not, and the candidate is reported at
 *              nu = (1/3,1/3,1/3), where the reference's fmin_bfgs call -- handed an ascent direction,
 *              Optimizer.py:255-265 -- returns its start (Optimizer.py:155-160).  nu -> mu is Optimizer.M3 as the reference
 *              runs it (a second fsolve call, MINPACK hybrd with forward differences, restated).  A candidate whose NLL
 *              comes out NaN (an all-zero tumour column, mostly) keeps ok = 1 / 2: the reference returns that tuple too
 *   mu[B*n], nll[B]
 *   vals[B*m]  the per-interval p* (third element of the reference's tuple); may be NULL
 * n=3: one lane per candidate, or -- B <= 2048 and m <= 512 -- one WAVE per candidate with the terms of an evaluation spread over its lanes
 * and added up in the reference's order: the same bits either way (THETA_SOLVE_NO_WAVE=1: the lane kernel throughout).
 */
int theta_solve_batch(theta_ctx *ctx, int n, int m, int tau, const int64_t *r, const int64_t *rN,
                      double max_normal, int B, const uint8_t *C, uint8_t *ok, double *mu,
                      double *nll, double *vals);

/*
 * Batched CalcAllC.L2 / CalcAllC.L3 (CalcAllC.py:44-76) on literal float matrices: B weighted
 * matrices Cw[B*m*n] (row-major m x n, float64, exactly what the reference passes), mu[B*n]
 * (for n=2 only mu[b*2] is used, like the reference's scalar mu), r[m] float64.
 * Row validity, the in-place scaling of L2 and the NaN behaviour follow the
 * reference; the library never modifies Cw (the host wrapper applies L2's mutation).
 *   nll[B]; vals[B*m] may be NULL; valid[B*m] (1 = row counted, 0 = the reference's 'X') may be NULL
 */
int theta_score_batch(theta_ctx *ctx, int n, int m, int B, const double *Cw, const double *mu,
                      const double *r, double *nll, double *vals, uint8_t *valid);

/* The same with one r per matrix, r[B*m]: matrices that differ in their rows' read counts -- e.g. the (m+1)-row matrices
 * calc_all_c_* builds, one extra interval each (CalcAllC.py:92-328) -- in ONE launch. */
int theta_score_batch_rows(theta_ctx *ctx, int n, int m, int B, const double *Cw, const double *mu,
                           const double *r, double *nll, double *vals, uint8_t *valid);

/*
 * Device memory the caller owns, so that chains of operators stay in HBM (no reference counterpart: the reference
 * allocates a fresh numpy array per candidate): theta_enumerate_device -> theta_solve_batch_device /
 * theta_score_masked_device -> theta_device_copy of the few numbers wanted.  to_device: 1 = host to device, 0 = back.
 */
int theta_device_alloc(theta_ctx *ctx, size_t bytes, void **out);
int theta_device_free(theta_ctx *ctx, void *p);
int theta_device_copy(theta_ctx *ctx, void *dst, const void *src, size_t bytes, int to_device);
/*
 * theta_solve_batch / theta_score_masked on DEVICE-resident candidates and results (d_* are device pointers on the
 * context's GPU: d_C u8[B*m*(n-1)], d_ok u8[B], d_mu f64[B*n], d_nll f64[B] or f64[B*S], d_vals f64[B*m] or NULL);
 * r, rN, w, mask stay host arrays (a few KB).  kernel_ms (may be NULL): HIP-event duration of the kernel.
 */
int theta_solve_batch_device(theta_ctx *ctx, int n, int m, int tau, const int64_t *r, const int64_t *rN, double max_normal,
                             int B, const void *d_C, void *d_ok, void *d_mu, void *d_nll, void *d_vals, double *kernel_ms);
int theta_score_masked_device(theta_ctx *ctx, int n, int m, int tau, int B, int S, const void *d_C, const double *w,
                              const double *r, const void *d_mu, const uint64_t *mask, void *d_nll, double *kernel_ms);

/*
 * Compact scorer for interval-subset resampling: B candidates as bytes C[B*m*(n-1)] with weights
 * w[m] (the normal counts), mu[B*n], and S row masks mask[S*ceil(m/64)] (uint64 words, bit i =
 * interval i takes part; NULL = one all-ones mask).  Computes CalcAllC.L3/L2's NLL for every
 * (candidate, mask) pair with the masked rows' column 0 treated as 0 (CalcAllC.py:70-75).
 * nll[B*S], candidate-major.
 */
int theta_score_masked(theta_ctx *ctx, int n, int m, int tau, int B, int S, const uint8_t *C,
                       const double *w, const double *r, const double *mu, const uint64_t *mask,
                       double *nll, double *kernel_ms);

/* ---- several GPUs: the one communication step of a sharded search --------------------------------------- */
/*
 * Candidates shard by rank range, one process per GPU, no data-path collective.  What replaces the reference's merge of
 * its workers' lists (find_mins, RunTHetA.py:107-122, after the multiprocessing fan-out of RunTHetA.py:124-171) is one
 * exchange at the end: all-reduce(min) of the shard minima + all-gather of the finalists within the window -- over RCCL
 * (xGMI between the GPUs of a node), owned by this library: no torch, no MPI.
 *
 * theta_comm_create: collective over all ranks.  Rank 0 listens on addr:port (TCP), the others connect; rank 0's
 * ncclUniqueId travels over that connection, then every rank joins ncclCommInitRank on its context's GPU (librccl.so is
 * loaded here, on first use).  transport THETA_COMM_HOST keeps the TCP star for the collectives as well (ctx may be
 * NULL): for multi-process tests on machines without GPUs; the entry points and the merge are the same.
 * All buffers below are HOST memory (the payload is a few hundred bytes; the library stages it through HBM for RCCL).
 */
typedef struct theta_comm theta_comm;
enum { THETA_COMM_RCCL = 0, THETA_COMM_HOST = 1 };
int theta_comm_create(theta_ctx *ctx, int rank, int world, const char *addr, int port, int transport, theta_comm **out);
void theta_comm_destroy(theta_comm *c);
/* rccl_version: ncclGetVersion() (0 for the host transport); collectives: number issued so far on this communicator */
int theta_comm_info(theta_comm *c, int *rank, int *world, int *transport, int *rccl_version, uint64_t *collectives);
int theta_comm_barrier(theta_comm *c);
int theta_comm_allreduce_min(theta_comm *c, double *v, int count);   /* in place */
int theta_comm_allreduce_max(theta_comm *c, double *v, int count);
int theta_comm_allreduce_sum(theta_comm *c, double *v, int count);
int theta_comm_allgather(theta_comm *c, const void *send, size_t bytes, void *recv);   /* recv[world * bytes], rank-major */
/*
 * The exchange: `count` finalists of this rank's shard in (nll[count], mu[count*n], rank[count*2], C[count*m*(n-1)],
 * vals[count*m]) -- reference-order values from theta_solve_batch.  Every rank receives, in increasing rank order, the
 * finalists of ALL shards within `window` of the global minimum, plus every record whose NLL is NaN (the reference's
 * isClose counts NaN as close, Misc.py:44-46, so the replay needs them).  n_out = their number (or the capacity needed,
 * with THETA_ERR_CAPACITY); global_min (may be NULL) = the smallest finite NLL over all shards.
 */
int theta_exchange_finalists(theta_comm *c, int n, int m, int count, const double *nll, const double *mu,
                             const uint64_t *rank, const uint8_t *C, const double *vals, double window, int cap,
                             double *o_nll, double *o_mu, uint64_t *o_rank, uint8_t *o_C, double *o_vals, int *n_out,
                             double *global_min);

#ifdef __cplusplus
}
#endif
#endif /* THETA_HIP_H */
