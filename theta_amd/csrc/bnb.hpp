// Branch and bound above the sieve's prefix (bnb.hip): frontier nodes and the expansion launch.
#pragma once
#include "n3_core.hpp"

// A frontier node: the first d rows of a matrix are fixed (its path: one alphabet slot per row, kept beside the node).
struct BnbNode {
    uint64_t base_lo, base_hi;   // rank of the first matrix below the node (the reference's enumeration order)
    double w0, u1, u2;           // where the node's bound was attained, homogeneous coordinates: its children start there
    double bound;                // lower bound of the NLL of every matrix below the node (-inf: not established)
    unsigned state;              // the DFS node of its last row (n3_pack: slot, symmetry switch, ratio window, row)
    unsigned line;               // are the rows so far on one line? (N3Line packed, bnb_line_pack)
};

// A rank range handed to the search: every matrix below a surviving node of the emit depth.
struct BnbRange {
    uint64_t base_lo, base_hi, count_lo, count_hi;
};

#define BNB_STAT_SLOTS 64
#define BNB_STAT_STRIDE 8        // 64-bit words per slot (a cache line of its own)
// statistics words of a slot: 0 children bounded (a Newton solve each), 1 Newton iterations, 2 children pruned, 3 children kept
// because their rows are collinear (follow_line), 4 children kept without an established bound

struct BnbArgs {
    int d;                       // rows fixed in the parents; the children fix row d
    int emit_depth;              // children with this many rows are emitted as rank ranges ...
    unsigned long long emit_max; // ... and so are children with at most this many matrices below them
    double thr;                  // a child whose bound exceeds it is pruned (+inf: none is)
    int full_bound;              // 1: iterate every child's bound to convergence (beam search: the bounds are sorted), else stop as soon as the decision is made
    int follow_line;             // 1: children whose rows are still collinear are kept whatever their bound (their matrices include the
                                 // rank-deficient ones, which the reference values off their optimum)
    double constc;               // the constant of depth d + 1: R' ln(om) + R' ln(Rtot / R') - sum over the free intervals of r ln(r / (Rtot N))
    double Z0;                   // sum of the normal counts of intervals 0 .. d, / N
    double rd, nd;               // r_d and rN_d / N
    int path_stride;             // bytes per path (a multiple of 4)
    const BnbNode *in;
    const unsigned char *in_path;
    unsigned n_in;
    BnbNode *out;
    unsigned char *out_path;
    unsigned long long out_cap;
    BnbRange *ranges;
    unsigned long long range_cap;
    unsigned long long *counters;   // [0] children written to `out`, [1] ranges written (both may exceed the capacities: the host checks)
    unsigned long long *stats;      // [BNB_STAT_SLOTS][BNB_STAT_STRIDE]
};

void bnb_launch_expand(const N3Dev &P, const BnbArgs &A, hipStream_t st);
void bnb_launch_bounds(const BnbNode *nodes, unsigned long long n, double *out, hipStream_t st);
void bnb_launch_compact(const BnbNode *nodes, const unsigned char *paths, unsigned long long n, int path_stride, double cut, BnbNode *out,
                        unsigned char *out_paths, unsigned long long out_cap, unsigned long long *counter, hipStream_t st);

// ---- branch and bound over the mixture space (bnb.hip, second half) ----------------------------------------------------------
// A box of the search.  line == 0: a box of mixtures v = s (mu0, mu1, mu2) >= 0 over the WHOLE alphabet (c.v = tau v0 + a v1 + b v2).
// line == l + 1: a box of (alpha, beta) -- lo/hi[0], lo/hi[1]; the third side has no width -- over the rows of line l of the
// alphabet alone (c.v = alpha + t beta for the row x0 + t dx, y0 + t dy): the matrices whose rows all lie on that line are
// rank deficient, and the reference may report them at a mixture with NEGATIVE entries (Optimizer.py:148-165, 318-330: hybrj on a
// singular Jacobian, M3's mu never range-checked); their objective depends on v through (alpha, beta) only, both of either sign.
struct MixCell {
    double lo[3], hi[3];
    double lb;                     // the box's bound (set when it is kept)
    unsigned key;                  // the path from its root: one bit per cut (the first 32) -- what a sharded search partitions by
    unsigned short depth;          // cuts so far
    unsigned short line;           // see above
};
static_assert(sizeof(MixCell) == 64, "MixCell is copied as four 16-byte words");

// a line of the alphabet's grid: the points (x0 + t dx, y0 + t dy), t = 0 .. T-1
struct MixLine {
    signed char x0, y0, dx, dy;
    int T;
};

// counters of a search (64-bit words on the device; the host reads them once per BATCH of iterations, not per level)
enum {
    MIX_TOP0 = 0, MIX_TOP1,        // boxes on the stack, by iteration parity
    MIX_WK0, MIX_WK1,              // children kept by the iteration, by parity
    MIX_LEAVES,                    // leaves in the list (reset when the list is walked)
    MIX_LISTED, MIX_CUT,           // matrices written by the leaf walks / walks that went over their share
    MIX_TESTED,                    // children bounded so far
    MIX_MAXTOP, MIX_OVERFLOW,      // deepest stack; bit 0: the stack overflowed, bit 1: leaves were dropped (proposal passes may)
    MIX_ITERS,                     // iterations that had work
    MIX_MINB, MIX_MINB_LINE,       // smallest bound among the leaves (order-preserving image of the double, mix_ord), whole alphabet / lines
    MIX_LEAVES_ALL, MIX_LEAVES_LINE,   // leaves found in total / those of lines
    MIX_NCTR = 16
};
__host__ __device__ inline unsigned long long mix_ord(double x) {
    unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__host__ __device__ inline double mix_unord(unsigned long long u) {
    u = (u >> 63) ? (u & 0x7fffffffffffffffull) : ~u;
    return __builtin_bit_cast(double, u);
}

// constants of an interval: rN, r, r / rN, ln rN, and phi's minimum r - r ln r (phi_i(t) = rN t - r ln(rN t) at rN t = r; 0 for r = 0)
struct MixIv {
    double N, r, ts, lnN, phimin;
};
struct MixArgs {
    int m, Q, tau;
    const double *r, *rN;          // [m]
    const MixIv *iv;               // [m] the intervals' constants (bnb.hip: MixIv)
    const unsigned char *rowtab;   // [Q] slot -> a | b << 4
    const unsigned char *slot_of;  // [256] a | b << 4 -> slot, 0xff: not a row of the alphabet
    const unsigned char *lb, *ub;  // [m] order-adjusted bounds
    double cst;                    // -Rtot + Rtot ln Rtot
    double thr;
    double leaf[3];                // a box is a leaf when its widths are at most these
    double leaf_line[3];           // ... a box of a line
    const MixLine *lines;
    int n_lines;
    int shard_g, shard_G;          // G > 1: of the boxes that have just become as small as shard_mult leaves a side this rank keeps those whose position hashes to g
    double shard_mult;
    unsigned chunk;                // boxes taken off the stack per iteration
    double dive_blend;             // ... what a dive ranks by: bound + dive_blend x (relaxed objective at the centre - bound)
    int dive_niches;               // ... within groups of like mixtures first (mix_select_kernel)
    int dive;                      // 1: no threshold, boxes are RANKED (mix_select_kernel) by the relaxed objective at their centre, kept in `lb`
};

void mix_launch_iteration(const MixArgs &A, MixCell *stack, unsigned long long stack_cap, MixCell *work, MixCell *leaves, unsigned long long leaf_cap,
                          unsigned long long *ctr, int parity, int drop_leaves, unsigned long long n_max, unsigned beam, hipStream_t st);
#define MIX_BEAM_LIMIT 1024
void mix_launch_list(const MixArgs &A, const MixCell *leaves, unsigned long long n_leaves, unsigned char *out, unsigned long long out_cap,
                     unsigned long long per_thread_cap, unsigned long long max_steps, unsigned long long *ctr, unsigned *seen_tab, unsigned long long seen_mask,
                     hipStream_t st);
