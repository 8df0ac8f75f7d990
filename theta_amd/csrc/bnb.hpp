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
struct MixCell {
    double lo[3], hi[3];
    double lb;                     // the box's bound (set when it is kept)
};

struct MixArgs {
    int m, Q, tau;
    const double *r, *rN;          // [m]
    const unsigned char *rowtab;   // [Q] slot -> a | b << 4
    const unsigned char *lb, *ub;  // [m] order-adjusted bounds
    double cst;                    // -Rtot + Rtot ln Rtot
    double thr;
    double leaf[3];                // a box is a leaf when its widths are at most these
};

void mix_launch_split(const MixArgs &A, const MixCell *in, unsigned long long n_in, MixCell *out, unsigned long long out_cap, MixCell *leaves,
                      unsigned long long leaf_cap, unsigned long long *counters, hipStream_t st);
void mix_launch_list(const MixArgs &A, const MixCell *leaves, unsigned long long n_leaves, unsigned char *out, unsigned long long out_cap, int per_thread_cap,
                     unsigned long long *counters, hipStream_t st);
