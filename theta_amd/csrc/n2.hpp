// n = 2 search: device-side data and the host-side table builder.
#pragma once
#include "common.hpp"

#define N2_KVS 16  // stride of the count table (values 0..15)

struct N2Dev {
    int m, kv, tau;              // kv: values are 0..kv-1 (kv = max(ub)+1)
    double max_normal, N, Rtot, K0;
    const double *PR, *PN;       // [m+1] exact prefix sums of r and rN
    const unsigned long long *P; // [m][N2_KVS] cumulative prefix counts
    const unsigned char *lb, *ub; // [m] order-adjusted bounds
    const short *lbpos;          // [N2_KVS+1] first index whose lb >= v (m if none)
    unsigned long long total;    // number of candidates
    int first_zero_r;            // smallest interval index with r_i == 0 (m if none)
    int quick;                   // search: dismiss a candidate by a rigorous lower bound of its optimum after one evaluation at the
                                 // thread's chain point (n2_quick); 0 = solve every candidate (option "n2_no_dismiss")
};

struct N2Host {
    int m = 0, kv = 0;
    std::vector<int> lb, ub;
    std::vector<unsigned long long> P;
    unsigned long long total = 0;
};

// Applies Enumerator._check_bound_order (Enumerator.py:90-113) and builds the cumulative count
// table (the cumulative form of TimeEstimate.count_number_matrices_2, TimeEstimate.py:91-111).
int n2_build_host(int m, const int32_t *lb, const int32_t *ub, N2Host &h);

