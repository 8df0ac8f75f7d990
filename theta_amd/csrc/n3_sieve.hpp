// n = 3 search, fast path (n3_sieve.hip): what the sieve kernel hands to the finish kernel.
#pragma once
#include "n3_core.hpp"

// A contender: a candidate whose optimum may lie within the window of the running minimum (or one the packed-FP32
// sieve could not handle).  rows = the whole matrix, m x {a, b} bytes.
struct SvSurvivor {
    uint64_t rank_lo, rank_hi;
    double u1, u2;               // the iterate the sieve left it at (the finish kernel's Newton starts there; NaN: from the simplex centre)
    unsigned char rows[2 * N3_MAX_M_WIDE];
};

// The sieve kernel's waves fetch their tasks from a counter of the launch: it sits SV_TASKCTR_OFF words behind the launch's
// contender counter (`surv_count`), zero at launch.
#define SV_TASKCTR_OFF 64

int n3_sieve_levels(const N3Dev &P);
void n3_launch_sieve(const N3Dev &P, const SearchArgs &A, const N3Task *tasks, const unsigned *stbuf, int ntasks, SvSurvivor *surv,
                     unsigned surv_cap, unsigned *surv_count, hipStream_t st);
// the same kernel with the per-candidate records of theta_search_witness (n3_sieve.hip compiled with -DSV_WITNESS; A.wit set)
void n3_launch_sieve_witness(const N3Dev &P, const SearchArgs &A, const N3Task *tasks, const unsigned *stbuf, int ntasks, SvSurvivor *surv,
                             unsigned surv_cap, unsigned *surv_count, hipStream_t st);
void n3_launch_finish(const N3Dev &P, const SearchArgs &A, const SvSurvivor *surv, unsigned surv_cap, const unsigned *surv_count,
                      unsigned *accepted_count, hipStream_t st);
