// What scipy.optimize.fmin_bfgs does with the call Optimizer._solve_n3plus makes when fsolve's root is out of range
// (Optimizer.py:155: fmin_bfgs(L3_hat, [1/3, 1/3], fprime=dL3_hat)) -- as far as it decides the candidate's fate.
// dL3_hat has the sign of MINUS the gradient of L3_hat (Optimizer.py:255-265 against :246-252), so BFGS's first search
// direction pk = -dL3_hat(x0) points uphill.  The first line search (line_search_wolfe1, MINPACK-2 dcsrch) cannot satisfy
// the decrease condition on an increasing function and fails; the second one (scipy's scalar_search_wolfe2 + zoom: the
// textbook strong-Wolfe search of Nocedal & Wright, Numerical Optimization, alg. 3.5/3.6, with scipy's step-doubling,
// 10 + 10 iterations, c1 = 1e-4, c2 = 0.9, first trial min(1, 1.01 * 2 (phi0 - old_phi0) / derphi0) = a step of length
// 1.01) fails as well -- as long as every trial point stays inside the domain of the logarithms.  BFGS then gives up and
// returns its start: the reference's nu = (1/3,1/3,1/3) fallback.  When the first trial point lies OUTSIDE the domain,
// L3_hat is NaN there, every comparison with it is false, and the search accepts a step on the strength of the (finite,
// wrong-signed) derivative alone: BFGS moves to a point with NaN value and stops; the point is out of range and the
// reference returns None.  This file restates that decision sequence (which trial points, which comparisons) on
// N3RefSystem's data; on the reference's m=6, K=3 table it reproduces all 284 None outcomes and all 4 466 fallbacks.
// scipy is a dependency of the reference, not part of its tree; version installed here: 1.15.
#pragma once
#include "n3_refsys.hpp"

#ifdef HYBRJ4_MANAGE_CONTRACT
#pragma clang fp contract(off)
#endif

template <class SYS>
struct N3RefBfgsT {
    const SYS &s;
    double pk0, pk1, phi0, d0;

    HYBRJ4_HD double fhat(double v0, double v1) const { return s.fhat(v0, v1); }       // (L3_hat, dL3_hat: n3_refsys.hpp)
    HYBRJ4_HD void ghat(double v0, double v1, double &g0, double &g1) const { s.ghat(v0, v1, g0, g1); }
    HYBRJ4_HD double phi(double a) const { return fhat(1.0 / 3.0 + a * pk0, 1.0 / 3.0 + a * pk1); }
    HYBRJ4_HD double dphi(double a) const {
        double g0, g1;
        ghat(1.0 / 3.0 + a * pk0, 1.0 / 3.0 + a * pk1, g0, g1);
        return g0 * pk0 + g1 * pk1;
    }
    static HYBRJ4_HD bool finite(double x) { return x == x && fabs(x) < INFINITY; }
    // minimiser of the quadratic / cubic through the given data; false if there is none (division by zero, NaN, ...)
    static HYBRJ4_HD bool quadmin(double a, double fa, double fpa, double b, double fb, double &x) {
        const double db = b - a;
        if (db * db == 0.0) return false;
        const double B = (fb - fa - fpa * db) / (db * db);
        if (2.0 * B == 0.0) return false;
        x = a - fpa / (2.0 * B);
        return finite(x);
    }
    static HYBRJ4_HD bool cubicmin(double a, double fa, double fpa, double b, double fb, double c, double fc, double &x) {
        const double C = fpa, db = b - a, dc = c - a;
        const double denom = ((db * dc) * (db * dc)) * (db - dc);
        if (denom == 0.0) return false;
        const double v0 = fb - fa - C * db, v1 = fc - fa - C * dc;
        double A = (dc * dc) * v0 + (-(db * db)) * v1;
        double B = (-(dc * dc * dc)) * v0 + (db * db * db) * v1;
        A /= denom;
        B /= denom;
        const double radical = B * B - 3.0 * A * C;
        if (!(radical >= 0.0) || 3.0 * A == 0.0) return false;
        x = a + (-B + sqrt(radical)) / (3.0 * A);
        return finite(x);
    }
    // zoom: true if a step is accepted
    HYBRJ4_HD bool zoom(double a_lo, double a_hi, double phi_lo, double phi_hi, double dphi_lo) const {
        const double c1 = 1e-4, c2 = 0.9;
        double phi_rec = phi0, a_rec = 0.0, a_j = 0.0;
        for (int i = 0; i <= 10; i++) {
            const double dalpha = a_hi - a_lo;
            const double a = dalpha < 0.0 ? a_hi : a_lo, b = dalpha < 0.0 ? a_lo : a_hi;
            bool have = false;
            if (i > 0) {
                const double cchk = 0.2 * dalpha;
                have = cubicmin(a_lo, phi_lo, dphi_lo, a_hi, phi_hi, a_rec, phi_rec, a_j) && !(a_j > b - cchk) && !(a_j < a + cchk);
            }
            if (!have) {
                const double qchk = 0.1 * dalpha;
                if (!(quadmin(a_lo, phi_lo, dphi_lo, a_hi, phi_hi, a_j) && !(a_j > b - qchk) && !(a_j < a + qchk)))
                    a_j = a_lo + 0.5 * dalpha;
            }
            const double phi_aj = phi(a_j);
            if (phi_aj > phi0 + c1 * a_j * d0 || phi_aj >= phi_lo) {
                phi_rec = phi_hi;
                a_rec = a_hi;
                a_hi = a_j;
                phi_hi = phi_aj;
            } else {
                const double d_aj = dphi(a_j);
                if (fabs(d_aj) <= -c2 * d0) return true;
                if (d_aj * (a_hi - a_lo) >= 0.0) {
                    phi_rec = phi_hi;
                    a_rec = a_hi;
                    a_hi = a_lo;
                    phi_hi = phi_lo;
                } else {
                    phi_rec = phi_lo;
                    a_rec = a_lo;
                }
                a_lo = a_j;
                phi_lo = phi_aj;
                dphi_lo = d_aj;
            }
        }
        return false;
    }
    // Does fmin_bfgs leave its start?  (false: it returns (1/3, 1/3) -- the reference's fallback)
    HYBRJ4_HD bool moves() {
        double g0, g1;
        phi0 = fhat(1.0 / 3.0, 1.0 / 3.0);
        ghat(1.0 / 3.0, 1.0 / 3.0, g0, g1);
        if (!(fmax(fabs(g0), fabs(g1)) > 1e-5)) return false;          // gnorm <= gtol: no iteration at all
        const double old_phi0 = phi0 + sqrt(g0 * g0 + g1 * g1) / 2.0;
        pk0 = -g0;
        pk1 = -g1;
        d0 = g0 * pk0 + g1 * pk1;
        const double c1 = 1e-4, c2 = 0.9, amax = 1e100;
        double alpha0 = 0.0, alpha1 = d0 != 0.0 ? fmin(1.0, 1.01 * 2.0 * (phi0 - old_phi0) / d0) : 1.0;
        if (alpha1 < 0.0) alpha1 = 1.0;
        alpha1 = fmin(alpha1, amax);
        double phi_a1 = phi(alpha1), phi_a0 = phi0, d_a0 = d0;
        for (int i = 0; i < 10; i++) {
            if (alpha1 == 0.0 || alpha0 > amax) return false;
            if (phi_a1 > phi0 + c1 * alpha1 * d0 || (phi_a1 >= phi_a0 && i > 0)) return zoom(alpha0, alpha1, phi_a0, phi_a1, d_a0);
            const double d_a1 = dphi(alpha1);
            if (fabs(d_a1) <= -c2 * d0) return true;
            if (d_a1 >= 0.0) return zoom(alpha1, alpha0, phi_a1, phi_a0, d_a1);
            const double alpha2 = fmin(2.0 * alpha1, amax);
            alpha0 = alpha1;
            alpha1 = alpha2;
            phi_a0 = phi_a1;
            phi_a1 = phi(alpha1);
            d_a0 = d_a1;
        }
        return true;                                                    // iteration limit: the last trial step is taken
    }
};

// The reference's outcome class for one candidate: 1 = fsolve's iterate is in [0,1]^3 (nu filled in), 2 = the
// nu = (1/3,1/3,1/3) fallback, 0 = None (BFGS left its start for a point out of range / with NaN likelihood).
template <class SYS>
HYBRJ4_HD inline int n3_ref_outcome(SYS &sys, double nu[3]) {
    n3_ref_fsolve(sys, nu, nullptr);
    bool in = true;
    for (int j = 0; j < 3; j++)
        if (nu[j] < 0.0 || nu[j] > 1.0) in = false;                   // (NaN passes, Misc.py:49-57)
    if (in) return 1;
    N3RefBfgsT<SYS> b{sys, 0.0, 0.0, 0.0, 0.0};
    if (b.moves()) return 0;
    nu[0] = nu[1] = nu[2] = 1.0 / 3.0;
    return 2;
}

// Everything Optimizer._solve_n3plus does for one candidate (Optimizer.py:128-165), as the reference does it: outcome of
// the fsolve / fmin_bfgs calls (above), nu -> mu by M3's own fsolve call (n3_ref_M3), then Optimizer.L3's sums
// (Optimizer.py:236-244: the denominator accumulates column by column -- for j ... for h ... --, the numerators left to
// right).  Returns 0 = None, 1 = reported with its own iterate, 2 = reported at the nu = 1/3 fallback; mu, nll (NaN is a
// legitimate value: the reference returns such tuples) and vals[m] (may be null) are filled for 1 / 2.
// An all-zero tumour column makes Chat NaN: hybrj then returns its start unchanged, exactly like the reference's fsolve
// call, nu = (1/3,1/3,1/3) passes inRange, M3 lands on a unit vector plus rounding residue, and L3 makes a finite number
// or NaN of it -- reproduced, not special-cased.  Host and device run this same code.
template <class SYS>
HYBRJ4_HD inline int n3_ref_solve(SYS &sys, double mu[3], double &nll, double *vals) {
    double nu[3];
    const int outcome = n3_ref_outcome(sys, nu);
    if (outcome == 0) return 0;
    n3_ref_M3(sys.S, nu, mu, nullptr);
    nll = sys.l3(mu, vals);
    return outcome;
}

#ifdef HYBRJ4_MANAGE_CONTRACT
#pragma clang fp contract(fast)
#endif
