// The n = 3 Lagrangian system of Optimizer._solve_n3plus in the reference's own operation order, as the FCN of
// hybrj4::hybrj (host and device).  Restates, for one candidate C (m x 2 bytes; column 0 of the matrix is tau):
//   weighted_C / normalize_C     Optimizer.py:167-182   Cw_ij = rN_i C_ij,  Chat_ij = Cw_ij / sum_i Cw_ij
//   equations / dLambda_dMu      Optimizer.py:273-286, 313-316
//   jacobian / second_deriv      Optimizer.py:288-311
// Sums run over the intervals top to bottom like the reference's Python loops; no fused multiply-adds.
#pragma once
#include "hybrj4.hpp"
#include "refpow.hpp"

struct N3RefSystem {
    int m;
    double tau;
    const double *r, *rN;          // [m]
    const unsigned char *c;        // [m][2]
    double S[3];                   // column sums of the weighted matrix

    HYBRJ4_HD void init() {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < m; i++) {
            s0 = s0 + rN[i] * tau;
            s1 = s1 + rN[i] * (double)c[2 * i];
            s2 = s2 + rN[i] * (double)c[2 * i + 1];
        }
        S[0] = s0;
        S[1] = s1;
        S[2] = s2;
    }
    HYBRJ4_HD void chat(int i, double &h0, double &h1, double &h2) const {
        h0 = (rN[i] * tau) / S[0];
        h1 = (rN[i] * (double)c[2 * i]) / S[1];
        h2 = (rN[i] * (double)c[2 * i + 1]) / S[2];
    }
    // x(1..4) = (nu0, nu1, nu2, lambda)
    HYBRJ4_HD void f(const double *x, double *fv) const {
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int i = 0; i < m; i++) {
            double h0, h1, h2;
            chat(i, h0, h1, h2);
            const double p = (h0 * x[1] + h1 * x[2]) + h2 * x[3];
            a0 = a0 + (r[i] * h0) / p;
            a1 = a1 + (r[i] * h1) / p;
            a2 = a2 + (r[i] * h2) / p;
        }
        fv[1] = (-a0) - x[4];
        fv[2] = (-a1) - x[4];
        fv[3] = (-a2) - x[4];
        fv[4] = 1.0 - ((x[1] + x[2]) + x[3]);
    }
    HYBRJ4_HD void jac(const double *x, double fj[hybrj4::N + 1][hybrj4::N + 1]) const {
        double J[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (int i = 0; i < m; i++) {
            double h[3];
            chat(i, h[0], h[1], h[2]);
            const double p = (h[0] * x[1] + h[1] * x[2]) + h[2] * x[3];
            const double den = refpow::square(p);          // `**2` on a numpy float64 is libm's pow(p, 2.0), not p*p (refpow.hpp)
            for (int k = 0; k < 3; k++)
                for (int q = 0; q < 3; q++) J[k][q] = J[k][q] + ((r[i] * h[k]) * h[q]) / den;
        }
        for (int k = 0; k < 3; k++)
            for (int q = 0; q < 3; q++) fj[k + 1][q + 1] = J[k][q];
        for (int k = 1; k <= 3; k++) {
            fj[4][k] = -1.0;
            fj[k][4] = -1.0;
        }
        fj[4][4] = 0.0;
    }
    // L3_hat (Optimizer.py:246-252) at (v0, v1); NaN outside the domain like numpy's log
    HYBRJ4_HD double fhat(double v0, double v1) const {
        double acc = 0.0;
        const double v2 = 1.0 - (v0 + v1);
        for (int i = 0; i < m; i++) {
            double h0, h1, h2;
            chat(i, h0, h1, h2);
            const double p = (h0 * v0 + h1 * v1) + h2 * v2;
            if (p < 0.0 || p != p) return NAN;
            acc = acc + r[i] * log(p);          // log(0) = -inf like numpy
        }
        return -acc;
    }
    // dL3_hat (Optimizer.py:255-265)
    HYBRJ4_HD void ghat(double v0, double v1, double &g0, double &g1) const {
        g0 = 0.0;
        g1 = 0.0;
        for (int i = 0; i < m; i++) {
            double h0, h1, h2;
            chat(i, h0, h1, h2);
            const double n0 = h0 - h2, n1 = h1 - h2;
            const double den = (n0 * v0 + n1 * v1) + h2;
            g0 = g0 + r[i] * (n0 / den);
            g1 = g1 + r[i] * (n1 / den);
        }
    }
    // Optimizer.L3's sums (Optimizer.py:236-244: the denominator accumulates column by column -- for j ... for h ... --, the numerators
    // left to right); vals[m] may be null
    HYBRJ4_HD double l3(const double mu[3], double *vals) const {
        const double m0 = mu[0], m1 = mu[1], m2 = mu[2];
        double den = 0.0;
        for (int h = 0; h < m; h++) den = den + (rN[h] * tau) * m0;
        for (int h = 0; h < m; h++) den = den + (rN[h] * (double)c[2 * h]) * m1;
        for (int h = 0; h < m; h++) den = den + (rN[h] * (double)c[2 * h + 1]) * m2;
        double tot = 0.0;
        for (int i = 0; i < m; i++) {
            const double nm = ((rN[i] * tau) * m0 + (rN[i] * (double)c[2 * i]) * m1) + (rN[i] * (double)c[2 * i + 1]) * m2;
            const double p = nm / den;
            tot = tot + r[i] * log(p);               // log of a negative number is NaN, like numpy's
            if (vals) vals[i] = p;
        }
        return -tot;
    }
};

// fsolve(equations, [1/3,1/3,1/3,1], fprime=jacobian) with scipy's defaults (xtol 1.49012e-8, maxfev 100 (n+1), factor 100).
// (SYS: N3RefSystem, or the wave-cooperative system of batch.hip -- same interface, same operations in the same order)
template <class SYS>
HYBRJ4_HD inline int n3_ref_fsolve(SYS &sys, double nu[3], int *nfev) {
    double x[hybrj4::N + 1] = {0.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0, 1.0};
    const int info = hybrj4::hybrj(sys, x, 1.49012e-8, 100 * (hybrj4::N + 1), 100.0, nfev);
    nu[0] = x[1];
    nu[1] = x[2];
    nu[2] = x[3];
    return info;
}

// Optimizer.M3 (Optimizer.py:318-330): nu -> mu as the reference computes it, fsolve(M_eq, [.33, .33, .33, 0]) WITHOUT a
// Jacobian, i.e. MINPACK's hybrd with forward differences, on the LINEAR system
//      eq_j = nu_j (sum_h x_h S_h) - x_j S_j - x_4   (j < 3),     eq_4 = (x_1 + x_2 + x_3) - 1
// (S = column sums of the weighted matrix).  Its solution is the closed form mu_j = (nu_j / S_j) / sum_h (nu_h / S_h) --
// unless a column sum is zero (an all-zero tumour column): then the exact solution puts ALL weight on that column, the
// mixture C.mu vanishes, and what the reference goes on to report (a finite NLL or NaN, Optimizer.L3) hangs on the
// rounding residue hybrd leaves in the other components (~1e-26).  Restated operation by operation, Python's sums left
// to right from 0.
struct N3RefM3 {
    double S[3], nu[3];
    HYBRJ4_HD void f(const double *x, double *fv) const {
        for (int j = 0; j < 3; j++) {
            const double t = ((x[1] * S[0]) + x[2] * S[1]) + x[3] * S[2];
            fv[j + 1] = ((nu[j] * t) - (x[j + 1] * S[j])) - x[4];
        }
        fv[4] = ((x[1] + x[2]) + x[3]) - 1.0;
    }
};
HYBRJ4_HD inline int n3_ref_M3(const double S[3], const double nu[3], double mu[3], int *nfev) {
    N3RefM3 sys;
    for (int j = 0; j < 3; j++) {
        sys.S[j] = S[j];
        sys.nu[j] = nu[j];
    }
    double x[hybrj4::N + 1] = {0.0, 0.33, 0.33, 0.33, 0.0};
    const int info = hybrj4::hybrd(sys, x, 1.49012e-8, 200 * (hybrj4::N + 1), 100.0, nfev);
    mu[0] = x[1];
    mu[1] = x[2];
    mu[2] = x[3];
    return info;
}

#ifdef HYBRJ4_MANAGE_CONTRACT
#pragma clang fp contract(fast)      // back to the device default for the rest of the including unit
#endif
