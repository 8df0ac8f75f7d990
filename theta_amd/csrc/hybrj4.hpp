// MINPACK's hybrj (Powell's hybrid method with a user Jacobian) restated for systems of N = 4 unknowns -- the solver
// behind scipy.optimize.fsolve(equations, x0, fprime=jacobian), which is what Optimizer._solve_n3plus calls
// (Optimizer.py:148) on the Lagrangian system of the n = 3 mixture fit.
//
// Why restate a root finder: what the reference REPORTS for a candidate is not "the optimum of its likelihood" but
// "whatever hybrj returns from the start (1/3, 1/3, 1/3, 1)": where the iteration ends on a root of the rational system
// outside [0,1]^3 -- although the likelihood has its minimum inside the simplex -- the reference falls through to its
// nu = (1/3,1/3,1/3) fallback (5 % of mid-size random instances have such a candidate as the best
// fit).  Reproducing that decision needs the same trajectory, so this follows the published algorithm (More, Garbow,
// Hillstrom: User Guide for MINPACK-1, ANL-80-74; routines hybrj, dogleg, qrfac, qform, r1updt, r1mpyq, enorm) step by
// step, in double precision, in the same operation order, with no fused multiply-adds (the unit is compiled with
// -ffp-contract=off).  MINPACK is a dependency of the reference (through scipy), not part of /root/reference.
//
// Indices are 1-based like the Fortran (arrays carry an unused element 0) to keep the packed-triangle bookkeeping
// identical.  FCN is a functor:  f(const double x[5], double fvec[5])  and  jac(const double x[5], double fj[5][5])
// with fj[i][j] = d f_i / d x_j.
#pragma once
#include <math.h>

#ifndef HYBRJ4_HD
#ifdef __HIPCC__
#define HYBRJ4_HD __host__ __device__
#else
#define HYBRJ4_HD
#endif
#endif

// A translation unit compiled with fused multiply-adds allowed (n3.hip) defines HYBRJ4_MANAGE_CONTRACT before including
// this header: the trajectory has to be the one of separately rounded products and sums.
#ifdef HYBRJ4_MANAGE_CONTRACT
#pragma clang fp contract(off)
#endif

namespace hybrj4 {

constexpr int N = 4;
constexpr int LR = N * (N + 1) / 2;
constexpr double EPSMCH = 2.220446049250313e-16;    // dpmpar(1)
constexpr double GIANT = 1.79769313485e308;         // dpmpar(3)

// enorm: scaled Euclidean norm of x(lo..hi)
HYBRJ4_HD inline double enorm(const double *x, int lo, int hi) {
    const double rdwarf = 3.834e-20, rgiant = 1.304e19;
    double s1 = 0.0, s2 = 0.0, s3 = 0.0, x1max = 0.0, x3max = 0.0;
    const double agiant = rgiant / (double)(hi - lo + 1);
    for (int i = lo; i <= hi; i++) {
        const double xabs = fabs(x[i]);
        if (xabs > rdwarf && xabs < agiant) {
            s2 += xabs * xabs;
        } else if (xabs <= rdwarf) {
            if (xabs > x3max) {
                const double t = x3max / xabs;
                s3 = 1.0 + s3 * (t * t);
                x3max = xabs;
            } else if (xabs != 0.0) {
                const double t = xabs / x3max;
                s3 += t * t;
            }
        } else {
            if (xabs > x1max) {
                const double t = x1max / xabs;
                s1 = 1.0 + s1 * (t * t);
                x1max = xabs;
            } else {
                const double t = xabs / x1max;
                s1 += t * t;
            }
        }
    }
    if (s1 != 0.0) return x1max * sqrt(s1 + (s2 / x1max) / x1max);
    if (s2 != 0.0) {
        if (s2 >= x3max) return sqrt(s2 * (1.0 + (x3max / s2) * (x3max * s3)));
        return sqrt(x3max * ((s2 / x3max) + (x3max * s3)));
    }
    return x3max * sqrt(s3);
}

// qrfac without pivoting: Householder QR of the N x N matrix a (in place), rdiag = diagonal of R, acnorm = column norms
HYBRJ4_HD inline void qrfac(double a[N + 1][N + 1], double *rdiag, double *acnorm) {
    double col[N + 1];
    for (int j = 1; j <= N; j++) {
        for (int i = 1; i <= N; i++) col[i] = a[i][j];
        acnorm[j] = enorm(col, 1, N);
        rdiag[j] = acnorm[j];
    }
    for (int j = 1; j <= N; j++) {
        for (int i = j; i <= N; i++) col[i] = a[i][j];
        double ajnorm = enorm(col, j, N);
        if (ajnorm != 0.0) {
            if (a[j][j] < 0.0) ajnorm = -ajnorm;
            for (int i = j; i <= N; i++) a[i][j] = a[i][j] / ajnorm;
            a[j][j] = a[j][j] + 1.0;
            for (int k = j + 1; k <= N; k++) {
                double sum = 0.0;
                for (int i = j; i <= N; i++) sum = sum + a[i][j] * a[i][k];
                const double temp = sum / a[j][j];
                for (int i = j; i <= N; i++) a[i][k] = a[i][k] - temp * a[i][j];
            }
        }
        rdiag[j] = -ajnorm;
    }
}

// qform: accumulate Q from its factored form
HYBRJ4_HD inline void qform(double q[N + 1][N + 1]) {
    double wa[N + 1];
    for (int j = 2; j <= N; j++)
        for (int i = 1; i <= j - 1; i++) q[i][j] = 0.0;
    for (int l = 1; l <= N; l++) {
        const int k = N - l + 1;
        for (int i = k; i <= N; i++) {
            wa[i] = q[i][k];
            q[i][k] = 0.0;
        }
        q[k][k] = 1.0;
        if (wa[k] != 0.0) {
            for (int j = k; j <= N; j++) {
                double sum = 0.0;
                for (int i = k; i <= N; i++) sum = sum + q[i][j] * wa[i];
                const double temp = sum / wa[k];
                for (int i = k; i <= N; i++) q[i][j] = q[i][j] - temp * wa[i];
            }
        }
    }
}

// dogleg: the convex combination of the Gauss-Newton and the scaled gradient direction inside the trust region delta
HYBRJ4_HD inline void dogleg(const double *r, const double *diag, const double *qtb, double delta, double *x, double *wa1,
                             double *wa2) {
    int jj = (N * (N + 1)) / 2 + 1;
    for (int k = 1; k <= N; k++) {
        const int j = N - k + 1, jp1 = j + 1;
        jj = jj - k;
        int l = jj + 1;
        double sum = 0.0;
        for (int i = jp1; i <= N; i++) {
            sum = sum + r[l] * x[i];
            l = l + 1;
        }
        double temp = r[jj];
        if (temp == 0.0) {
            l = j;
            for (int i = 1; i <= j; i++) {
                temp = fmax(temp, fabs(r[l]));
                l = l + N - i;
            }
            temp = EPSMCH * temp;
            if (temp == 0.0) temp = EPSMCH;
        }
        x[j] = (qtb[j] - sum) / temp;
    }
    for (int j = 1; j <= N; j++) {
        wa1[j] = 0.0;
        wa2[j] = diag[j] * x[j];
    }
    const double qnorm = enorm(wa2, 1, N);
    if (qnorm <= delta) return;
    int l = 1;
    for (int j = 1; j <= N; j++) {
        const double temp = qtb[j];
        for (int i = j; i <= N; i++) {
            wa1[i] = wa1[i] + r[l] * temp;
            l = l + 1;
        }
        wa1[j] = wa1[j] / diag[j];
    }
    const double gnorm = enorm(wa1, 1, N);
    double sgnorm = 0.0, alpha = delta / qnorm;
    if (gnorm != 0.0) {
        for (int j = 1; j <= N; j++) wa1[j] = (wa1[j] / gnorm) / diag[j];
        l = 1;
        for (int j = 1; j <= N; j++) {
            double sum = 0.0;
            for (int i = j; i <= N; i++) {
                sum = sum + r[l] * wa1[i];
                l = l + 1;
            }
            wa2[j] = sum;
        }
        double temp = enorm(wa2, 1, N);
        sgnorm = (gnorm / temp) / temp;
        alpha = 0.0;
        if (sgnorm < delta) {
            const double bnorm = enorm(qtb, 1, N);
            temp = (bnorm / gnorm) * (bnorm / qnorm) * (sgnorm / delta);
            const double dq = delta / qnorm, sd = sgnorm / delta;
            temp = temp - dq * (sd * sd) + sqrt((temp - dq) * (temp - dq) + (1.0 - dq * dq) * (1.0 - sd * sd));
            alpha = (dq * (1.0 - sd * sd)) / temp;
        }
    }
    const double temp = (1.0 - alpha) * fmin(sgnorm, delta);
    for (int j = 1; j <= N; j++) x[j] = temp * wa1[j] + alpha * x[j];
}

// r1updt: QR factors of  s + u v^T  from those of s (packed triangle), Givens rotations recorded in v and w
HYBRJ4_HD inline void r1updt(double *s, const double *u, double *v, double *w, bool &sing) {
    int jj = (N * (2 * N - N + 1)) / 2;
    int l = jj;
    for (int i = N; i <= N; i++) {
        w[i] = s[l];
        l = l + 1;
    }
    for (int nmj = 1; nmj <= N - 1; nmj++) {
        const int j = N - nmj;
        jj = jj - (N - j + 1);
        w[j] = 0.0;
        if (v[j] != 0.0) {
            double cs, sn, tau;
            if (fabs(v[N]) < fabs(v[j])) {
                const double cotan = v[N] / v[j];
                sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                cs = sn * cotan;
                tau = 1.0;
                if (fabs(cs) * GIANT > 1.0) tau = 1.0 / cs;
            } else {
                const double tn = v[j] / v[N];
                cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                sn = cs * tn;
                tau = sn;
            }
            v[N] = sn * v[j] + cs * v[N];
            v[j] = tau;
            l = jj;
            for (int i = j; i <= N; i++) {
                const double temp = cs * s[l] - sn * w[i];
                w[i] = sn * s[l] + cs * w[i];
                s[l] = temp;
                l = l + 1;
            }
        }
    }
    for (int i = 1; i <= N; i++) w[i] = w[i] + v[N] * u[i];
    sing = false;
    for (int j = 1; j <= N - 1; j++) {
        if (w[j] != 0.0) {
            double cs, sn, tau;
            if (fabs(s[jj]) < fabs(w[j])) {
                const double cotan = s[jj] / w[j];
                sn = 0.5 / sqrt(0.25 + 0.25 * (cotan * cotan));
                cs = sn * cotan;
                tau = 1.0;
                if (fabs(cs) * GIANT > 1.0) tau = 1.0 / cs;
            } else {
                const double tn = w[j] / s[jj];
                cs = 0.5 / sqrt(0.25 + 0.25 * (tn * tn));
                sn = cs * tn;
                tau = sn;
            }
            l = jj;
            for (int i = j; i <= N; i++) {
                const double temp = cs * s[l] + sn * w[i];
                w[i] = -sn * s[l] + cs * w[i];
                s[l] = temp;
                l = l + 1;
            }
            w[j] = tau;
        }
        if (s[jj] == 0.0) sing = true;
        jj = jj + (N - j + 1);
    }
    l = jj;
    for (int i = N; i <= N; i++) {
        s[l] = w[i];
        l = l + 1;
    }
    if (s[jj] == 0.0) sing = true;
}

// r1mpyq: apply the 2 (N-1) recorded Givens rotations to the rows of a (M x N)
template <int M>
HYBRJ4_HD inline void r1mpyq(double a[M + 1][N + 1], const double *v, const double *w) {
    for (int nmj = 1; nmj <= N - 1; nmj++) {
        const int j = N - nmj;
        double cs, sn;
        if (fabs(v[j]) > 1.0) {
            cs = 1.0 / v[j];
            sn = sqrt(1.0 - cs * cs);
        } else {
            sn = v[j];
            cs = sqrt(1.0 - sn * sn);
        }
        for (int i = 1; i <= M; i++) {
            const double temp = cs * a[i][j] - sn * a[i][N];
            a[i][N] = sn * a[i][j] + cs * a[i][N];
            a[i][j] = temp;
        }
    }
    for (int j = 1; j <= N - 1; j++) {
        double cs, sn;
        if (fabs(w[j]) > 1.0) {
            cs = 1.0 / w[j];
            sn = sqrt(1.0 - cs * cs);
        } else {
            sn = w[j];
            cs = sqrt(1.0 - sn * sn);
        }
        for (int i = 1; i <= M; i++) {
            const double temp = cs * a[i][j] + sn * a[i][N];
            a[i][N] = -sn * a[i][j] + cs * a[i][N];
            a[i][j] = temp;
        }
    }
}

// hybrj / hybrd, mode 1 (automatic scaling), nprint 0.  x(1..4): start on entry, final iterate on return.  Returns
// MINPACK's info.  FD = false: hybrj, the Jacobian comes from fcn.jac.  FD = true: hybrd, the Jacobian is built by
// fdjac1's forward differences (dense form: ml + mu + 1 >= n, epsfcn = 0, i.e. steps of sqrt(epsmch) |x_j|), which costs N
// function evaluations each time -- the solver behind scipy.optimize.fsolve(func, x0) WITHOUT fprime, which is how the
// reference maps nu to mu (Optimizer.M3, Optimizer.py:327-330).
template <bool FD, class FCN>
HYBRJ4_HD inline int hybr(FCN &fcn, double *x, double xtol, int maxfev, double factor, int *nfev_out) {
    double fvec[N + 1], fjac[N + 1][N + 1], diag[N + 1], r[LR + 1], qtf[N + 1], wa1[N + 1], wa2[N + 1], wa3[N + 1], wa4[N + 1];
    int info = 0, nfev = 0;
    for (int j = 1; j <= N; j++) diag[j] = 1.0;
    fcn.f(x, fvec);
    nfev = 1;
    double fnorm = enorm(fvec, 1, N);
    int iter = 1, ncsuc = 0, ncfail = 0, nslow1 = 0, nslow2 = 0;
    double delta = 0.0, xnorm = 0.0;
    while (true) {                                   // outer loop: new Jacobian
        bool jeval = true;
        if constexpr (FD) {                          // fdjac1, dense
            const double eps = sqrt(EPSMCH);
            for (int j = 1; j <= N; j++) {
                const double temp = x[j];
                double h = eps * fabs(temp);
                if (h == 0.0) h = eps;
                x[j] = temp + h;
                fcn.f(x, wa1);
                x[j] = temp;
                for (int i = 1; i <= N; i++) fjac[i][j] = (wa1[i] - fvec[i]) / h;
            }
            nfev = nfev + N;
        } else {
            fcn.jac(x, fjac);
        }
        qrfac(fjac, wa1, wa2);
        if (iter == 1) {
            for (int j = 1; j <= N; j++) {
                diag[j] = wa2[j];
                if (wa2[j] == 0.0) diag[j] = 1.0;
            }
            for (int j = 1; j <= N; j++) wa3[j] = diag[j] * x[j];
            xnorm = enorm(wa3, 1, N);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 1; i <= N; i++) qtf[i] = fvec[i];
        for (int j = 1; j <= N; j++) {
            if (fjac[j][j] != 0.0) {
                double sum = 0.0;
                for (int i = j; i <= N; i++) sum = sum + fjac[i][j] * qtf[i];
                const double temp = -sum / fjac[j][j];
                for (int i = j; i <= N; i++) qtf[i] = qtf[i] + fjac[i][j] * temp;
            }
        }
        bool sing = false;
        for (int j = 1; j <= N; j++) {
            int l = j;
            for (int i = 1; i <= j - 1; i++) {
                r[l] = fjac[i][j];
                l = l + N - i;
            }
            r[l] = wa1[j];
            if (wa1[j] == 0.0) sing = true;
        }
        (void)sing;
        qform(fjac);
        for (int j = 1; j <= N; j++) diag[j] = fmax(diag[j], wa2[j]);
        while (true) {                               // inner loop: rank-one updates of the factorisation
            dogleg(r, diag, qtf, delta, wa1, wa2, wa3);
            for (int j = 1; j <= N; j++) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = enorm(wa3, 1, N);
            if (iter == 1) delta = fmin(delta, pnorm);
            fcn.f(wa2, wa4);
            nfev = nfev + 1;
            const double fnorm1 = enorm(wa4, 1, N);
            double actred = -1.0;
            if (fnorm1 < fnorm) actred = 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            int l = 1;
            for (int i = 1; i <= N; i++) {
                double sum = 0.0;
                for (int j = i; j <= N; j++) {
                    sum = sum + r[l] * wa1[j];
                    l = l + 1;
                }
                wa3[i] = qtf[i] + sum;
            }
            const double temp = enorm(wa3, 1, N);
            double prered = 0.0;
            if (temp < fnorm) prered = 1.0 - (temp / fnorm) * (temp / fnorm);
            double ratio = 0.0;
            if (prered > 0.0) ratio = actred / prered;
            if (ratio >= 0.1) {
                ncfail = 0;
                ncsuc = ncsuc + 1;
                if (ratio >= 0.5 || ncsuc > 1) delta = fmax(delta, pnorm / 0.5);
                if (fabs(ratio - 1.0) <= 0.1) delta = pnorm / 0.5;
            } else {
                ncsuc = 0;
                ncfail = ncfail + 1;
                delta = 0.5 * delta;
            }
            if (ratio >= 1.0e-4) {
                for (int j = 1; j <= N; j++) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                    fvec[j] = wa4[j];
                }
                xnorm = enorm(wa2, 1, N);
                fnorm = fnorm1;
                iter = iter + 1;
            }
            nslow1 = nslow1 + 1;
            if (actred >= 1.0e-3) nslow1 = 0;
            if (jeval) nslow2 = nslow2 + 1;
            if (actred >= 0.1) nslow2 = 0;
            if (delta <= xtol * xnorm || fnorm == 0.0) info = 1;
            if (info != 0) goto done;
            if (nfev >= maxfev) info = 2;
            if (0.1 * fmax(0.1 * delta, pnorm) <= EPSMCH * xnorm) info = 3;
            if (nslow2 == 5) info = 4;
            if (nslow1 == 10) info = 5;
            if (info != 0) goto done;
            if (ncfail == 2) break;                  // recompute the Jacobian
            for (int j = 1; j <= N; j++) {
                double sum = 0.0;
                for (int i = 1; i <= N; i++) sum = sum + fjac[i][j] * wa4[i];
                wa2[j] = (sum - wa3[j]) / pnorm;
                wa1[j] = diag[j] * ((diag[j] * wa1[j]) / pnorm);
                if (ratio >= 1.0e-4) qtf[j] = sum;
            }
            r1updt(r, wa1, wa2, wa3, sing);
            r1mpyq<N>(fjac, wa2, wa3);
            {
                double q1[2][N + 1];
                for (int j = 1; j <= N; j++) q1[1][j] = qtf[j];
                r1mpyq<1>(q1, wa2, wa3);
                for (int j = 1; j <= N; j++) qtf[j] = q1[1][j];
            }
            jeval = false;
        }
    }
done:
    if (nfev_out) *nfev_out = nfev;
    return info;
}

template <class FCN>
HYBRJ4_HD inline int hybrj(FCN &fcn, double *x, double xtol, int maxfev, double factor, int *nfev_out) {
    return hybr<false>(fcn, x, xtol, maxfev, factor, nfev_out);
}
template <class FCN>
HYBRJ4_HD inline int hybrd(FCN &fcn, double *x, double xtol, int maxfev, double factor, int *nfev_out) {
    return hybr<true>(fcn, x, xtol, maxfev, factor, nfev_out);
}

}   // namespace hybrj4
