// Host build of the hybrj restatement (theta_amd/csrc/hybrj4.hpp + n3_refsys.hpp) for checking it against scipy's fsolve:
//   g++ -O2 -ffp-contract=off -shared -fPIC tools/hybrj_check.cpp -o build_ab/libhybrj_check.so
#include "../theta_amd/csrc/n3_refbfgs.hpp"
extern "C" int hybrj_check_solve(int m, int tau, const double *r, const double *rN, const unsigned char *c, double *nu, int *nfev) {
    N3RefSystem s;
    s.m = m;
    s.tau = (double)tau;
    s.r = r;
    s.rN = rN;
    s.c = c;
    s.init();
    return n3_ref_fsolve(s, nu, nfev);
}
extern "C" int hybrj_check_outcome(int m, int tau, const double *r, const double *rN, const unsigned char *c, double *nu) {
    N3RefSystem s;
    s.m = m;
    s.tau = (double)tau;
    s.r = r;
    s.rN = rN;
    s.c = c;
    s.init();
    return n3_ref_outcome(s, nu);
}
extern "C" int hybrj_check_M3(const double *S, const double *nu, double *mu, int *nfev) { return n3_ref_M3(S, nu, mu, nfev); }
// the per-candidate procedure of theta_solve_batch (n = 3) on the host: the SAME function the device kernel calls
extern "C" void hybrj_check_table(int B, int m, int tau, const double *r, const double *rN, const unsigned char *C, unsigned char *ok,
                                  double *mu, double *nll) {
    for (int b = 0; b < B; b++) {
        N3RefSystem s;
        s.m = m;
        s.tau = (double)tau;
        s.r = r;
        s.rN = rN;
        s.c = C + (size_t)b * m * 2;
        s.init();
        double value = NAN;
        ok[b] = (unsigned char)n3_ref_solve(s, mu + 3 * b, value, nullptr);
        nll[b] = value;
    }
}
// trace of the function evaluations of one fsolve run: xs[k][4], fs[k][4] for evaluation k (k < cap); returns the count
struct TraceSys {
    N3RefSystem s;
    double *xs, *fs;
    int cap, n;
    void f(const double *x, double *fv) {
        s.f(x, fv);
        if (n < cap) {
            for (int j = 0; j < 4; j++) {
                xs[4 * n + j] = x[j + 1];
                fs[4 * n + j] = fv[j + 1];
            }
        }
        n++;
    }
    void jac(const double *x, double fj[hybrj4::N + 1][hybrj4::N + 1]) const { s.jac(x, fj); }
};
extern "C" int hybrj_check_trace(int m, int tau, const double *r, const double *rN, const unsigned char *c, double *xs, double *fs, int cap,
                                 int *info) {
    TraceSys t;
    t.s.m = m;
    t.s.tau = (double)tau;
    t.s.r = r;
    t.s.rN = rN;
    t.s.c = c;
    t.s.init();
    t.xs = xs;
    t.fs = fs;
    t.cap = cap;
    t.n = 0;
    double x[hybrj4::N + 1] = {0.0, 1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0, 1.0};
    *info = hybrj4::hybrj(t, x, 1.49012e-8, 100 * (hybrj4::N + 1), 100.0, nullptr);
    return t.n;
}
// the restated square of refpow.hpp and libm's own pow(x, 2.0) (this unit is compiled with -fno-builtin-pow so that the
// compiler does not turn the call into x*x) on n arguments
extern "C" void refpow_check_square(int n, const double *x, double *out) {
    for (int i = 0; i < n; i++) out[i] = refpow::square(x[i]);
}
extern "C" void refpow_check_libm(int n, const double *x, double *out) {
    volatile double two = 2.0;
    for (int i = 0; i < n; i++) out[i] = pow(x[i], two);
}
