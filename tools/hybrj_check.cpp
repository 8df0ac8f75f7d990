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
