// pow(x, 2.0) as glibc's libm computes it (glibc >= 2.28, sysdeps/ieee754/dbl-64/e_pow.c -- Szabolcs Nagy's table-driven
// pow: log(x) to ~68 bits as k ln2 + log c + log1p(z/c - 1), times y, then exp through a 128-entry 2^(k/128) table), in the
// variant x86-64 machines with FMA dispatch to at run time (multiarch e_pow-fma.c).  Restated operation by operation,
// including WHICH multiply-adds are fused in that build (read off the instruction sequence the library executes: the
// polynomial evaluations, kd*Ln2hi + logc, z*InvLn2N + Shift, scale + scale*tmp are single roundings there).
//
// Why: the reference squares numpy float64 scalars with `**2` in its Jacobian (Optimizer.py:303-311: second_deriv), and
// numpy hands that to libm's pow -- whose result differs from the correctly rounded x*x in the last bit for about one
// argument in 1 300.  On a candidate whose tumour columns are linearly dependent with tau (x + y = const, ...) the bordered
// Jacobian is exactly singular, MINPACK's QR leaves rounding noise where a zero belongs and the whole trajectory of hybrj
// -- hence whether the reference reports the candidate's own optimum, its nu = 1/3 fallback or None -- hangs on that bit
// (8 of the 32 604 candidates of two small spaces in round 2's verdict).  Reproducing the reference's outcome on those
// candidates takes the reference's square.  glibc is a dependency of the reference's interpreter, not part of
// /root/reference; the tables are regenerated from glibc's published formulas (tools/gen_refpow_tables.py), and
// tests/test_refpow_cpu.py checks this function bit for bit against the libm of the machine the oracle runs on.
//
// Only y = 2 is restated (ehi = 2 hi and elo = 2 lo are exact, the general y*hi splitting is not needed).  Arguments
// outside the main path of the original -- zero, subnormal, infinite, NaN, and |2 log x| >= 512 or < 2^-54 (x = 1) -- return
// x*x, which is what pow returns there (exactly, or to within the underflow it rounds through); negative x squares |x|
// (pow's even-integer rule).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifndef HYBRJ4_HD
#ifdef __HIPCC__
#define HYBRJ4_HD __host__ __device__
#else
#define HYBRJ4_HD
#endif
#endif

namespace refpow {

struct LogEntry {
    double invc, logc, logctail;
};
struct ExpEntry {
    uint64_t tail, scale;
};

HYBRJ4_HD inline uint64_t bits_of(double x) {
    uint64_t u;
    memcpy(&u, &x, 8);
    return u;
}
HYBRJ4_HD inline double as_double(uint64_t u) {
    double x;
    memcpy(&x, &u, 8);
    return x;
}

HYBRJ4_HD inline double square(double x) {
#define REFPOW_LOG(a, b, c) {a, b, c},
#define REFPOW_EXP(a, b)
    static constexpr LogEntry LT[128] = {
#include "refpow_tables.inc"
    };
#undef REFPOW_LOG
#undef REFPOW_EXP
#define REFPOW_LOG(a, b, c)
#define REFPOW_EXP(a, b) {a, b},
    static constexpr ExpEntry ET[128] = {
#include "refpow_tables.inc"
    };
#undef REFPOW_LOG
#undef REFPOW_EXP
    // e_pow_log_data.c: ln2hi, ln2lo, poly A[0..6] (scaled as the evaluation wants them)
    constexpr double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    constexpr double A0 = -0x1p-1, A1 = 0x1.555555555556p-2 * -2, A2 = -0x1.0000000000006p-2 * -2, A3 = 0x1.999999959554ep-3 * 4,
                     A4 = -0x1.555555529a47ap-3 * 4, A5 = 0x1.2495b9b4845e9p-3 * -8, A6 = -0x1.0002b8b263fc3p-3 * -8;
    // e_exp_data.c (N = 128): invln2N, shift, negln2hiN, negln2loN, poly C2..C5
    constexpr double InvLn2N = 0x1.71547652b82fep0 * 128, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8,
                     NegLn2loN = -0x1.cf79abc9e3b3ap-47;
    constexpr double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;

    const uint64_t ix = bits_of(x) & 0x7fffffffffffffffULL;                 // pow(-x, 2) = pow(x, 2)
    const uint32_t topx = (uint32_t)(ix >> 52);
    if (topx - 1u >= 0x7ffu - 1u) return x * x;                               // 0, subnormal, inf, NaN
    // log_inline: x = 2^k z, z in [OFF, 2 OFF), c near the centre of z's subinterval
    const uint64_t tmp = ix - 0x3fe6955500000000ULL;
    const int i = (int)((tmp >> 45) & 127);
    const int k = (int)((int64_t)tmp >> 52);
    const uint64_t iz = ix - (tmp & (0xfffULL << 52));
    const double z = as_double(iz), kd = (double)k;
    const double invc = LT[i].invc, logc = LT[i].logc, logctail = LT[i].logctail;
    const double r = __builtin_fma(z, invc, -1.0);
    const double t1 = __builtin_fma(kd, Ln2hi, logc);
    const double t2 = t1 + r;
    const double lo1 = __builtin_fma(kd, Ln2lo, logctail);
    const double lo2 = (t1 - t2) + r;
    const double ar = A0 * r;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = __builtin_fma(ar, r, -ar2);
    const double lo4 = (t2 - hi) + ar2;
    const double q12 = __builtin_fma(r, A2, A1), q34 = __builtin_fma(r, A4, A3), q56 = __builtin_fma(r, A6, A5);
    const double q = __builtin_fma(ar2, __builtin_fma(q56, ar2, q34), q12);
    const double lo = __builtin_fma(ar3, q, ((lo1 + lo2) + lo3) + lo4);
    const double ylog = hi + lo;
    const double ltail = (hi - ylog) + lo;
    // y = 2
    const double ehi = 2.0 * ylog;
    const double elo = __builtin_fma(2.0, ltail, __builtin_fma(ylog, 2.0, -ehi));
    // exp_inline(ehi, elo)
    const uint32_t abstop = (uint32_t)(bits_of(ehi) >> 52) & 0x7ff;
    if (abstop - 0x3c9u >= 0x3fu) return x * x;
    const double kds = __builtin_fma(ehi, InvLn2N, Shift);
    const uint64_t ki = bits_of(kds);
    const double kd2 = kds - Shift;
    double rr = __builtin_fma(kd2, NegLn2loN, __builtin_fma(kd2, NegLn2hiN, ehi));
    rr = elo + rr;
    const int idx = (int)(ki & 127);
    const uint64_t sbits = ET[idx].scale + (ki << 45);
    const double tail = as_double(ET[idx].tail);
    const double r2 = rr * rr;
    const double p23 = __builtin_fma(rr, C3, C2), p45 = __builtin_fma(rr, C5, C4);
    const double inner = __builtin_fma(p23, r2, rr + tail);
    const double tm = __builtin_fma(p45, r2 * r2, inner);
    const double scale = as_double(sbits);
    return __builtin_fma(tm, scale, scale);
}

}   // namespace refpow
