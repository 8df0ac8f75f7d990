// The scorers' logarithm (batch.hip: score_masked_mfma_kernel, score_plain_kernel).  Host + device: tools/smx_log_check.hip
// evaluates it on the CPU against 60-digit arithmetic (tests/test_smx_log_cpu.py).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

// ln(x) for the scorers below (one per (candidate, interval) and one per (candidate, mask) pair), table-driven:
// x = 2^k m, m in [1, 2); entry j = top 7 mantissa bits holds inv_c ~ 1 / (1 + (j + 1/2)/128) and log_c = -ln(inv_c) of
// that very double (tools/gen_log_table.py, 60-digit arithmetic), so ln x = k ln 2 + log_c + log1p(f) with
// f = m inv_c - 1 from ONE fma, |f| <= 2^-8, and log1p by its series to f^6 (next term < 2e-18).  Absolute error
// ~2e-16; 15 vector instructions and one 16-byte LDS read, against ~35 for the fdlibm reduction used before and ~90
// for the library call.  Anything that is not a positive normal number goes to the library.
// (bit-level access written with memcpy so that the same lines compile for the host)
__host__ __device__ __forceinline__ int smx_hi(double x) {
    unsigned long long u;
    __builtin_memcpy(&u, &x, 8);
    return (int)(u >> 32);
}
__host__ __device__ __forceinline__ double smx_with_hi(double x, int hi) {
    unsigned long long u;
    __builtin_memcpy(&u, &x, 8);
    u = (u & 0xffffffffull) | ((unsigned long long)(unsigned)hi << 32);
    double y;
    __builtin_memcpy(&y, &u, 8);
    return y;
}
__device__ const unsigned long long smx_log_table[256] = {
#include "smx_log_table.inc"
};
#define SMX_TAB_BYTES 2048
// d = a b + c in one rounding, as v_fma_f64 with three register operands.  Written out because the compiler turns a polynomial
// step with a loop-invariant constant addend into v_mov_b64 (copy the constant) + v_fmac_f64: one more vector instruction per
// step, three per logarithm, in kernels whose bound is the number of vector instructions issued.
__host__ __device__ __forceinline__ double smx_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
#else
    return __builtin_fma(a, b, c);
#endif
}
// The device table is staged with inv_c DOUBLED: the device form below takes the mantissa from v_frexp_mant_f64 -- in
// [1/2, 1) -- instead of assembling the [1, 2) one from three bit operations; (m/2)(2 inv_c) - 1 is the same fma bit for bit.
__device__ __forceinline__ void smx_log_stage(double2 *tab) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) {
        double2 e = ((const double2 *)smx_log_table)[i];
        e.x *= 2.0;
        tab[i] = e;
    }
}
// x a positive normal number?  (one v_cmp_class_f64)
__host__ __device__ __forceinline__ bool smx_log_fast_ok(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_class(x, 0x100);
#else
    return (unsigned)(smx_hi(x) - 0x00100000) < 0x7fe00000u;
#endif
}
// ln x for a positive normal x, no guard (anything else: a finite or non-finite number of no meaning, never a fault --
// callers test smx_log_fast_ok, in a loop typically by or-ing the test into a flag and redoing the rare flagged item with
// smx_log).  15 vector instructions on the device.
__host__ __device__ __forceinline__ double smx_log_fast(double x, const double2 *tab) {
    const int hx = smx_hi(x);
    const double2 e = tab[(hx >> 13) & 0x7f];
#if defined(__HIP_DEVICE_COMPILE__)
    const double f = __builtin_fma(__builtin_amdgcn_frexp_mant(x), e.x, -1.0);
    const double dk = (double)(__builtin_amdgcn_frexp_exp(x) - 1);
#else
    const double mnt = smx_with_hi(x, (hx & 0x000fffff) | 0x3ff00000);
    const double f = __builtin_fma(mnt, e.x, -1.0);
    const double dk = (double)((hx >> 20) - 1023);
#endif
    double third = 1.0 / 3.0;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(third));            // (its own register pair: 1/3 and -1/6 share their low word, and the compiler rebuilds one of the pairs per call)
#endif
    double p = smx_fma(f, -1.0 / 6.0, 0.2);
    p = smx_fma(f, p, -0.25);
    p = smx_fma(f, p, third);
    p = smx_fma(f, p, -0.5);
    p = smx_fma(f, p, 1.0);
    return __builtin_fma(f, p, __builtin_fma(dk, 6.931471805599453094e-01, e.y));
}
__host__ __device__ __forceinline__ double smx_log(double x, const double2 *tab) {
    if (!smx_log_fast_ok(x)) return log(x);
    return smx_log_fast(x, tab);
}
