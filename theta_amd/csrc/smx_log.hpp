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
__device__ __forceinline__ void smx_log_stage(double2 *tab) {
    for (int i = threadIdx.x; i < 128; i += blockDim.x) tab[i] = ((const double2 *)smx_log_table)[i];
}
__host__ __device__ __forceinline__ double smx_log(double x, const double2 *tab) {
    const int hx = smx_hi(x);
    if ((unsigned)(hx - 0x00100000) >= 0x7fe00000u) return log(x);
    const double2 e = tab[(hx >> 13) & 0x7f];
    const double mnt = smx_with_hi(x, (hx & 0x000fffff) | 0x3ff00000);
    const double f = __builtin_fma(mnt, e.x, -1.0);
    const double dk = (double)((hx >> 20) - 1023);
    double p = __builtin_fma(f, -1.0 / 6.0, 0.2);
    p = __builtin_fma(f, p, -0.25);
    p = __builtin_fma(f, p, 1.0 / 3.0);
    p = __builtin_fma(f, p, -0.5);
    p = __builtin_fma(f, p, 1.0);
    return __builtin_fma(f, p, __builtin_fma(dk, 6.931471805599453094e-01, e.y));
}
