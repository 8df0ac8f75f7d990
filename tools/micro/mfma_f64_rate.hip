// Rate of v_mfma_f64_16x16x4_f64 on gfx950: NACC independent accumulators per wave, WPS waves per SIMD, all CUs.
// Also: the same loop with f64 VALU FMAs interleaved (do MFMA-f64 and VALU-f64 co-execute?).
// Measured on MI355X (2.4 GHz nominal): 76-80 nominal cycles per MFMA alone (66 TFLOP/s = 84 % of the 78.6 spec), +6.5-7 cycles
// per interleaved f64 FMA, +10 per v_and_b32: vector instructions do NOT co-execute with the f64 MFMA of another wave.
// hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f64_rate.hip -o build_ab/mfma_f64_rate && build_ab/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 2048
template <int NACC, int NVALU, int NIALU>
__global__ __launch_bounds__(256) void k(double *out, double seed) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = d4{seed, seed, seed, seed};
    double a = seed + threadIdx.x * 1e-3, b = seed * 0.5;
    double v[8];
    int iv[8];
    for (int i = 0; i < 8; i++) { v[i] = seed + i; iv[i] = threadIdx.x + i; }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NVALU; j++) v[j] = __builtin_fma(v[j], b, b);
#pragma unroll
            for (int j = 0; j < NIALU; j++) asm volatile("v_and_b32 %0, %0, %1" : "+v"(iv[j]) : "v"(iv[(j + 1) & 7]));
        }
    }
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; i++) s += v[i] + iv[i];
    if (s == 12345.678) out[0] = s;
}
template <int NACC, int NVALU, int NIALU>
void run(const char *name, int wps) {
    double *out;
    hipMalloc(&out, 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    dim3 grid(256 * wps), block(256);
    hipLaunchKernelGGL((k<NACC, NVALU, NIALU>), grid, block, 0, 0, out, 1.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, NVALU, NIALU>), grid, block, 0, 0, out, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)grid.x * 4 * ITERS * NACC;
    printf("%-34s waves/SIMD %d: %.3f ms  %.1f TFLOP/s  (%.1f SIMD-cycles per MFMA at 2.4 GHz)\n", name, wps, ms,
           mf * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / mf);
    hipFree(out);
}
int main() {
    run<1, 0, 0>("1 acc (dependent chain)", 1);
    run<1, 0, 0>("1 acc (dependent chain)", 4);
    run<4, 0, 0>("4 acc", 1);
    run<4, 0, 0>("4 acc", 2);
    run<4, 4, 0>("4 acc + 4 f64 FMA per MFMA", 2);
    run<4, 8, 0>("4 acc + 8 f64 FMA per MFMA", 2);
    run<4, 0, 8>("4 acc + 8 int VALU per MFMA", 2);
    return 0;
}
