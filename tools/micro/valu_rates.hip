// Issue-rate microbenchmark for the VALU instructions the search kernels lean on (gfx950).
// hipcc -O3 --offload-arch=gfx950 tools/micro/valu_rates.hip -o build_ab/valu_rates && build_ab/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
#define ITERS 4096
#define UNR 8
template <int OP>
__global__ void k(double *out, double seed) {
    double d[UNR];
    float f[UNR];
    v2f p[UNR];
    for (int i = 0; i < UNR; i++) {
        d[i] = seed + i + threadIdx.x * 1e-3;
        f[i] = (float)d[i];
        p[i] = v2f{f[i], f[i] + 1.f};
    }
    const double cd = seed * 0.5;
    const float cf = (float)cd;
    const v2f cp = {cf, cf};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int i = 0; i < UNR; i++) {
            if (OP == 0) d[i] = __builtin_fma(d[i], cd, cd);
            if (OP == 1) f[i] = __builtin_fmaf(f[i], cf, cf);
            if (OP == 2) p[i] = __builtin_elementwise_fma(p[i], cp, cp);
            if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[i]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
            if (OP == 5) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
            if (OP == 6) d[i] = d[i] * cd;
            if (OP == 7) d[i] = d[i] + cd;
            if (OP == 8) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(cf));
            if (OP == 9) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            if (OP == 11) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[i]));
            if (OP == 12) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 13) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 14) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(d[i]));
            if (OP == 15) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 16) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(d[i]), "v"(f[i]) : "vcc");
            if (OP == 17) asm volatile("v_bfe_u32 %0, %0, 13, 7" : "+v"(f[i]));
            if (OP == 18) asm volatile("v_lshl_add_u32 %0, %0, 4, %1" : "+v"(f[i]) : "v"(cf));
            if (OP == 19) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(f[i]));
            if (OP == 20) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 21) asm volatile("v_mov_b64 %0, %1" : "=v"(d[i]) : "v"(cd));
            if (OP == 22) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[i]) : "v"(cf));
            if (OP == 23) asm volatile("v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(f[i]) : "v"(cf));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < UNR; i++) s += d[i] + f[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / (double)(ITERS * UNR);
}
template <int OP>
void run(const char *name) {
    double *o;
    hipMalloc(&o, 1 << 24);
    for (int wpb = 1; wpb <= 16; wpb *= 2) {       // waves per block; one block per CU (256 blocks)
        if (wpb != 4 && wpb != 8 && wpb != 16) continue;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * wpb), 0, 0, o, 1.0001);
        hipDeviceSynchronize();
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(64 * wpb), 0, 0, o, 1.0001);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        double h;
        hipMemcpy(&h, o, 8, hipMemcpyDeviceToHost);
        // instructions per SIMD = waves/SIMD * ITERS*UNR ; cycles at 2.4 GHz
        double wps = wpb / 4.0;
        double ns_per_inst_simd = ms * 1e6 / (wps * ITERS * UNR);
        printf("%-14s waves/SIMD %4.1f : %.2f ns per wave-instruction per SIMD (%.1f cycles @2.4GHz); in-wave counter %.1f ticks/inst\n",
               name, wps, ns_per_inst_simd, ns_per_inst_simd * 2.4, h);
    }
    hipFree(o);
}
int main() {
    run<0>("v_fma_f64");
    run<6>("v_mul_f64");
    run<7>("v_add_f64");
    run<1>("v_fma_f32");
    run<2>("v_pk_fma_f32");
    run<3>("v_rcp_f64");
    run<4>("v_rcp_f32");
    run<5>("v_log_f32");
    run<8>("v_min3_f32");
    run<9>("v_cvt_f32_f64");
    run<10>("v_and_b32");
    run<11>("v_lshlrev_b64");
    run<12>("v_cvt_f64_u32");
    run<13>("v_cvt_f64_i32");
    run<14>("v_frexp_mant_f64");
    run<15>("v_frexp_exp_i32_f64");
    run<16>("v_cmp_class_f64");
    run<17>("v_bfe_u32");
    run<18>("v_lshl_add_u32");
    run<19>("v_cvt_f32_ubyte1");
    run<20>("v_cvt_f64_f32");
    run<21>("v_mov_b64");
    run<22>("v_add_u32");
    run<23>("v_and_b32_sdwa");
    return 0;
}
