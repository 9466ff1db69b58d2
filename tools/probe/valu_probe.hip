// VALU issue-rate probe for gfx950: wave-instructions per cycle per SIMD for v_fma_f32,
// v_pk_fma_f32 and v_rcp_f32 at 1..8 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    const float m = 1.0001f, c = 0.5f;
    const v2f m2 = {m, m}, c2 = {c, c};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                             "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));
            }
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                             "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        } else {   // dependent chain of v_fma_f32
#pragma unroll
            for (int u = 0; u < 64; u++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y;
}
template <int MODE> void run(const char *name, float *out)
{
    const int iters = 4000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        dim3 grid(256 * wps), block(256);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double inst_per_wave = (double)iters * 64, cycles = ms * 1e-3 * 2.4e9;
        printf("%-14s waves/SIMD %d: %8.3f ms  %6.2f cycles per wave-instr per SIMD (at 2.4 GHz)\n", name, wps, ms,
               cycles / (inst_per_wave * wps));
    }
}
int main()
{
    float *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    run<0>("v_fma_f32", out);
    run<1>("v_pk_fma_f32", out);
    run<2>("v_rcp_f32", out);
    run<3>("fma dep chain", out);
    hipFree(out);
    return 0;
}
