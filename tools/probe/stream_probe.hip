// Probe (not product): what HBM rate does K1's access pattern reach with no arithmetic?  4 dword planes in, float4 out.
//   v0: one pixel per lane (4 x 256-byte row segments per wave, 1 KiB store)       v1: two rows per lane
//   v2: four consecutive pixels per lane (dwordx4 loads, four float4 stores)        v3: plain float4 copy 133 MB -> 133 MB
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define W 3840
#define H 2160
__global__ __launch_bounds__(256) void v0(const unsigned *a, const unsigned *b, const unsigned *c, const float *d, v4f *o)
{
    unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    size_t i = (size_t)y * W + x;
    unsigned ua = __builtin_nontemporal_load(a + i), ub = __builtin_nontemporal_load(b + i), uc = __builtin_nontemporal_load(c + i);
    float fd = __builtin_nontemporal_load(d + i);
    __builtin_nontemporal_store(v4f{ (float)ua, (float)ub, (float)uc, fd }, o + i);
}
__global__ __launch_bounds__(256) void v1(const unsigned *a, const unsigned *b, const unsigned *c, const float *d, v4f *o)
{
    unsigned x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 8 + threadIdx.y;
    size_t i = (size_t)y * W + x, j = i + 4 * W;
    unsigned ua = __builtin_nontemporal_load(a + i), ub = __builtin_nontemporal_load(b + i), uc = __builtin_nontemporal_load(c + i);
    float fd = __builtin_nontemporal_load(d + i);
    unsigned va = __builtin_nontemporal_load(a + j), vb = __builtin_nontemporal_load(b + j), vc = __builtin_nontemporal_load(c + j);
    float gd = __builtin_nontemporal_load(d + j);
    __builtin_nontemporal_store(v4f{ (float)ua, (float)ub, (float)uc, fd }, o + i);
    __builtin_nontemporal_store(v4f{ (float)va, (float)vb, (float)vc, gd }, o + j);
}
__global__ __launch_bounds__(256) void v2(const v4u *a, const v4u *b, const v4u *c, const v4f *d, v4f *o)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // group of 4 pixels
    v4u ua = __builtin_nontemporal_load(a + i), ub = __builtin_nontemporal_load(b + i), uc = __builtin_nontemporal_load(c + i);
    v4f fd = __builtin_nontemporal_load(d + i);
    __builtin_nontemporal_store(v4f{ (float)ua.x, (float)ub.x, (float)uc.x, fd.x }, o + 4 * i);
    __builtin_nontemporal_store(v4f{ (float)ua.y, (float)ub.y, (float)uc.y, fd.y }, o + 4 * i + 1);
    __builtin_nontemporal_store(v4f{ (float)ua.z, (float)ub.z, (float)uc.z, fd.z }, o + 4 * i + 2);
    __builtin_nontemporal_store(v4f{ (float)ua.w, (float)ub.w, (float)uc.w, fd.w }, o + 4 * i + 3);
}
__global__ __launch_bounds__(256) void v3(const v4f *a, v4f *o)
{
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    __builtin_nontemporal_store(__builtin_nontemporal_load(a + i), o + i);
}
int main()
{
    const size_t px = (size_t)W * H;
    const int SETS = 3;
    unsigned *a[SETS], *b[SETS], *c[SETS]; float *d[SETS]; v4f *o[SETS], *src[SETS];
    for (int s = 0; s < SETS; s++) {
        hipMalloc(&a[s], px * 4); hipMalloc(&b[s], px * 4); hipMalloc(&c[s], px * 4); hipMalloc(&d[s], px * 4);
        hipMalloc(&o[s], px * 16); hipMalloc(&src[s], px * 16);
        hipMemset(a[s], 1, px * 4); hipMemset(b[s], 2, px * 4); hipMemset(c[s], 3, px * 4); hipMemset(d[s], 0, px * 4); hipMemset(src[s], 0, px * 16);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < 4; v++) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            for (int it = 0; it < 30; it++) {
                int s = it % SETS;
                if (v == 0) hipLaunchKernelGGL(v0, dim3(W / 64, H / 4), dim3(64, 4), 0, 0, a[s], b[s], c[s], d[s], o[s]);
                if (v == 1) hipLaunchKernelGGL(v1, dim3(W / 64, H / 8), dim3(64, 4), 0, 0, a[s], b[s], c[s], d[s], o[s]);
                if (v == 2) hipLaunchKernelGGL(v2, dim3(px / 4 / 256), dim3(256), 0, 0, (v4u *)a[s], (v4u *)b[s], (v4u *)c[s], (v4f *)d[s], o[s]);
                if (v == 3) hipLaunchKernelGGL(v3, dim3(px / 256), dim3(256), 0, 0, src[s], o[s]);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        printf("v%d: %.1f us  %.2f TB/s\n", v, ms / 30 * 1e3, px * 32.0 / (ms / 30 * 1e-3) / 1e12);
    }
    return 0;
}
