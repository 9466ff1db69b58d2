// Probe (not product): how fast does the chip start and retire waves that do nothing, as a function of the VGPR
// allocation and the LDS allocation per workgroup?  (K2's sky tiles are short waves: is their phase launch-rate bound?)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V> __global__ __launch_bounds__(256) void empty(int *o)
{
    extern __shared__ int lds[];
    if (V == 1) asm volatile("v_mov_b32 v120, 0" ::: "v120");     // forces a 121+ VGPR allocation
    if (o == (int *)1) { lds[threadIdx.x] = 1; o[0] = lds[0]; }
}
int main()
{
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grids[] = { 8100, 32400 };
    for (int v = 0; v < 2; v++)
        for (int lds : { 0, 18432 })
            for (int wpb : { 4, 1 })
                for (int gi = 0; gi < 1; gi++) {
                    const int nwaves = 32400, g = nwaves / wpb;
                    float ms = 0;
                    for (int rep = 0; rep < 2; rep++) {
                        (void)hipEventRecord(e0);
                        for (int it = 0; it < 20; it++) {
                            if (v) hipLaunchKernelGGL(empty<1>, dim3(g), dim3(64 * wpb), lds / 4 * wpb, 0, (int *)nullptr);
                            else hipLaunchKernelGGL(empty<0>, dim3(g), dim3(64 * wpb), lds / 4 * wpb, 0, (int *)nullptr);
                        }
                        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                        (void)hipEventElapsedTime(&ms, e0, e1);
                    }
                    printf("vgpr %s lds/wg %5d waves/wg %d: %.2f us per launch of %d waves\n", v ? "128" : "min", lds / 4 * wpb, wpb, ms / 20 * 1e3, nwaves);
                }
    (void)grids;
    return 0;
}
