// mil_light_ex_noalloc: the one entry point of libmil.so that allocates nothing (include/mil.h).  It has no counterpart in
// the reference; it exists for callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops,
// harnesses that forbid allocation.
//
// mil_light_ex stages the light arrays (raw + prepared + cull records, mil_internal.h Slot) through a ring of pinned-host /
// device buffers that mil_create allocates, with events and a copy stream.  Here the point lights are already in DEVICE
// memory, the prepared records are derived by a kernel on the caller's stream, and the staging buffer is a file-scope
// __device__ array: `scratch` below, 334 080 bytes, held by every GPU the library runs on until the process ends.
// Every call on a GPU shares that GPU's array and overwrites the records it reads (nothing has to start at zero):
// calls that overlap on one GPU (two streams, two threads) race - serialise them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "mil_internal.h"

namespace {

constexpr size_t DIR_B = (size_t)MIL_MAX_DIR_LIGHTS * 32;
constexpr size_t PT_B = (size_t)MIL_NOALLOC_MAX_POINT_LIGHTS * 32;
constexpr size_t SCRATCH_BYTES = 2 * DIR_B + (size_t)MIL_NOALLOC_MAX_POINT_LIGHTS * (32 + 32 + 16);   // the Slot layout
static_assert(SCRATCH_BYTES == 334080 && SCRATCH_BYTES % 256 == 0, "stated size of the static scratch array");

// In host code this name is only a stand-in for the symbol: the device address comes from hipGetSymbolAddress, per device.
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];

struct DirBlock { float rec[MIL_MAX_DIR_LIGHTS][8]; };      // the directional records travel by value in the kernarg (3200 B)

// One thread per light: the same arithmetic as the host staging of mil_api.cpp stage_lights (one correctly rounded FP32
// operation each, so the prepared records are the same bits).
__global__ __launch_bounds__(256) void k_prepare_lights(const DirBlock dirs, uint32_t n_dir, const float4 *point_lights, uint32_t n_point,
                                                        unsigned char *dst)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_dir) {
        const float *l = dirs.rec[i];       // {direction.xyz, intensity, color.xyz, pad}
        float4 *raw = (float4 *)dst + 2 * i, *prep = (float4 *)(dst + DIR_B) + 2 * i;
        raw[0] = make_float4(l[0], l[1], l[2], l[3]);
        raw[1] = make_float4(l[4], l[5], l[6], 0.0f);
        prep[0] = make_float4(l[0], l[1], l[2], 0.0f);
        prep[1] = make_float4(__fmul_rn(l[4], l[3]), __fmul_rn(l[5], l[3]), __fmul_rn(l[6], l[3]), 0.0f);
    }
    if (i < n_point) {
        const float4 a = point_lights[2 * i], b = point_lights[2 * i + 1];   // {position.xyz, radius} {color.xyz, intensity}
        float4 *raw = (float4 *)(dst + 2 * DIR_B) + 2 * i, *prep = (float4 *)(dst + 2 * DIR_B + PT_B) + 2 * i;
        float4 *cull = (float4 *)(dst + 2 * DIR_B + 2 * PT_B) + i;
        raw[0] = a; raw[1] = b;
        const float r2 = __fmul_rn(a.w, a.w);
        prep[0] = make_float4(a.x, a.y, a.z, r2);
        prep[1] = make_float4(__fmul_rn(b.x, b.w), __fmul_rn(b.y, b.w), __fmul_rn(b.z, b.w), __fdiv_rn(1.0f, r2));
        *cull = a;
    }
}

// Test hook of the multi-GPU layer (mil_mg_debug_delay_gather): one wave that sleeps until `ticks` of the 100 MHz counter
// have passed - an exit condition it reaches by itself; the host caps the time at 100 ms.
__global__ __launch_bounds__(64) void k_delay(unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}

} // namespace

hipError_t milhost::launch_delay(hipStream_t s, uint32_t microseconds)
{
    unsigned long long ticks = 100ull * (microseconds > 100000u ? 100000u : microseconds);
    void *args[] = { (void *)&ticks };
    return hipLaunchKernel((const void *)k_delay, dim3(1), dim3(64), args, 0, s);
}

// mil_create, once per context, with the context's device current: each GPU holds its own copy of the array at its own
// address.
int milhost::noalloc_init(mil_ctx *c)
{
    unsigned char *ptr = nullptr;
    const hipError_t e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !ptr) return hip_fail(c, e == hipSuccess ? hipErrorInvalidSymbol : e);
    c->static_scratch = ptr;
    return MIL_OK;
}

extern "C" int mil_light_ex_noalloc(mil_ctx *c, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                                    const mil_directional_light *lights, const mil_push_constants *push,
                                    const mil_extras *ex, void *out, uint32_t out_pitch_px, mil_out_format fmt,
                                    uint32_t flags, void *stream)
{
    using namespace milhost;
    if (!c || !c->static_scratch) return MIL_ERR_BAD_ARG;
    const uint32_t n_point = ex ? ex->n_point : 0;
    if (n_point > MIL_NOALLOC_MAX_POINT_LIGHTS) return MIL_ERR_CAPACITY;          // the static array holds no more
    const PreStaged pre = { c->static_scratch, MIL_NOALLOC_MAX_POINT_LIGHTS };
    int rc = shade_check(c, gb, frame, lights, push, ex, nullptr, out, out_pitch_px, fmt, nullptr, false, &pre);
    if (rc) return rc;                                                            // nothing was enqueued
    if ((rc = set_device(c))) return rc;
    const uint32_t n_dir = push->light_count;
    if (n_dir > MIL_DIR_INLINE || n_point > 0) {                                  // else every record travels in the kernarg
        DirBlock dirs;
        memcpy(dirs.rec, lights, (size_t)n_dir * 32);
        const float4 *pl = (const float4 *)(ex ? ex->point_lights : nullptr);
        unsigned char *dst = c->static_scratch;                                   // the device address, never the array's name
        void *args[] = { (void *)&dirs, (void *)&n_dir, (void *)&pl, (void *)&n_point, (void *)&dst };
        const uint32_t n = n_dir > n_point ? n_dir : n_point;
        const hipError_t e = hipLaunchKernel((const void *)k_prepare_lights, dim3((n + 255) / 256), dim3(256), args, 0, (hipStream_t)stream);
        if (e != hipSuccess) return hip_fail(c, e);
    }
    return shade(c, gb, frame, lights, push, ex, nullptr, out, out_pitch_px, fmt, flags, stream, nullptr, false, &pre);
}
