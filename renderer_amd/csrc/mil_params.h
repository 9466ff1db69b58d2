// Kernel-argument block shared by the host runtime (mil_api.cpp) and the two
// device translation units (mil_fast.hip / mil_strict.hip).  Passed BY VALUE in the
// kernarg segment: the 272-byte FrameMatrixData and the 12-byte push constants of
// the reference draw (render/mod.rs:201-216) never need a device-side copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mil {

#define MIL_DIR_INLINE 4u
#define MIL_MAX_ROW_ORDER 288u   // tile rows of a 4608-row frame; taller frames use plain order
#define MIL_WS_BUCKETS 32u       // cost buckets of the two-kernel tiled path
#define MIL_WS_SHARDS 64u        // counter shards per bucket
#define MIL_WS_LIST_MAX 960u     // entries a half-tile list holds at most (k_bin builds 32 of them in LDS: 60 KB, two workgroups per CU); a half tile more lights reach is shaded by brute force
#define MIL_WS_POOL_PARTS 8u      // partitions of the list pool, one counter word (own 128-byte line) each: a workgroup asks its XCD's first
#define MIL_WS_BRUTE 0x80000000u
#define MIL_WS_NONE 0xFFFFFFFFu     // meta of a half tile that is not an item (no lit pixel, or outside the launch)

struct ShadeParams {
    // G-buffer planes (device), see mil_gbuffer
    const uint32_t *normal, *albedo, *rm;
    const float *depth;
    uint32_t width, height, pitch, y_offset;
    // Band dealing (multi-GPU, mil_band_layout): the planes hold every n-th band of band_rows = 1 << band_shift rows of the
    // frame, compacted.  global row = local + y_offset + (local >> band_shift) * band_extra, band_extra = (n - 1) * band_rows
    // (y_offset carries rank * band_rows).  Off: band_shift = band_extra = 0 and the formula is the strip's local + y_offset.
    // The output is indexed by local rows, or (out_band_extra = band_extra) by rows of the assembled frame.
    uint32_t band_shift, band_extra, out_band_extra;
    // FrameMatrixData subset read by the lighting shader (fs_deferred_lighting.glsl:34-36,59)
    float inv_proj[16], inv_view[16], cam[3];
    float res_x, res_y;     // push_constants.resolution
    float sx, ox, sy, oy;   // clip.xy = (x, y_global) * s + o  ==  ((x+0.5)/res)*2-1
    float pos_m[16];        // fast path: rows 0-2 = inv_view * inv_proj * pixel->clip, row 3 = w row of inv_proj * pixel->clip
    // staged light arrays (device): each record = 2 float4
    uint32_t n_dir, n_point;
    float dir_inline[MIL_DIR_INLINE][8]; // first records in this launch's mode layout (raw if strict, else prepared)
    const float4 *dir_raw;    // {direction.xyz, intensity} {color.xyz, pad}               (strict)
    const float4 *dir_prep;   // {direction.xyz, 0} {color.xyz * intensity, 0}             (fast)
    const float4 *point_raw;  // {position.xyz, radius} {color.xyz, intensity}             (strict)
    const float4 *point_prep; // {position.xyz, radius^2} {color.xyz * intensity, 1/r^2}   (fast)
    const float4 *point_cull; // {position.xyz, radius}, 16 bytes per light                 (tile culling, both modes)
    uint32_t rows_per_block;  // K1: pixels per lane (1 or 4; tuning)
    uint32_t *out2;           // HDR launches: also the tonemapped sRGB8 image of the same pixels (mil_extras.ldr_out), or null
    uint32_t out2_pitch;
    uint32_t tune;            // experiment bits (env MIL_TUNE): 1 = K2 tile rows in plain order, 2/4 = 1/2 waves per workgroup
    float ambient[3];
    uint32_t has_ambient;
    // shadow extension
    uint32_t has_shadow, shadow_size, shadow_lds;
    float shadow_bias;
    float lvp[16];
    const float *shadow_map;
    // output
    void *out;
    uint32_t out_pitch;
    uint32_t tiles_x, tiles_y;
    uint16_t row_order[MIL_MAX_ROW_ORDER]; // K2 visits tile row row_order[i] i-th (a permutation; host heuristic, speed only)
    uint32_t *tile_counts;      // optional per-tile list length (diagnostics)
    uint32_t *tile_stamps;      // optional 4 x u32 per tile: start, end (100 MHz ticks), list length, HW_ID (diagnostics)
    const uint8_t *lut11, *lut10; // sRGB8 of every R11G11B10F channel code <= 1.0
    // Two-kernel tiled path (k_bin -> k_order -> k_shade; fast mode): the context's workspace (mil_internal.h TileWorkspace).
    // An ITEM is a 16x8 half of a 16x16 screen tile: item = 2 * (tile_y * tiles_x + tile_x) + half.
    // k_bin builds every lit half tile's light list in LDS, takes ONE returning atomic per 64x64 macro tile on ws_pool_used
    // for the 32 lists of the macro tile together and copies them to the pool back to back: the lists are compact (a 4K
    // frame with 4096 lights holds ~2 M entries = 4 MB; rounds 1-2 reserved 2 KiB per half tile = 133 MB).  It also counts
    // the item in the histogram [cost bucket][shard] and keeps the count it found there as the item's RANK in that segment
    // (a returning atomic; one counter word serves ~88 of them per microsecond, so the words are sharded by tile %
    // MIL_WS_SHARDS; bucket 0 = most expensive).  k_order turns the histogram into segment starts and writes every item to
    // ws_order[start + rank] - no atomics: one launch order, most expensive half tiles first, which k_shade walks.
    uint32_t *ws_hist;              // [MIL_WS_SHARDS][MIL_WS_BUCKETS] items per (shard, bucket)     } zeroed before k_bin
    uint32_t *ws_total;             // number of listed items                                         } by one memset
    uint32_t *ws_pool_used;         // [MIL_WS_POOL_PARTS] x 32 words apart: entries handed out per partition; then brute count }
    uint32_t *ws_order;             // [ws_items_cap] item ids, most expensive first
    uint32_t *ws_rank;              // [ws_items_cap] the item's rank within its (bucket, shard) segment
    uint2 *ws_meta;                 // [ws_items_cap] { list length | bucket << 16 | MIL_WS_BRUTE, or MIL_WS_NONE ; offset of the list in ws_pool }
    uint16_t *ws_pool;              // [ws_pool_cap] light indices, every list ascending
    uint32_t ws_items_cap, ws_list_cap, ws_pool_cap;    // ws_list_cap <= MIL_WS_LIST_MAX: a longer list marks the half tile for brute force
    // whole-frame launch (mil_shade_frame): the sky and transparency-resolve subpasses folded into the store
    uint32_t has_sky;
    uint32_t accum_pitch, reveal_pitch;
    uint32_t resolve_dma;             // K2 may fetch accum / reveal tiles by dword LDS-DMA (even width and pitch, 4-byte aligned base)
    const uint16_t *accum, *reveal;   // RGBA16F / R16F; accum NULL = no resolve
    float hw[11][4];                  // HosekWilkieParams: A B C D E F G H I Z sun_direction
};

struct TonemapParams {
    const float4 *hdr;
    uint32_t *out;
    uint32_t width, height, hdr_pitch, out_pitch;
    const uint8_t *lut11, *lut10;
    uint32_t rows;              // rows per lane: 1, 2 or 4 (tuning; 0 = default)
};

struct FxaaParams {
    const uint32_t *ldr;        // whole frame, B10G11R11_UFLOAT_PACK32
    uint32_t *out;              // rows [row_begin, row_begin + row_count), B8G8R8A8_SRGB
    uint32_t width, height, ldr_pitch, out_pitch, row_begin, row_count;
    const uint8_t *lut11, *lut10; // sRGB8 of every R11G11B10F channel code <= 1.0
    uint32_t rows;              // fast kernel: output rows per lane, 2 or 4 (tuning; 0 = default)
};

struct SkyParams {            // sky subpass: depth in, hdr in/out
    const float *depth;
    float *hdr;
    uint32_t width, height, depth_pitch, hdr_pitch, y_offset;
    float inv_proj[16], inv_view[16], cam[3];
    float res_x, res_y;
    float pos_m[16];          // fast path: folded camera-relative position matrix (as ShadeParams.pos_m)
    float hw[11][4];          // A B C D E F G H I Z sun_direction
};

struct ResolveParams {        // transparency resolve: accum RGBA16F + reveal R16F in, hdr in/out
    const uint16_t *accum, *reveal;
    float *hdr;
    uint32_t width, height, accum_pitch, reveal_pitch, hdr_pitch;
};

struct AccumParams {          // transparency accumulate over a fragment list (fs_mcguire13_accumulation.glsl:90-127)
    const float4 *pos_z, *nrm_op, *alb_rough;   // see mil_transparent_fragments
    const float *metallic;
    float4 *accum;
    float *reveal;
    uint32_t count, n_dir;
    float cam[3];
    float dir_inline[MIL_DIR_INLINE][8];        // RAW records {direction, intensity} {color, pad} in both modes
    const float4 *dir_raw;                      // (the shader does not use the intensity)
};

// implemented once per numeric mode; every launcher returns the status of ITS OWN launch (hipLaunchKernel), not the
// thread's sticky error state
namespace fast {
hipError_t launch_stream(int fmt, const ShadeParams &p, bool frame, hipStream_t s);
hipError_t launch_tiled(int fmt, const ShadeParams &p, bool frame, hipStream_t s);
hipError_t launch_tonemap(int fmt, const TonemapParams &p, hipStream_t s);
hipError_t launch_count_pairs(const ShadeParams &p, unsigned long long *count, hipStream_t s);
hipError_t launch_bin(int fmt, const ShadeParams &p, hipStream_t s);                 // k_bin: classify, cull, bucket (then:)
hipError_t launch_order(const ShadeParams &p, hipStream_t s);                        // k_order: buckets -> one launch order
hipError_t launch_shade(int fmt, const ShadeParams &p, bool frame, hipStream_t s);               // k_shade: one wave per listed half tile
hipError_t launch_fxaa(const FxaaParams &p, hipStream_t s);
hipError_t launch_sky(const SkyParams &p, hipStream_t s);
hipError_t launch_resolve(const ResolveParams &p, hipStream_t s);
hipError_t launch_accumulate(const AccumParams &p, hipStream_t s);
}
namespace strict {
hipError_t launch_stream(int fmt, const ShadeParams &p, bool frame, hipStream_t s);
hipError_t launch_tiled(int fmt, const ShadeParams &p, bool frame, hipStream_t s);
hipError_t launch_tonemap(int fmt, const TonemapParams &p, hipStream_t s);
hipError_t launch_fxaa(const FxaaParams &p, hipStream_t s);
hipError_t launch_sky(const SkyParams &p, hipStream_t s);
hipError_t launch_resolve(const ResolveParams &p, hipStream_t s);
hipError_t launch_accumulate(const AccumParams &p, hipStream_t s);
}

} // namespace mil
