/*
 * mil.h — C ABI of libmil.so, the MI355X-native lighting + tonemap pass for
 * dobrakmato/renderer's deferred path.
 *
 * The reference has no plugin / operator / FFI interface for this path: the pass
 * is the inline `draw_indexed` at renderer/src/render/mod.rs:201-216 (lighting)
 * and :297-309 (tonemap).  The boundary below is therefore defined from that
 * draw's Vulkan bindings; every struct is byte-identical to what the reference
 * binds, so the Rust side can pass `&FrameMatrixData` / `&[DirectionalLight; N]`
 * unchanged (binding stub: INTEGRATION.md).  All paths are relative to
 * /root/reference/.
 *
 * Conventions: every image pointer is DEVICE memory owned by the caller; every
 * uniform struct / light array pointer is HOST memory, copied during the call
 * (mirrors the reference's per-frame CpuBufferPool ring, render/pools.rs:42-57).
 * Calls enqueue work on `stream` (a hipStream_t, 0 = default stream) and return;
 * no per-frame call allocates after mil_create (the external-memory set-up calls below do).  One context = one thread at a time;
 * calls of one context may name different streams (the staged light copies are tracked per stream: up to four streams
 * per light set run unthrottled, a fifth waits for the oldest).
 * Every call makes the context's device the calling thread's current HIP device and leaves it so.
 * All functions return 0 on success or a negative mil_status; nothing aborts.
 */
#ifndef MIL_H
#define MIL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIL_VERSION 3
#define MIL_MAX_DIR_LIGHTS 100u /* MAX_LIGHTS, renderer/shaders/inc_structs.glsl:1; pool render/pbr/mod.rs:36 */
#define MIL_TILE 16u            /* screen tile edge of the culled path */

typedef enum mil_status {
    MIL_OK = 0,
    MIL_ERR_BAD_ARG = -1,         /* null pointer, zero size, pitch < width, bad enum */
    MIL_ERR_TOO_MANY_LIGHTS = -2, /* the reference panics here (render/mod.rs:197-199); we return */
    MIL_ERR_HIP = -3,             /* a HIP runtime call failed; mil_last_hip_error() has the code */
    MIL_ERR_NO_DEVICE = -4,
    MIL_ERR_CAPACITY = -5,        /* more point lights than the context was created for */
    MIL_ERR_RCCL = -6             /* an RCCL call failed or librccl is missing; mil_mg_last_rccl_error() has the code */
} mil_status;

/* FrameMatrixData — renderer/src/render/ubo.rs:28-41; GLSL block
 * shaders/fs_deferred_lighting.glsl:16-22.  std140, 272 bytes, mat4 column-major
 * (cgmath Matrix4 = 4 column vectors).  Filled per frame at render/mod.rs:96-104. */
typedef struct mil_frame_matrix_data {
    float view[16];
    float projection[16];
    float inv_projection[16];
    float inv_view[16];
    float camera_position[3];
    float _pad;
} mil_frame_matrix_data;

/* DirectionalLight — render/ubo.rs:54-63; shaders/inc_structs.glsl:12-16.
 * std140 array stride 32.  `direction` points from the pixel toward the light
 * and is NOT normalised by the shader (the CPU does, main.rs:82,87). */
typedef struct mil_directional_light {
    float direction[3];
    float intensity;
    float color[3];
    float _pad;
} mil_directional_light;

/* Push constants of the lighting draw — shaders/fs_deferred_lighting.glsl:24-27,
 * set at render/mod.rs:211-214.  12 bytes.  `resolution` is the FULL framebuffer
 * size even when a strip is shaded. */
typedef struct mil_push_constants {
    float resolution[2];
    uint32_t light_count;
} mil_push_constants;

/* The four input attachments of the lighting subpass, in binding order
 * (shaders/fs_deferred_lighting.glsl:5-8; formats render/pbr/mod.rs:150-160;
 * encoding written by shaders/fs_deferred_geometry.glsl:47-49).  One u32/f32 per
 * pixel per plane, row 0 = top.  A strip of a larger frame sets y_offset to the
 * global row of its first row; PositionFromDepth uses global coordinates. */
typedef struct mil_gbuffer {
    const uint32_t *normal_l_model;     /* A2B10G10R10_UNORM_PACK32: R[9:0] G[19:10] B[29:20] A[31:30] */
    const uint32_t *albedo_occlusion;   /* R8G8B8A8_UNORM: albedo rgb, occlusion a */
    const uint32_t *roughness_metallic; /* R8G8B8A8_UNORM: roughness r, metallic g */
    const float *depth;                 /* D32_SFLOAT, viewport depth range 0..1, cleared to 1.0 */
    uint32_t width, height;             /* of this strip */
    uint32_t pitch_px;                  /* row pitch of all four planes, in pixels (>= width) */
    uint32_t y_offset;                  /* global y of row 0 (0 for a whole frame) */
} mil_gbuffer;

/* ---- build-defined extensions (no counterpart in the reference).
 * With n_point = 0, shadow = NULL and ambient = 0 the result is bit-identical to
 * mil_light_directional. ---- */

/* E1 point light, same 32-byte stride as DirectionalLight. */
typedef struct mil_point_light {
    float position[3];
    float radius;
    float color[3];
    float intensity;
} mil_point_light;

/* E3: 3x3 PCF shadow on directional light 0.  `map` is DEVICE memory, size x size
 * floats (size <= 16384), depth 0..1 as seen through light_view_proj (column-major, no w divide). */
typedef struct mil_shadow {
    float light_view_proj[16];
    const float *map;
    uint32_t size;
    float bias;
} mil_shadow;

typedef enum mil_out_format {
    MIL_OUT_HDR_RGBA32F = 0,  /* hdr attachment, R32G32B32A32_SFLOAT (render/pbr/mod.rs:32), alpha 1 */
    MIL_OUT_LDR_R11G11B10F = 1, /* tonemapped, B10G11R11_UFLOAT_PACK32 (render/pbr/mod.rs:161-173), RNE */
    MIL_OUT_SRGB8_BGRA = 2,   /* tonemapped, sRGB OETF, B8G8R8A8 (swapchain format render/renderer.rs:78-83) */
    MIL_OUT_SRGB8_BGRA_VIA_LDR = 3 /* as the reference chain: tonemap -> R11G11B10F -> sRGB8 (no FXAA) */
} mil_out_format;

enum {
    MIL_FLAG_STRICT = 1u << 0,      /* GLSL operator order, IEEE div/sqrt, no FMA contraction: bit-exact vs the CPU oracle */
    MIL_FLAG_NO_CULL = 1u << 1,     /* point lights by brute force (every light, every pixel) */
    MIL_FLAG_SHADOW_LDS = 1u << 2   /* stage each tile's shadow-map rectangle in LDS and tap it there (default: taps
                                       straight from global memory through L1, which measured faster) */
};

typedef struct mil_extras {
    const mil_point_light *point_lights; /* HOST */
    uint32_t n_point;
    float ambient[3];                    /* E4: result += ambient * albedo * occlusion */
    const mil_shadow *shadow;            /* HOST struct, device map; NULL = off */
    /* E6: the tonemap subpass in the same launch.  With fmt == MIL_OUT_HDR_RGBA32F the launch ALSO writes, for every pixel,
     * what mil_tonemap(hdr, MIL_OUT_SRGB8_BGRA) makes of the value it stores to `out` (bit for bit: the same function of
     * the same float), to this DEVICE image: the lighting draw (render/mod.rs:201-216) and the tonemap draw
     * (render/mod.rs:297-309, fs_tonemap.glsl:14-27) as one pass over the pixels, the hdr attachment still written for
     * the subpasses between them.  Rows as `out` (strip / dealt bands / assembled frame).  NULL = off; any other fmt:
     * MIL_ERR_BAD_ARG. */
    uint32_t *ldr_out;
    uint32_t ldr_out_pitch_px;
} mil_extras;

typedef struct mil_ctx mil_ctx;

/* Context: owns a ring of staged uniform/light buffers sized at creation.  max_point_lights <= 65535. */
int mil_create(mil_ctx **out, int device, uint32_t max_point_lights);
void mil_destroy(mil_ctx *ctx);

/* Replaces the lighting draw, render/mod.rs:201-216 + shaders/fs_deferred_lighting.glsl:40-67:
 * hdr = vec4(sum_i light(N, dir_i, V, color_i, alpha, albedo, metallic) * intensity_i * occlusion, 1).
 * hdr: DEVICE, height x hdr_pitch_px x 4 floats.  push->light_count <= MIL_MAX_DIR_LIGHTS. */
int mil_light_directional(mil_ctx *ctx, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                          const mil_directional_light *lights, const mil_push_constants *push,
                          float *hdr, uint32_t hdr_pitch_px, uint32_t flags, void *stream);

/* Replaces the tonemap draw, render/mod.rs:297-309 + shaders/fs_tonemap.glsl:14-27:
 * out = encode(clamp(ACESFilm(hdr.rgb), 0, 1)).  out: DEVICE, one u32 per pixel. */
int mil_tonemap(mil_ctx *ctx, const float *hdr, uint32_t hdr_pitch_px, uint32_t width, uint32_t height,
                uint32_t *out, uint32_t out_pitch_px, mil_out_format fmt, uint32_t flags, void *stream);

/* NEXT ROW.  Replaces the FXAA draw, render/mod.rs:312-329 + shaders/fs_fxaa.glsl:22-91, sampler
 * NEAREST / CLAMP_TO_EDGE (render/fxaa/mod.rs:76-88): ldr B10G11R11_UFLOAT_PACK32 -> swapchain B8G8R8A8_SRGB
 * (render/renderer.rs:78-83).  ldr is the WHOLE frame (width x height; taps reach +-4 rows around a pixel, the
 * halo a strip-sharded caller must have gathered); rows [row_begin, row_begin + row_count) are written to out,
 * whose row 0 is row_begin.  Both DEVICE, one u32 per pixel. */
int mil_fxaa(mil_ctx *ctx, const uint32_t *ldr, uint32_t ldr_pitch_px, uint32_t width, uint32_t height,
             uint32_t row_begin, uint32_t row_count, uint32_t *out, uint32_t out_pitch_px, uint32_t flags, void *stream);

/* HosekWilkieParams - render/hosek/mod.rs:222-256; GLSL block shaders/sky_hosek_frag.glsl:14-26.  std140, 176 bytes:
 * eleven vec3 at 16-byte stride, filled on the CPU by make_hosek_wilkie_params (hosek/mod.rs:181-220, from the
 * model's dataset: NOT part of this library; the caller passes the block it already builds per frame). */
typedef struct mil_hosek_wilkie_params {
    float a[3], _p0, b[3], _p1, c[3], _p2, d[3], _p3, e[3], _p4, f[3], _p5, g[3], _p6, h[3], _p7, i[3], _p8, z[3], _p9;
    float sun_direction[3], _p10;
} mil_hosek_wilkie_params;

/* NEXT ROW.  Replaces the sky draw, render/mod.rs:221-226 + shaders/sky_hosek_vert.glsl,
 * sky_hosek_frag.glsl:28-60, as a screen-space pass: every texel still at the clear depth (the draw's LessOrEqual
 * test at z = w - 1e-5, render/hosek/mod.rs:58-64) gets hdr = (radiance * 0.05, 1); the others keep hdr.
 * depth, hdr: DEVICE (strip: y_offset as in mil_gbuffer; push->resolution = full frame). */
int mil_sky_hosek(mil_ctx *ctx, const float *depth, uint32_t depth_pitch_px, uint32_t width, uint32_t height,
                  uint32_t y_offset, const mil_frame_matrix_data *frame, const mil_push_constants *push,
                  const mil_hosek_wilkie_params *params, float *hdr, uint32_t hdr_pitch_px, uint32_t flags, void *stream);

/* NEXT ROW.  Replaces the transparency resolve draw, render/mod.rs:282-294 +
 * shaders/fs_mcguire13_resolve.glsl:8-13 with the blend state of render/mcguire13/mod.rs:108-120:
 * hdr = (accum.rgb / clamp(accum.a, 1e-4, 5e4), reveal) * (1 - reveal) + hdr * reveal.
 * accum: DEVICE R16G16B16A16_SFLOAT, reveal: DEVICE R16_SFLOAT (mcguire13/mod.rs:22-23), hdr: DEVICE, in/out. */
int mil_transparency_resolve(mil_ctx *ctx, const void *accum, uint32_t accum_pitch_px, const void *reveal,
                             uint32_t reveal_pitch_px, uint32_t width, uint32_t height, float *hdr,
                             uint32_t hdr_pitch_px, uint32_t flags, void *stream);

/* The fragments of the transparency accumulate draw (render/mod.rs:229-281), one record per rasterised fragment of
 * the translucent objects that passed the depth test, as four DEVICE planes holding what the rasteriser and the texture
 * units hand the fragment shader (shaders/fs_mcguire13_accumulation.glsl:4-8, 91-98, 123). */
typedef struct mil_transparent_fragments {
    const float *position_depth;   /* float4: in_wsPosition.xyz, gl_FragCoord.z */
    const float *normal_opacity;   /* float4: in_normal.xyz (interpolated, NOT normalised), opacity = material_data.opacity * texel (:96) */
    const float *albedo_roughness; /* float4: albedo = material_data.albedo_color * texel (:91), roughness = material_data.roughness * texel (:93) */
    const float *metallic;         /* float:  material_data.metallic * texel (:94) */
    uint32_t count;                /* < 2^30; 0 = nothing to do */
} mil_transparent_fragments;

/* NEXT ROW (second half).  The fragment shader of the transparency accumulate draw,
 * shaders/fs_mcguire13_accumulation.glsl:100-126 with specular() / diffuse() of inc_brdf.glsl:34-47, evaluated over a
 * fragment list (rasterisation, texturing and the blend stay with the caller's raster pipeline or OIT builder):
 *   lighting = sum_i (diffuse(albedo) + specular(..) * color_i * NdotL_i)     no intensity, no occlusion, roughness unclamped
 *   accum_i  = vec4(lighting * opacity, opacity) * w10(gl_FragCoord.z, opacity),   reveal_i = opacity
 * the values the blend state of render/mcguire13/mod.rs:57-84 adds into accum (ONE, ONE) and multiplies into reveal
 * (ZERO, ONE_MINUS_SRC_ALPHA).  accum: DEVICE float4 per fragment, reveal: DEVICE float per fragment.  lights / push as in
 * mil_light_directional (the same lights UBO is bound, render/mod.rs:252); only push->light_count is read. */
int mil_transparency_accumulate(mil_ctx *ctx, const mil_transparent_fragments *fragments,
                                const mil_frame_matrix_data *frame, const mil_directional_light *lights,
                                const mil_push_constants *push, float *accum, float *reveal, uint32_t flags,
                                void *stream);

/* Lighting with the extensions, optionally fused with the tonemap (valid only when
 * nothing else writes hdr in between, unlike the reference's sky/transparency
 * subpasses, render/pbr/mod.rs:420-434).  `out` is float4 per pixel for
 * MIL_OUT_HDR_RGBA32F, else one u32 per pixel; out_pitch_px in pixels.
 * Point lights are culled per 16x16 tile into LDS lists unless MIL_FLAG_NO_CULL. */
int mil_light_ex(mil_ctx *ctx, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                 const mil_directional_light *lights, const mil_push_constants *push,
                 const mil_extras *extras, void *out, uint32_t out_pitch_px, mil_out_format fmt,
                 uint32_t flags, void *stream);

/* NO COUNTERPART IN THE REFERENCE.  mil_light_ex for callers that cannot allocate: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same arguments, results bit-identical to mil_light_ex, except:
 *   - extras->point_lights is a DEVICE pointer (n_point x mil_point_light, 16-byte aligned): the lights already live in
 *     HBM; the prepared records the kernels read are derived by a small kernel on `stream`;
 *   - n_point <= MIL_NOALLOC_MAX_POINT_LIGHTS, else MIL_ERR_CAPACITY (the context's own capacity does not apply);
 *   - the staging buffer is a static __device__ array of the library (334 080 bytes per GPU, held until the process
 *     ends) instead of the context's ring.  The call makes no hipMalloc / hipFree / hipMemcpy and creates nothing.  It
 *     enqueues on `stream`: k_prepare_lights (when there are point lights or more than 4 directional ones), then ONE
 *     shading kernel (k_stream2 / k_tiled), or from the context's two-kernel threshold on (1280 point lights) a
 *     hipMemsetAsync of 8.4 KB of workspace counters + k_bin + k_order + k_shade.  Every call on one GPU shares that
 *     array: calls that OVERLAP on one GPU (two streams, two threads, two contexts) race - serialise them.  The K2 launch
 *     order is the plain one (the host never sees the lights); speed only;
 *   - the two-kernel form uses the context's workspace, which this entry point never grows (a frame beyond the reserved
 *     size - mil_create: 7680 x 4320, mil_reserve raises it - takes k_tiled).  Outside stream capture the call joins the
 *     context's cross-stream protocol like mil_light_ex (hipStreamWaitEvent / hipEventRecord on the workspace's event:
 *     neither allocates), so it may alternate with other shading calls of the same context on other streams.  UNDER
 *     CAPTURE (hipStreamBeginCapture on `stream`) no outside event may be touched: the captured launch is outside that
 *     protocol - ordering replays of the graph against other shading calls of the same context is the caller's - and the
 *     workspace is pinned from then on (the graph holds its addresses): it is never freed or moved again, later frames
 *     that would need a larger one take k_tiled, and mil_reserve / mil_set_list_pool return MIL_ERR_CAPACITY when they
 *     would have to move it. */
#define MIL_NOALLOC_MAX_POINT_LIGHTS 4096u
int mil_light_ex_noalloc(mil_ctx *ctx, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                         const mil_directional_light *lights, const mil_push_constants *push,
                         const mil_extras *extras, void *out, uint32_t out_pitch_px, mil_out_format fmt,
                         uint32_t flags, void *stream);

/* The hdr-writing subpasses that follow the lighting draw inside the reference's one render pass
 * (render/pbr/mod.rs:409-440), for mil_shade_frame.  NULL members switch a subpass off. */
typedef struct mil_frame_passes {
    const mil_hosek_wilkie_params *sky; /* HOST; sky subpass, render/mod.rs:221-226 (as mil_sky_hosek) */
    const void *accum;                  /* DEVICE R16G16B16A16_SFLOAT; resolve subpass, render/mod.rs:282-294 */
    const void *reveal;                 /* DEVICE R16_SFLOAT (both or neither) */
    uint32_t accum_pitch_px, reveal_pitch_px;
} mil_frame_passes;

/* Replaces the subpass chain lighting -> sky -> transparency resolve -> tonemap of Frame::build
 * (render/mod.rs:201-216, 221-226, 282-294, 297-309) with ONE launch: per pixel, the result of mil_light_ex is
 * replaced by the sky radiance where the depth is still the clear value, blended with the resolved transparency,
 * and encoded as `fmt`, all in registers.  The reference keeps hdr as a transient attachment of one render pass
 * (render/pbr/mod.rs:76-95: it need never reach memory); here it does not exist at all unless
 * fmt = MIL_OUT_HDR_RGBA32F asks for it.  Result = mil_light_ex(HDR) + mil_sky_hosek + mil_transparency_resolve
 * + mil_tonemap(fmt) on the same inputs.  The forward accumulate pass that fills accum / reveal
 * (render/mod.rs:232-280) reads only depth and must have run before this call.  extras, passes: may be NULL. */
int mil_shade_frame(mil_ctx *ctx, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                    const mil_directional_light *lights, const mil_push_constants *push,
                    const mil_extras *extras, const mil_frame_passes *passes, void *out, uint32_t out_pitch_px,
                    mil_out_format fmt, uint32_t flags, void *stream);

/* Band dealing (SURVEY.md 8 e, multi-GPU strips): instead of one contiguous strip, a rank holds every n_ranks-th band
 * of band_rows rows of the frame, compacted: row k of the planes is frame row
 *     (((k / band_rows) + first_band) * n_ranks + rank) * band_rows + k % band_rows.
 * Dealt bands equalise the cost of the ranks without a calibration frame (a contiguous strip of the floor costs a
 * multiple of a strip of sky).  band_rows: a power of two >= 16 (a 16x16 tile never straddles two bands).
 * first_band: the planes start at the rank's first_band-th band (0 = the rank's whole image; a launch over part of a
 * rank's rows passes the sub-view's first band). */
typedef struct mil_band_layout {
    uint32_t band_rows, n_ranks, rank, first_band;
} mil_band_layout;

/* mil_shade_frame on a rank's dealt bands (no counterpart in the reference, which has one device: render/vulkan.rs:106-108).
 * gb: the rank's compacted planes (gb->height local rows; gb->y_offset must be 0), push->resolution the whole frame.
 * out: the rank's compacted output (out_is_frame = 0) or the assembled frame, written at frame rows (out_is_frame = 1:
 * how rank 0 of a gather avoids a copy).  passes->accum / reveal, when given, are compacted like the planes.
 * Every pixel equals the monolithic mil_shade_frame pixel bit for bit.  bands = NULL: same as mil_shade_frame. */
int mil_shade_bands(mil_ctx *ctx, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                    const mil_directional_light *lights, const mil_push_constants *push, const mil_extras *extras,
                    const mil_frame_passes *passes, const mil_band_layout *bands, void *out, uint32_t out_pitch_px,
                    int out_is_frame, mil_out_format fmt, uint32_t flags, void *stream);
/* Local rows rank `rank` holds of a frame of `height` rows under `bands` (host arithmetic only; first_band ignored). */
uint32_t mil_band_local_rows(const mil_band_layout *bands, uint32_t height);

/* NEXT ROW, the HIP half of the Vulkan <-> HIP hand-over.  The reference's attachments are Vulkan
 * images (render/pbr/mod.rs:76-95,150-160); to be read or written by this library their memory must be one allocation
 * visible to both APIs.  Either direction works through a POSIX fd (on amdgpu a dma-buf):
 *   mil_import_external_memory   Vulkan allocates with VkExportMemoryAllocateInfo (OPAQUE_FD), vkGetMemoryFdKHR gives fd
 *   mil_alloc_exportable_memory  this library allocates and returns an fd for VkImportMemoryFdInfoKHR
 * dev_ptr is what mil_gbuffer / hdr / out take (linear, buffer-backed images).  The fd stays the
 * caller's: close it once the other side has imported it.  These are set-up calls (they allocate); queue ordering is the
 * caller's (an exported semaphore, or vkQueueWaitIdle / mil_sync). */
typedef struct mil_external_memory mil_external_memory;
int mil_import_external_memory(mil_ctx *ctx, int fd, uint64_t size, mil_external_memory **mem, void **dev_ptr);
int mil_alloc_exportable_memory(mil_ctx *ctx, uint64_t size, mil_external_memory **mem, void **dev_ptr, int *fd,
                                uint64_t *allocated_size /* size rounded up to the allocation granularity; may be NULL */);
int mil_release_external_memory(mil_ctx *ctx, mil_external_memory *mem);

/* Diagnostics for the roofline accounting: number of (pixel, point light) pairs
 * with d^2 < r^2 and occlusion > 0, by brute force (independent of the culling).
 * count: DEVICE uint64, overwritten. */
int mil_count_pairs(mil_ctx *ctx, const mil_gbuffer *gb, const mil_frame_matrix_data *frame,
                    const mil_push_constants *push, const mil_point_light *point_lights, uint32_t n_point,
                    unsigned long long *count, void *stream);

/* Tuning.  Fast-mode launches (mil_light_ex, mil_shade_bands, mil_shade_frame; the last with a sky or a resolve subpass
 * from twice the number) with at least min_point_lights point lights take the two-kernel form
 * (k_bin: classification + two-level culling into compact per-half-tile lists; k_order: counting sort by list length;
 * k_shade: one wave per listed half tile, most expensive first), the others the one-kernel form (k_tiled); results are
 * bit-identical.  Default MIL_TWO_KERNEL_DEFAULT (measured crossover on a 4K frame); 1 = always, 0xFFFFFFFF = never.
 * One workspace per context: launches of one context on different streams are serialised. */
#define MIL_TWO_KERNEL_DEFAULT 1280u
int mil_set_two_kernel_threshold(mil_ctx *ctx, uint32_t min_point_lights);

/* The two-kernel form's workspace.  NO COUNTERPART IN THE REFERENCE beyond this: its attachments are allocated when the
 * window is created and recreated on a resize (render/pbr/mod.rs:98-261, 263-346), never inside Frame::build; likewise no
 * per-frame entry point of this library allocates for frames up to the reserved size.  mil_create (max_point_lights > 0)
 * reserves 16 bytes per half tile of a 7680 x 4320 frame + a POOL of 3072 light-list entries (2 bytes) per point light of
 * the capacity, clamped to 1 Mi .. 16 Mi entries: 29.3 MB for 4096 lights (the 8K / 4096-light frame of BASELINE.json asks
 * for 7.65 M entries, the 4K one for 2.2 M).
 *   mil_reserve        make per-frame calls on frames up to width x height allocation-free (call it where the reference
 *                      recreates its attachments); a larger frame still works: the call grows the workspace once
 *                      (hipFree + hipMalloc: waits for the device; counted in frame_call_allocations).
 *   mil_set_list_pool  resize the pool (entries).  A pool too small for a frame costs speed, never correctness: the 64x64
 *                      macro tiles whose lists found no room are shaded by the brute-force loop (same bits).
 *   mil_workspace_info what is reserved and what the last two-kernel launch used (waits for that launch). */
typedef struct mil_workspace_stats {
    uint64_t bytes;                  /* device bytes of the workspace */
    uint32_t half_tiles;             /* half tiles (16x8) it holds: 2 * ceil(w/16) * ceil(h/16) of the reserved frame */
    uint32_t pool_entries;           /* list entries the pool holds */
    uint32_t list_entries_max;       /* longest list kept (<= 960); a half tile more lights reach is shaded by brute force */
    uint32_t last_items;             /* half tiles the last two-kernel launch listed */
    uint32_t last_pool_entries;      /* list entries it asked for */
    uint32_t last_brute_macro_tiles; /* 64x64 macro tiles whose lists found no room in the pool (shaded by brute force) */
    uint32_t frame_call_allocations; /* times a per-frame call had to grow the workspace since mil_create */
} mil_workspace_stats;
int mil_reserve(mil_ctx *ctx, uint32_t width, uint32_t height);
int mil_set_list_pool(mil_ctx *ctx, uint32_t entries);
int mil_workspace_info(mil_ctx *ctx, mil_workspace_stats *out);

/* After a culled mil_light_ex: per-tile light-list lengths of that call are written to
 * `counts` (DEVICE, n_tiles uint32) when set before the call; NULL disables.  A launch with more than n_tiles
 * tiles (tiles_x * tiles_y) leaves the buffer untouched. */
int mil_set_tile_count_buffer(mil_ctx *ctx, uint32_t *counts, uint32_t n_tiles);
/* Diagnostics: per tile {start, end} in 100 MHz ticks (low 32 bits), list length, HW_ID of the shading wave.
 * stamps: DEVICE, 4 * n_tiles uint32; NULL disables; larger launches leave it untouched. */
int mil_set_tile_stamp_buffer(mil_ctx *ctx, uint32_t *stamps, uint32_t n_tiles);
/* Diagnostics, host arithmetic only (no GPU): the scheduling weight of every 16-row tile row of a culled launch =
 * the number of point lights whose bounding sphere projects onto it (rows are launched in descending weight; speed
 * only).  weights: HOST, ceil(gb_height / 16) uint32.  bands may be NULL. */
int mil_row_weights(const mil_frame_matrix_data *frame, const mil_push_constants *push, uint32_t gb_height,
                    uint32_t y_offset, const mil_band_layout *bands, const mil_point_light *point_lights,
                    uint32_t n_point, uint32_t *weights);

/* ---- Multi-GPU (SURVEY.md 8 b "Threading", 8 e): strips of one frame / one frame per GPU, gathered to GPU 0. ----
 * No counterpart in the reference, which records the lighting draw for ONE device (render/mod.rs:189-219; the device is
 * picked at render/vulkan.rs:106-108): these calls replace that draw for N devices.  A mil_mg owns, per rank it drives,
 * two shading lanes (a mil_ctx and a compute stream each, taken in turn frame by frame so that one frame's ramp runs under
 * the previous frame's tail), a gather stream and an RCCL communicator.  The pass has no data-path exchange; the only
 * communication is the gather, grouped ncclSend / ncclRecv to rank 0 (RCCL has no gather primitive), issued per CHUNK of a
 * rank's rows on the gather stream behind an event, so it runs beside the shading of the next chunk.  The assembled frame
 * equals the single-GPU frame bit for bit.  RCCL is bound at run time (dlopen librccl.so.1): MIL_ERR_RCCL if it is missing.
 * One mil_mg = one thread at a time; calls enqueue and return, mil_mg_sync waits. */
typedef struct mil_mg mil_mg;
enum { MIL_MG_TRANSPORT_RCCL = 0,      /* grouped ncclSend / ncclRecv over xGMI */
       MIL_MG_TRANSPORT_PEER_COPY = 1  /* hipMemcpyAsync device-to-device into GPU 0's image (one process only; the intra-node
                                          alternative of SURVEY.md 8 e, and the way several ranks are rehearsed on one GPU) */ };
enum { MIL_MG_FLAG_LOOPBACK = 1u       /* testing: rank 0 shades into its own `out` and sends to itself through the transport
                                          instead of writing in place (exercises send / recv with a single GPU) */ };
typedef struct mil_mg_config {
    uint32_t max_point_lights;         /* capacity of every rank's context */
    uint32_t band_rows;                /* strips: 0 = one contiguous strip per rank, cut on 16-row multiples; else the rows of a
                                          band (power of two, 16..4096) dealt cyclically over the ranks (mil_band_layout) */
    uint32_t chunks;                   /* launches per rank and frame (1..64, 0 = 4): the gather of chunk k runs beside the shading of
                                          chunk k + 1.  1 = best throughput when frames are enqueued back to back (the gather of frame
                                          i then hides under frame i + 1); 2 = lower latency of a single frame; every launch pays the
                                          tail of its heaviest tiles, so more chunks cost more than they hide */
    uint32_t transport, flags;
} mil_mg_config;
#define MIL_MG_UNIQUE_ID_BYTES 128

/* One process driving n_devices GPUs (ncclCommInitAll); rank i = devices[i], rank 0 receives. */
int mil_mg_create(mil_mg **out, int n_devices, const int *devices, const mil_mg_config *cfg);
/* One process per GPU: rank 0 calls mil_mg_unique_id (ncclGetUniqueId), the host distributes the 128 bytes, every
 * process calls mil_mg_create_rank (ncclCommInitRank; collective: returns when all ranks have called). */
int mil_mg_unique_id(void *id);
int mil_mg_create_rank(mil_mg **out, int world, int rank, int device, const void *id, const mil_mg_config *cfg);
void mil_mg_destroy(mil_mg *mg);
int mil_mg_world(const mil_mg *mg);
int mil_mg_local_ranks(const mil_mg *mg);   /* ranks this process drives (io[] has that many entries, in rank order) */
int mil_mg_rccl_ranks(const mil_mg *mg);    /* ncclCommCount of the communicator (0 with the peer-copy transport) */
int mil_mg_last_rccl_error(const mil_mg *mg);
int mil_mg_last_hip_error(const mil_mg *mg);
/* The rows of a rank under cfg (host arithmetic, no GPU): how many, and which frame row each local row is. */
uint32_t mil_mg_rows_of_rank(const mil_mg_config *cfg, int world, int rank, uint32_t height);
int mil_mg_frame_rows_of_rank(const mil_mg_config *cfg, int world, int rank, uint32_t height, uint32_t *frame_rows);

/* What one rank contributes; every pointer on that rank's device. */
typedef struct mil_mg_rank_io {
    mil_gbuffer gb;               /* the rank's rows of the planes, compacted (gb.height = mil_mg_rows_of_rank; y_offset ignored) */
    const mil_extras *extras;     /* HOST struct (point lights HOST, shadow map on that device); may be NULL; ldr_out must be NULL */
    void *out;                    /* the rank's rows of the output, dense (out_pitch_px = width); rank 0 writes straight into the
                                     destination instead (out may be NULL) unless MIL_MG_FLAG_LOOPBACK */
    uint32_t out_pitch_px;
} mil_mg_rank_io;

/* ONE frame over all ranks (replaces render/mod.rs:201-216 + :297-309 for N devices): rank r shades its rows with
 * mil_light_ex semantics (lighting fused with the encode of fmt) and the rows are gathered into `full` on rank 0's device
 * (whole frame, dense; ignored by processes that do not drive rank 0).  push->resolution = the whole frame.
 * The call returns when the work is enqueued (mil_mg_sync waits).  Calls may follow one another without a sync: a frame
 * waits for the transfers of an earlier frame only when it writes a buffer (io[].out, full, a slot) that frame used, so a
 * caller that rotates its buffers has the gather of frame i running under the shading of frame i + 1; up to four frames are
 * tracked.  Dealt bands are received into a staging area on rank 0's device and scattered into `full` from there:
 * mil_mg_reserve allocates it (and every local context's workspace, mil_reserve) for a frame size AHEAD of the frame loop,
 * so that no per-frame call allocates; without it the first frame (or a larger one) allocates it inside the call. */
int mil_mg_reserve(mil_mg *mg, uint32_t width, uint32_t height, mil_out_format fmt);
int mil_mg_render_strips(mil_mg *mg, const mil_mg_rank_io *io, const mil_frame_matrix_data *frame,
                         const mil_directional_light *lights, const mil_push_constants *push, mil_out_format fmt,
                         uint32_t flags, void *full, uint32_t full_pitch_px);
/* ONE FRAME PER RANK (frame-parallel batches): frames[i] is the camera of local rank i, every rank shades a whole frame
 * (its own planes, its own extras) and rank r's frame is gathered into slots[r] on rank 0's device (world pointers). */
int mil_mg_render_frames(mil_mg *mg, const mil_mg_rank_io *io, const mil_frame_matrix_data *frames,
                         const mil_directional_light *lights, const mil_push_constants *push, mil_out_format fmt,
                         uint32_t flags, void *const *slots, uint32_t slot_pitch_px);
int mil_mg_sync(mil_mg *mg);          /* waits for every local rank's compute and gather stream; reports async RCCL errors */
/* Test hook, NO COUNTERPART IN THE REFERENCE: idles rank 0's gather stream for `microseconds` (<= 100 000) at the point where
 * it is called, so that a test can hold the scatter of one frame back while the next frame is already being shaded and sent
 * (the ordering of the staging slices).  Does nothing in a process that does not drive rank 0. */
int mil_mg_debug_delay_gather(mil_mg *mg, uint32_t microseconds);
/* The same on the gather streams of the local ranks OTHER than 0 (the senders): a frame's rows then leave a sender's buffer
 * late, which is what the ordering of the next frame's shading behind them (two lanes per rank) is tested against. */
int mil_mg_debug_delay_senders(mil_mg *mg, uint32_t microseconds);

int mil_sync(mil_ctx *ctx, void *stream);
int mil_last_hip_error(const mil_ctx *ctx);
const char *mil_strerror(int status);
int mil_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MIL_H */
