/*
 * FP32 C restatement of the reference's deferred-lighting + tonemap shaders, in
 * GLSL operator order.  TEST INFRASTRUCTURE ONLY: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (renderer_amd/, libmil.so) never does.
 *
 * PARITY: the reference has no tests, fixtures or golden images for its render
 * path and cannot be built in this image (no rustc, no GLSL compiler, no Vulkan
 * ICD).  The shader arithmetic below is pinned to the reference's own source:
 * oracle/glsl_exec.py executes the .glsl files and this file reproduces their
 * output bit for bit (tests/test_glsl_reference.py, tests/golden/glsl_reference.npz).
 * UNPINNED: texel format conversions, GLSL built-in precision / evaluation order
 * (the rules listed below), blending, and the build-defined extensions.  Also
 * pinned by the hand-derived known answers in tests/test_oracle_kat.py and by the
 * independent float64 restatement oracle/mil_oracle.py.
 *
 * Build (oracle/Makefile): gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp.
 * With contraction off and SSE arithmetic every operation below is one
 * correctly-rounded IEEE-754 binary32 operation, which is what makes the
 * "strict" HIP kernels comparable bit for bit.
 *
 * Evaluation rules chosen where GLSL leaves them to the implementation:
 *   a op b op c            left to right, no FMA contraction
 *   dot(a,b)               (a.x*b.x + a.y*b.y) + a.z*b.z
 *   mat4 * vec4            ((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w   (column-major)
 *   normalize(v)           v * (1.0 / sqrt(dot(v,v)))
 *   mix(x,y,a)             x*(1-a) + y*a           (GLSL spec definition)
 *   clamp(x,lo,hi)         min(max(x,lo),hi)
 *   pow(x,5)               x2=x*x; x4=x2*x2; x4*x
 *   UNORM decode           (float)c / (2^b - 1)
 *
 * All citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_LIT 3.14159f /* renderer/shaders/inc_brdf.glsl:6,46 */

typedef struct { float x, y, z; } v3;

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 normalize3(v3 v)
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    v3 r = { v.x * inv, v.y * inv, v.z * inv };
    return r;
}
/* column-major m[c*4+r] */
static inline float mrow(const float *m, int r, float x, float y, float z, float w)
{
    return ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

/* inc_brdf.glsl:2-8 */
static inline float D_GGX(float roughness, float NdotH)
{
    float oneMinusNoHSquared = 1.0f - NdotH * NdotH;
    float a = NdotH * roughness;
    float k = roughness / (oneMinusNoHSquared + a * a);
    float d = (k * k) * (1.0f / PI_LIT);
    return d;
}

/* inc_brdf.glsl:14-19 */
static inline float V_SmithGGXCorrelated(float roughness, float NdotV, float NdotL)
{
    float a2 = roughness * roughness;
    float GGXV = NdotL * sqrtf((NdotV - a2 * NdotV) * NdotV + a2);
    float GGXL = NdotV * sqrtf((NdotL - a2 * NdotL) * NdotL + a2);
    return 0.5f / (GGXV + GGXL);
}

/* inc_brdf.glsl:34-43 specular(), with fresnel() / F_Schlick :25-32 inlined (the vec3 h argument is unused) */
static inline v3 brdf_specular(float roughness, v3 albedo, float metallic, float NdotV, float NdotL, float NdotH, float LdotH)
{
    float om = 1.0f - metallic;
    v3 F0 = { 0.04f * om + albedo.x * metallic, 0.04f * om + albedo.y * metallic, 0.04f * om + albedo.z * metallic };
    float D = D_GGX(roughness, NdotH);
    float Vis = V_SmithGGXCorrelated(roughness, NdotV, NdotL);
    const float c = 16.5f; /* 50.0 * 0.33, inc_brdf.glsl:30 */
    float f90 = clampf((F0.x * c + F0.y * c) + F0.z * c, 0.0f, 1.0f);
    float x = 1.0f - LdotH, x2 = x * x, x4 = x2 * x2, pw = x4 * x; /* pow(1-VdotH, 5), :26 */
    v3 F = { F0.x + (f90 - F0.x) * pw, F0.y + (f90 - F0.y) * pw, F0.z + (f90 - F0.z) * pw };
    float DV = D * Vis;
    v3 spec = { DV * F.x, DV * F.y, DV * F.z };
    return spec;
}

/* inc_brdf.glsl:49-63 light(), with diffuse() :45-47 inlined */
static inline v3 brdf_light(v3 N, v3 L, v3 V, v3 lightColor, float roughness, v3 albedo, float metallic)
{
    v3 Hs = { L.x + V.x, L.y + V.y, L.z + V.z };
    v3 H = normalize3(Hs);
    float NdotV = clampf(dot3(N, V), 0.0001f, 1.0f);
    float NdotL = clampf(dot3(N, L), 0.0f, 1.0f);
    float NdotH = clampf(dot3(N, H), 0.0f, 1.0f);
    float LdotH = clampf(dot3(L, H), 0.0f, 1.0f);

    float om = 1.0f - metallic;
    v3 spec = brdf_specular(roughness, albedo, metallic, NdotV, NdotL, NdotH, LdotH);
    v3 diff = { albedo.x / PI_LIT, albedo.y / PI_LIT, albedo.z / PI_LIT };
    /* color = diffuse*(1-metallic) + mix(specular, specular*albedo, metallic)   :60 */
    v3 color = { diff.x * om + (spec.x * om + (spec.x * albedo.x) * metallic),
                 diff.y * om + (spec.y * om + (spec.y * albedo.y) * metallic),
                 diff.z * om + (spec.z * om + (spec.z * albedo.z) * metallic) };
    v3 r = { (color.x * lightColor.x) * NdotL, (color.y * lightColor.y) * NdotL, (color.z * lightColor.z) * NdotL };
    return r;
}

/* EXTENSION E3: 3x3 PCF for directional light 0 */
static inline float pcf_shadow(v3 P, const float *lvp, const float *map, int S, float bias)
{
    float qx = mrow(lvp, 0, P.x, P.y, P.z, 1.0f);
    float qy = mrow(lvp, 1, P.x, P.y, P.z, 1.0f);
    float zr = mrow(lvp, 2, P.x, P.y, P.z, 1.0f);
    float u = qx * 0.5f + 0.5f, v = qy * 0.5f + 0.5f;
    if (!(u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f) || zr > 1.0f) return 1.0f;
    int tx = (int)floorf(u * (float)S), ty = (int)floorf(v * (float)S);
    float zc = zr - bias;
    int cnt = 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            int xx = tx + dx, yy = ty + dy;
            xx = xx < 0 ? 0 : (xx > S - 1 ? S - 1 : xx);
            yy = yy < 0 ? 0 : (yy > S - 1 ? S - 1 : yy);
            cnt += (zc <= map[(size_t)yy * S + xx]) ? 1 : 0;
        }
    return (float)cnt / 9.0f;
}

/*
 * fs_deferred_lighting.glsl:40-67 main(), one call per strip.
 * frame_ubo: 68 floats (FrameMatrixData, render/ubo.rs:28-41)
 * dir_lights: n_dir x 8 floats {direction.xyz,intensity,color.xyz,pad} (ubo.rs:54-63)
 * point_lights: n_point x 8 floats {position.xyz,radius,color.xyz,intensity}   EXTENSION E1
 * ambient: 3 floats or NULL (EXTENSION E4); shadow_lvp: 16 floats or NULL (EXTENSION E3)
 * hdr: height x width x 4 floats, densely packed.
 * pairs_out (optional): number of (pixel, point light) pairs that were in radius.
 */
void milo_shade(const uint32_t *normal, const uint32_t *albedo_occ, const uint32_t *rough_metal, const float *depth,
                int width, int height, int pitch_px, int y_offset, int full_height,
                const float *frame_ubo, const float *dir_lights, int n_dir,
                const float *point_lights, int n_point, const float *ambient,
                const float *shadow_lvp, const float *shadow_map, int shadow_size, float shadow_bias,
                float *hdr, int nthreads, long long *pairs_out)
{
    const float *invProj = frame_ubo + 32, *invView = frame_ubo + 48;
    const v3 cam = { frame_ubo[64], frame_ubo[65], frame_ubo[66] };
    const float resx = (float)width, resy = (float)full_height; /* push_constants.resolution, render/mod.rs:211-214 */
    long long pairs = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : pairs)
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            size_t i = (size_t)y * pitch_px + x;
            uint32_t u1 = normal[i], u2 = albedo_occ[i], u3 = rough_metal[i];
            /* subpassLoad + UNORM decode (:42-45) */
            v3 b1 = { (float)(u1 & 0x3FF) / 1023.0f, (float)((u1 >> 10) & 0x3FF) / 1023.0f, (float)((u1 >> 20) & 0x3FF) / 1023.0f };
            v3 albedo = { (float)(u2 & 0xFF) / 255.0f, (float)((u2 >> 8) & 0xFF) / 255.0f, (float)((u2 >> 16) & 0xFF) / 255.0f };
            float occlusion = (float)(u2 >> 24) / 255.0f;
            float roughness = clampf((float)(u3 & 0xFF) / 255.0f, 0.0001f, 1.0f); /* :51 */
            float metallic = (float)((u3 >> 8) & 0xFF) / 255.0f;
            v3 nrm = { b1.x * 2.0f - 1.0f, b1.y * 2.0f - 1.0f, b1.z * 2.0f - 1.0f }; /* :48 */
            /* PositionFromDepth :30-38 */
            float cx = ((float)x + 0.5f) / resx, cy = ((float)(y + y_offset) + 0.5f) / resy;
            float px = cx * 2.0f - 1.0f, py = cy * 2.0f - 1.0f, pz = depth[i];
            float vx = mrow(invProj, 0, px, py, pz, 1.0f), vy = mrow(invProj, 1, px, py, pz, 1.0f);
            float vz = mrow(invProj, 2, px, py, pz, 1.0f), vw = mrow(invProj, 3, px, py, pz, 1.0f);
            vx = vx / vw; vy = vy / vw; vz = vz / vw; vw = vw / vw;
            v3 P = { mrow(invView, 0, vx, vy, vz, vw), mrow(invView, 1, vx, vy, vz, vw), mrow(invView, 2, vx, vy, vz, vw) };
            roughness = roughness * roughness; /* :56 */
            v3 N = normalize3(nrm);
            v3 toCam = { cam.x - P.x, cam.y - P.y, cam.z - P.z };
            v3 V = normalize3(toCam);

            float rx = 0.0f, ry = 0.0f, rz = 0.0f;
            for (int l = 0; l < n_dir; l++) { /* :62-64 */
                const float *lt = dir_lights + 8 * l;
                v3 L = { lt[0], lt[1], lt[2] }, col = { lt[4], lt[5], lt[6] };
                v3 c = brdf_light(N, L, V, col, roughness, albedo, metallic);
                float ax = (c.x * lt[3]) * occlusion, ay = (c.y * lt[3]) * occlusion, az = (c.z * lt[3]) * occlusion;
                if (l == 0 && shadow_lvp) {
                    float s = pcf_shadow(P, shadow_lvp, shadow_map, shadow_size, shadow_bias);
                    ax = ax * s; ay = ay * s; az = az * s;
                }
                rx += ax; ry += ay; rz += az;
            }
            for (int l = 0; l < n_point; l++) { /* EXTENSION E1 */
                const float *lt = point_lights + 8 * l;
                v3 Lv = { lt[0] - P.x, lt[1] - P.y, lt[2] - P.z };
                float d2 = dot3(Lv, Lv), r2 = lt[3] * lt[3];
                if (d2 < r2 && occlusion > 0.0f) {
                    pairs++;
                    float inv = 1.0f / sqrtf(fmaxf(d2, 1e-12f));
                    v3 L = { Lv.x * inv, Lv.y * inv, Lv.z * inv }, col = { lt[4], lt[5], lt[6] };
                    float q = d2 / r2, w = clampf(1.0f - q * q, 0.0f, 1.0f);
                    float att = (w * w) / (d2 + 1e-4f);
                    v3 c = brdf_light(N, L, V, col, roughness, albedo, metallic);
                    rx += ((c.x * lt[7]) * att) * occlusion;
                    ry += ((c.y * lt[7]) * att) * occlusion;
                    rz += ((c.z * lt[7]) * att) * occlusion;
                }
            }
            if (ambient) { /* EXTENSION E4 */
                rx += (ambient[0] * albedo.x) * occlusion;
                ry += (ambient[1] * albedo.y) * occlusion;
                rz += (ambient[2] * albedo.z) * occlusion;
            }
            float *o = hdr + ((size_t)y * width + x) * 4;
            o[0] = rx; o[1] = ry; o[2] = rz; o[3] = 1.0f; /* :66 */
        }
    }
    if (pairs_out) *pairs_out = pairs;
}

/* fs_tonemap.glsl:14-21 */
static inline float aces1(float x)
{
    float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
    return clampf((x * (a * x + b)) / (x * (c * x + d) + e), 0.0f, 1.0f);
}

/* B10G11R11_UFLOAT_PACK32 channel encode, round-to-nearest-even; x in [0,1] */
static inline uint32_t ufloat_enc(float x, int mbits)
{
    uint32_t bits;
    memcpy(&bits, &x, 4);
    int sh = 23 - mbits;
    if (x >= 6.103515625e-05f) { /* 2^-14: normal in the target format */
        uint32_t r = bits + ((1u << (sh - 1)) - 1u) + ((bits >> sh) & 1u);
        return (r >> sh) - ((127u - 15u) << mbits);
    }
    return (uint32_t)lrintf(x * (float)(1u << (14 + mbits))); /* denormal: m * 2^-(14+mbits), RNE */
}
static inline float ufloat_dec(uint32_t code, int mbits)
{
    uint32_t e = code >> mbits, m = code & ((1u << mbits) - 1u);
    if (e == 0) return (float)m * (1.0f / (float)(1u << (14 + mbits)));
    return ldexpf(1.0f + (float)m / (float)(1u << mbits), (int)e - 15);
}
uint32_t milo_pack_r11g11b10f(float r, float g, float b)
{
    return ufloat_enc(r, 6) | (ufloat_enc(g, 6) << 11) | (ufloat_enc(b, 5) << 22);
}
void milo_unpack_r11g11b10f(uint32_t u, float *rgb)
{
    rgb[0] = ufloat_dec(u & 0x7FF, 6); rgb[1] = ufloat_dec((u >> 11) & 0x7FF, 6); rgb[2] = ufloat_dec(u >> 22, 5);
}

/* sRGB OETF + 8-bit, B8G8R8A8_SRGB (render/renderer.rs:78-83); double pow so the LUT is exact */
static inline uint32_t srgb8(double c)
{
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    double s = c <= 0.0031308 ? 12.92 * c : 1.055 * pow(c, 1.0 / 2.4) - 0.055;
    return (uint32_t)floor(s * 255.0 + 0.5);
}

/* fs_tonemap.glsl:23-27 + attachment encode.  fmt: 1 R11G11B10F, 2 sRGB8 direct, 3 sRGB8 via R11G11B10F */
void milo_tonemap(const float *hdr, int width, int height, int hdr_pitch_px, int fmt, uint32_t *out, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const float *p = hdr + ((size_t)y * hdr_pitch_px + x) * 4;
            float r = aces1(p[0]), g = aces1(p[1]), b = aces1(p[2]);
            uint32_t o;
            if (fmt == 1) o = milo_pack_r11g11b10f(r, g, b);
            else if (fmt == 2) o = srgb8(b) | (srgb8(g) << 8) | (srgb8(r) << 16) | 0xFF000000u;
            else {
                float q[3];
                milo_unpack_r11g11b10f(milo_pack_r11g11b10f(r, g, b), q);
                o = srgb8(q[2]) | (srgb8(q[1]) << 8) | (srgb8(q[0]) << 16) | 0xFF000000u;
            }
            out[(size_t)y * width + x] = o;
        }
}

/*
 * NEXT ROW f-1: FXAA, renderer/shaders/fs_fxaa.glsl:22-91, drawn at render/mod.rs:312-329 from the
 * ldr attachment (B10G11R11_UFLOAT) to the swapchain image (B8G8R8A8_SRGB, render/renderer.rs:78-83).
 * Sampler (render/fxaa/mod.rs:76-88): NEAREST, CLAMP_TO_EDGE, normalised coordinates, so
 *   texture(tex, uv) = texel (clamp(floor(uv.x * W), 0, W-1), clamp(floor(uv.y * H), 0, H-1)).
 * gl_FragCoord.xy = (x + 0.5, y + 0.5).  GLSL constant expressions (1.0/3.0 - 0.5), (2.0/3.0 - 0.5),
 * (0.25 * FXAA_REDUCE_MUL) are folded the way glslang does it: in double, then rounded to float.
 */
static inline void fxaa_tap(const uint32_t *ldr, int W, int H, int pitch, float u, float v, float *rgb)
{
    float fx = floorf(u * (float)W), fy = floorf(v * (float)H);
    int ix = fx < 0.0f ? 0 : (fx > (float)(W - 1) ? W - 1 : (int)fx);
    int iy = fy < 0.0f ? 0 : (fy > (float)(H - 1) ? H - 1 : (int)fy);
    milo_unpack_r11g11b10f(ldr[(size_t)iy * pitch + ix], rgb);
}
static inline float fxaa_luma(const float *c) { return (c[0] * 0.299f + c[1] * 0.587f) + c[2] * 0.114f; }

void milo_fxaa(const uint32_t *ldr, int width, int height, int pitch_px, int row_begin, int row_count, uint32_t *out,
               int nthreads)
{
    const float K1 = (float)(1.0 / 3.0 - 0.5), K2 = (float)(2.0 / 3.0 - 0.5), KRED = (float)(0.25 * (1.0 / 8.0));
    const float REDUCE_MIN = (float)(1.0 / 128.0), SPAN = 8.0f;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int yy = 0; yy < row_count; yy++)
        for (int x = 0; x < width; x++) {
            const int y = row_begin + yy;
            const float ivx = 1.0f / (float)width, ivy = 1.0f / (float)height;      /* :23 inverseVP */
            const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
            float nw[3], ne[3], sw[3], se[3], m[3], t0[3], t1[3], t2[3], t3[3];
            fxaa_tap(ldr, width, height, pitch_px, (fx + -1.0f) * ivx, (fy + -1.0f) * ivy, nw);   /* :24-28, :37-41 */
            fxaa_tap(ldr, width, height, pitch_px, (fx + 1.0f) * ivx, (fy + -1.0f) * ivy, ne);
            fxaa_tap(ldr, width, height, pitch_px, (fx + -1.0f) * ivx, (fy + 1.0f) * ivy, sw);
            fxaa_tap(ldr, width, height, pitch_px, (fx + 1.0f) * ivx, (fy + 1.0f) * ivy, se);
            const float umx = fx * ivx, umy = fy * ivy;
            fxaa_tap(ldr, width, height, pitch_px, umx, umy, m);
            const float lNW = fxaa_luma(nw), lNE = fxaa_luma(ne), lSW = fxaa_luma(sw), lSE = fxaa_luma(se), lM = fxaa_luma(m);
            const float lmin = fminf(lM, fminf(fminf(lNW, lNE), fminf(lSW, lSE)));           /* :49-50 */
            const float lmax = fmaxf(lM, fmaxf(fmaxf(lNW, lNE), fmaxf(lSW, lSE)));
            float dx = -((lNW + lNE) - (lSW + lSE)), dy = ((lNW + lSW) - (lNE + lSE));        /* :53-54 */
            const float reduce = fmaxf((((lNW + lNE) + lSW) + lSE) * KRED, REDUCE_MIN);       /* :56-57 */
            const float rcpmin = 1.0f / (fminf(fabsf(dx), fabsf(dy)) + reduce);               /* :59 */
            dx = fminf(SPAN, fmaxf(-SPAN, dx * rcpmin)) * ivx;                                 /* :60-62 */
            dy = fminf(SPAN, fmaxf(-SPAN, dy * rcpmin)) * ivy;
            fxaa_tap(ldr, width, height, pitch_px, umx + dx * K1, umy + dy * K1, t0);         /* :64 */
            fxaa_tap(ldr, width, height, pitch_px, umx + dx * K2, umy + dy * K2, t1);
            fxaa_tap(ldr, width, height, pitch_px, umx + dx * -0.5f, umy + dy * -0.5f, t2);   /* :65 */
            fxaa_tap(ldr, width, height, pitch_px, umx + dx * 0.5f, umy + dy * 0.5f, t3);
            float a[3], b[3];
            for (int c = 0; c < 3; c++) {
                a[c] = 0.5f * (t0[c] + t1[c]);
                b[c] = a[c] * 0.5f + 0.25f * (t2[c] + t3[c]);
            }
            const float lB = fxaa_luma(b);
            const float *o = (lB < lmin || lB > lmax) ? a : b;                                 /* :67-71 */
            out[(size_t)yy * width + x] = srgb8(o[2]) | (srgb8(o[1]) << 8) | (srgb8(o[0]) << 16) | 0xFF000000u;
        }
}

/*
 * NEXT ROW f-4: the Hosek-Wilkie sky subpass, renderer/shaders/sky_hosek_frag.glsl:28-60 (+ sky_hosek_vert.glsl),
 * drawn at render/mod.rs:221-226 into hdr.  The vertex stage puts the sky mesh at depth (w - 1e-5) / w and the
 * pipeline tests LessOrEqual without depth writes (render/hosek/mod.rs:58-64): the sky lands on the texels still
 * at the clear depth 1.0 (or one ulp below: 1 - 1e-5/w rounds to 0x3F7FFFFF for w <= 200).  The fragment's
 * interpolated `position` lies on the view ray of the pixel, so normalize(position - cameraPosition) is the
 * normalised PositionFromDepth ray (fs_deferred_lighting.glsl:30-38) whatever the mesh is.
 * exp / pow / acos are taken in double and rounded (the strict HIP kernel does the same), every other
 * operation is one FP32 operation in GLSL order.  params: 11 vec3 at 16-byte stride (A B C D E F G H I Z sun).
 */
static inline float f_exp(float x) { return (float)exp((double)x); }
void milo_sky_hosek(const float *depth, int width, int height, int pitch_px, int y_offset, int full_height,
                    const float *ubo, const float *params, float *hdr, int hdr_pitch_px, int nthreads)
{
    const float *inv_proj = ubo + 32, *inv_view = ubo + 48, *cam = ubo + 64;
    const float res_x = (float)width, res_y = (float)full_height;
    const float *A = params, *B = params + 4, *C = params + 8, *D = params + 12, *E = params + 16, *F = params + 20,
                *G = params + 24, *H = params + 28, *I = params + 32, *Z = params + 36, *sun = params + 40;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const float d = depth[(size_t)y * pitch_px + x];
            if (!(d >= 0.99999994f)) continue;                       /* depth test LessOrEqual fails: hdr untouched */
            float cx = ((float)x + 0.5f) / res_x, cy = ((float)(y + y_offset) + 0.5f) / res_y;
            float px = cx * 2.0f - 1.0f, py = cy * 2.0f - 1.0f;
            float vx = mrow(inv_proj, 0, px, py, d, 1.0f), vy = mrow(inv_proj, 1, px, py, d, 1.0f);
            float vz = mrow(inv_proj, 2, px, py, d, 1.0f), vw = mrow(inv_proj, 3, px, py, d, 1.0f);
            vx = vx / vw; vy = vy / vw; vz = vz / vw; vw = vw / vw;
            v3 P = { mrow(inv_view, 0, vx, vy, vz, vw), mrow(inv_view, 1, vx, vy, vz, vw), mrow(inv_view, 2, vx, vy, vz, vw) };
            v3 vd = normalize3((v3){ P.x - cam[0], P.y - cam[1], P.z - cam[2] });            /* :56-58 */
            v3 s = { sun[0], sun[1], sun[2] };
            const float cg = fmaxf(0.001f, dot3(s, vd));                                      /* :46 */
            const float ct = fmaxf(0.001f, (vd.x * 0.0f + vd.y * 1.0f) + vd.z * 0.0f);        /* :47 */
            const float gamma = (float)acos((double)cg);                                      /* :49 */
            float *o = hdr + ((size_t)y * hdr_pitch_px + x) * 4;
            for (int c = 0; c < 3; c++) {
                const float chi = (1.0f + cg * cg) / (float)pow((double)((1.0f + H[c] * H[c]) - (2.0f * H[c]) * cg), 1.5);   /* :39 */
                const float left = 1.0f + A[c] * f_exp(B[c] / (ct + 0.01f));                  /* :41 */
                const float right = (((C[c] + D[c] * f_exp(E[c] * gamma)) + (F[c] * cg) * cg) + G[c] * chi) + I[c] * sqrtf(ct);
                o[c] = ((left * right) * Z[c]) * 0.05f;                                       /* :51, :58 */
            }
            o[3] = 1.0f;
        }
}

/*
 * NEXT ROW f-2: transparency resolve, renderer/shaders/fs_mcguire13_resolve.glsl:8-13 with the blend state of
 * render/mcguire13/mod.rs:108-120 (colour and alpha: src * (1 - srcAlpha) + dst * srcAlpha), drawn at
 * render/mod.rs:282-294 onto hdr.  accum: R16G16B16A16_SFLOAT, reveal: R16_SFLOAT (mcguire13/mod.rs:22-23).
 */
static inline float half_dec(uint16_t h)
{
    uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FF;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf(1.0f + (float)m / 1024.0f, (int)e - 15);
    return (h & 0x8000) ? -v : v;
}
void milo_transparency_resolve(const uint16_t *accum, int accum_pitch_px, const uint16_t *reveal, int reveal_pitch_px,
                               int width, int height, float *hdr, int hdr_pitch_px, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            const uint16_t *a = accum + ((size_t)y * accum_pitch_px + x) * 4;
            const float r = half_dec(reveal[(size_t)y * reveal_pitch_px + x]);
            const float den = fminf(5e4f, fmaxf(1e-4f, half_dec(a[3])));
            float *o = hdr + ((size_t)y * hdr_pitch_px + x) * 4;
            const float om = 1.0f - r;
            for (int c = 0; c < 3; c++) o[c] = (half_dec(a[c]) / den) * om + o[c] * r;
            o[3] = r * om + o[3] * r;
        }
}

/*
 * NEXT ROW f-2, second half: the fragment shader of the transparency ACCUMULATE draw,
 * renderer/shaders/fs_mcguire13_accumulation.glsl:90-127 (drawn at render/mod.rs:229-281), over a list of
 * fragments.  What the rasteriser and the texture units hand the shader arrives as four planes:
 *   pos_z      float4  in_wsPosition.xyz, gl_FragCoord.z
 *   nrm_op     float4  in_normal.xyz (interpolated, not normalised), opacity  = material_data.opacity   * texel  (:96)
 *   alb_rough  float4  albedo.rgb = material_data.albedo_color * texel (:91), roughness = material_data.roughness * texel (:93)
 *   metallic   float   material_data.metallic * texel (:94)
 * Out, per fragment: accum = vec4(Ci, ai) * w10(zi, ai) (:121-125), reveal = ai (:126): the values the blend
 * stage (mcguire13/mod.rs:57-84) adds into the RGBA16F / multiplies into the R16F attachment.
 * Unlike light() there is no NdotL on the diffuse term, no intensity, no occlusion, and roughness is NOT
 * clamped before it is squared (:104): roughness = 0 makes D * V = 0 * inf = NaN wherever N.L = 0, as in the shader.
 * lights: n_dir records of 8 floats {direction.xyz, intensity, color.xyz, pad}; cam: cameraPosition.
 */
void milo_transparency_accumulate(const float *pos_z, const float *nrm_op, const float *alb_rough, const float *metallic,
                                  int count, const float *cam, const float *lights, int n_dir, float *accum,
                                  float *reveal, int nthreads)
{
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (int i = 0; i < count; i++) {
        const float *pz = pos_z + 4 * (size_t)i, *no = nrm_op + 4 * (size_t)i, *ar = alb_rough + 4 * (size_t)i;
        v3 albedo = { ar[0], ar[1], ar[2] };
        float roughness = ar[3], met = metallic[i], opacity = no[3];
        roughness = roughness * roughness;                                   /* :104 */
        v3 nrm = { no[0], no[1], no[2] }, N = normalize3(nrm);               /* :106 */
        v3 Vd = { cam[0] - pz[0], cam[1] - pz[1], cam[2] - pz[2] }, V = normalize3(Vd); /* :107 */
        v3 lighting = { 0.0f, 0.0f, 0.0f };
        for (int l = 0; l < n_dir; l++) {                                    /* :110-119 */
            const float *r = lights + 8 * (size_t)l;
            v3 L = { r[0], r[1], r[2] };
            v3 Hs = { L.x + V.x, L.y + V.y, L.z + V.z }, H = normalize3(Hs);
            float NdotV = clampf(dot3(N, V), 0.0001f, 1.0f);
            float NdotL = clampf(dot3(N, L), 0.0f, 1.0f);
            float NdotH = clampf(dot3(N, H), 0.0f, 1.0f);
            float LdotH = clampf(dot3(L, H), 0.0f, 1.0f);
            v3 spec = brdf_specular(roughness, albedo, met, NdotV, NdotL, NdotH, LdotH);
            /* lighting += diffuse(..) + specular(..) * color * NdotL */
            lighting.x = lighting.x + (albedo.x / PI_LIT + (spec.x * r[4]) * NdotL);
            lighting.y = lighting.y + (albedo.y / PI_LIT + (spec.y * r[5]) * NdotL);
            lighting.z = lighting.z + (albedo.z / PI_LIT + (spec.z * r[6]) * NdotL);
        }
        float f = 1.0f - pz[3];                                              /* w10, :84-87 */
        float w = opacity * fmaxf(0.01f, ((3000.0f * f) * f) * f);
        float *a = accum + 4 * (size_t)i;
        a[0] = (lighting.x * opacity) * w; a[1] = (lighting.y * opacity) * w; a[2] = (lighting.z * opacity) * w;
        a[3] = opacity * w;
        reveal[i] = opacity;
    }
}

int milo_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
