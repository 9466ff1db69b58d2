// Device code of libmil: deferred lighting (directional, tiled point lights, PCF
// shadow, ambient) and ACES tonemap / pack for gfx950.  Included twice:
//   mil_fast.hip    MIL_STRICT=0  hoisted per-pixel invariants, v_rcp/v_rsq/v_sqrt, FMA
//   mil_strict.hip  MIL_STRICT=1  (-ffp-contract=off) literal GLSL operator order with
//                                  IEEE divide / sqrt: bit-exact against oracle/mil_oracle.c
// What the arithmetic follows (paths relative to /root/reference/):
//   renderer/shaders/fs_deferred_lighting.glsl:30-67, inc_brdf.glsl:2-63, fs_tonemap.glsl:14-27.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mil_params.h"

#ifndef MIL_STRICT
#error "define MIL_STRICT to 0 or 1"
#endif
#if MIL_STRICT
#define MIL_NS strict
#else
#define MIL_NS fast
#endif

#define DEV static __device__ __forceinline__

namespace mil {
namespace MIL_NS {

constexpr float PI_LIT = 3.14159f;   // inc_brdf.glsl:6,46 (literal, not pi)
constexpr int TILE = 16;
constexpr uint32_t DIR_INLINE = MIL_DIR_INLINE;
constexpr int CHUNK = 1024;          // point lights culled per LDS pass (u16 list, 2 KiB per wave)
constexpr int SH_CAP = 576;          // floats of shadow map staged in LDS per wave/tile (24x24 texels, 2.25 KiB)

enum { FMT_HDR = 0, FMT_LDR = 1, FMT_SRGB8 = 2, FMT_SRGB8_VIA_LDR = 3 };

template <typename T> struct V3 { T x, y, z; };
typedef V3<float> f3;
typedef float v4f __attribute__((ext_vector_type(4)));

DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
DEV float sat(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }

// uniform (wave-invariant) fetch of a staged light record through the scalar cache
typedef const __attribute__((address_space(4))) v4f *cptr4;
DEV v4f ldu(const float4 *p, uint32_t i) { return ((cptr4)(uintptr_t)p)[i]; }

// Directional light record l in this mode's layout (mil_params.h).  The first DIR_INLINE
// records travel in the kernarg segment itself, so the common 1-4 light frame needs no
// dependent pointer chase before its first BRDF evaluation.
struct DirRec { v4f a, b; };
DEV DirRec dir_rec(const ShadeParams &p, uint32_t l)
{
    DirRec d;
    if (l < DIR_INLINE) {
        const float *r = p.dir_inline[l];
        d.a = v4f{ r[0], r[1], r[2], r[3] }; d.b = v4f{ r[4], r[5], r[6], r[7] };
    } else {
#if MIL_STRICT
        d.a = ldu(p.dir_raw, 2 * l); d.b = ldu(p.dir_raw, 2 * l + 1);
#else
        d.a = ldu(p.dir_prep, 2 * l); d.b = ldu(p.dir_prep, 2 * l + 1);
#endif
    }
    return d;
}

// Rows.  A kernel addresses its planes by LOCAL rows; PositionFromDepth needs the row of the whole frame
// (fs_deferred_lighting.glsl:31: gl_FragCoord.y) and the output may be the assembled frame.  Both differ from the local row
// by an offset that is uniform over any 16-row-aligned group of rows (mil_params.h, band dealing), so every wave computes
// them once, on the scalar unit, from the first row it covers.
struct RowMap { uint32_t g, o; };      // frame row = local + g; output row = local + o
DEV RowMap row_map(const ShadeParams &p, uint32_t y_base)
{
    const uint32_t b = y_base >> p.band_shift;
    return RowMap{ p.y_offset + b * p.band_extra, b * p.out_band_extra };
}

// ---------------------------------------------------------------------------------
// per-pixel state
// ---------------------------------------------------------------------------------
#if MIL_STRICT
struct Px {
    f3 N, V, P;
    float occ;
    f3 albedo;
    float alpha, metallic;
};
#else
// T = float: one pixel per lane (K1); T = v2f: two pixels per lane, packed FP32 (K2)
typedef float v2f __attribute__((ext_vector_type(2)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
template <typename T> struct PxT {
    V3<T> N, V, P;
    T occ;
    T alpha, a2, oma2, NdotV, sV;      // oma2 = 1 - alpha^2, sV = sqrt((NdotV - a2 NdotV) NdotV + a2)
    V3<T> G0, G1;                      // tint * F0, tint * (f90 - F0), tint = ((1-m) + albedo*m)*occ
    V3<T> dif;                         // albedo/PI*(1-m)*occ
};
#endif

#if MIL_STRICT
// ---- literal transcription, one IEEE operation per GLSL operator ----
DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
DEV f3 normalize3(f3 v)
{
    float inv = 1.0f / sqrtf(dot3(v, v));
    return f3{ v.x * inv, v.y * inv, v.z * inv };
}
DEV float mrow(const float *m, int r, float x, float y, float z, float w)
{
    return ((m[0 + r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * w;
}

// PositionFromDepth, fs_deferred_lighting.glsl:30-38 (yg is the row of the whole frame)
struct Pos { f3 P; };
DEV Pos pixel_position(const ShadeParams &p, uint32_t x, uint32_t yg, float depth)
{
    float cx = ((float)x + 0.5f) / p.res_x, cy = ((float)yg + 0.5f) / p.res_y;
    float px = cx * 2.0f - 1.0f, py = cy * 2.0f - 1.0f;
    float vx = mrow(p.inv_proj, 0, px, py, depth, 1.0f), vy = mrow(p.inv_proj, 1, px, py, depth, 1.0f);
    float vz = mrow(p.inv_proj, 2, px, py, depth, 1.0f), vw = mrow(p.inv_proj, 3, px, py, depth, 1.0f);
    vx = vx / vw; vy = vy / vw; vz = vz / vw; vw = vw / vw;
    return Pos{ f3{ mrow(p.inv_view, 0, vx, vy, vz, vw), mrow(p.inv_view, 1, vx, vy, vz, vw), mrow(p.inv_view, 2, vx, vy, vz, vw) } };
}

DEV void setup_pixel(Px &q, const ShadeParams &p, Pos pos, uint32_t u1, uint32_t u2, uint32_t u3)
{
    const f3 P = pos.P;
    f3 b1 = { (float)(u1 & 0x3FF) / 1023.0f, (float)((u1 >> 10) & 0x3FF) / 1023.0f, (float)((u1 >> 20) & 0x3FF) / 1023.0f };
    q.albedo = f3{ (float)(u2 & 0xFF) / 255.0f, (float)((u2 >> 8) & 0xFF) / 255.0f, (float)((u2 >> 16) & 0xFF) / 255.0f };
    q.occ = (float)(u2 >> 24) / 255.0f;
    float roughness = clampf((float)(u3 & 0xFF) / 255.0f, 0.0001f, 1.0f);     // fs_deferred_lighting.glsl:51
    q.metallic = (float)((u3 >> 8) & 0xFF) / 255.0f;
    f3 nrm = { b1.x * 2.0f - 1.0f, b1.y * 2.0f - 1.0f, b1.z * 2.0f - 1.0f };  // :48
    q.P = P;
    q.alpha = roughness * roughness;                                         // :56
    q.N = normalize3(nrm);                                                   // :58
    q.V = normalize3(f3{ p.cam[0] - q.P.x, p.cam[1] - q.P.y, p.cam[2] - q.P.z }); // :59
}

// specular(), inc_brdf.glsl:34-43, with D_GGX :2-8, V_SmithGGXCorrelated :14-19, fresnel :25-32
DEV f3 brdf_specular(float alpha, f3 albedo, float metallic, float NdotV, float NdotL, float NdotH, float LdotH)
{
    float om = 1.0f - metallic;
    f3 F0 = { 0.04f * om + albedo.x * metallic, 0.04f * om + albedo.y * metallic, 0.04f * om + albedo.z * metallic };
    float oneMinusNoHSquared = 1.0f - NdotH * NdotH;
    float a = NdotH * alpha;
    float k = alpha / (oneMinusNoHSquared + a * a);
    float D = (k * k) * (1.0f / PI_LIT);
    float a2 = alpha * alpha;
    float GGXV = NdotL * sqrtf((NdotV - a2 * NdotV) * NdotV + a2);
    float GGXL = NdotV * sqrtf((NdotL - a2 * NdotL) * NdotL + a2);
    float Vis = 0.5f / (GGXV + GGXL);
    const float c = 16.5f;
    float f90 = clampf((F0.x * c + F0.y * c) + F0.z * c, 0.0f, 1.0f);
    float w = 1.0f - LdotH, w2 = w * w, w4 = w2 * w2, pw = w4 * w;
    f3 F = { F0.x + (f90 - F0.x) * pw, F0.y + (f90 - F0.y) * pw, F0.z + (f90 - F0.z) * pw };
    float DV = D * Vis;
    return f3{ DV * F.x, DV * F.y, DV * F.z };
}
// light(), inc_brdf.glsl:49-63, with diffuse :45-47
DEV f3 brdf_light(const Px &q, f3 L, f3 lightColor)
{
    f3 H = normalize3(f3{ L.x + q.V.x, L.y + q.V.y, L.z + q.V.z });
    float NdotV = clampf(dot3(q.N, q.V), 0.0001f, 1.0f);
    float NdotL = clampf(dot3(q.N, L), 0.0f, 1.0f);
    float NdotH = clampf(dot3(q.N, H), 0.0f, 1.0f);
    float LdotH = clampf(dot3(L, H), 0.0f, 1.0f);
    float om = 1.0f - q.metallic;
    f3 spec = brdf_specular(q.alpha, q.albedo, q.metallic, NdotV, NdotL, NdotH, LdotH);
    f3 diff = { q.albedo.x / PI_LIT, q.albedo.y / PI_LIT, q.albedo.z / PI_LIT };
    f3 color = { diff.x * om + (spec.x * om + (spec.x * q.albedo.x) * q.metallic),
                 diff.y * om + (spec.y * om + (spec.y * q.albedo.y) * q.metallic),
                 diff.z * om + (spec.z * om + (spec.z * q.albedo.z) * q.metallic) };
    return f3{ (color.x * lightColor.x) * NdotL, (color.y * lightColor.y) * NdotL, (color.z * lightColor.z) * NdotL };
}

DEV void add_directional(const Px &q, DirRec d, float shadow, bool use_shadow, f3 &res)
{
    const v4f a = d.a, b = d.b;
    f3 c = brdf_light(q, f3{ a.x, a.y, a.z }, f3{ b.x, b.y, b.z });
    float ax = (c.x * a.w) * q.occ, ay = (c.y * a.w) * q.occ, az = (c.z * a.w) * q.occ;  // fs_deferred_lighting.glsl:63
    if (use_shadow) { ax = ax * shadow; ay = ay * shadow; az = az * shadow; }
    res.x += ax; res.y += ay; res.z += az;
}

// EXTENSION E1.  d2 and the in-range predicate are shared with the culling (see box_d2).
DEV float point_d2(const Px &q, v4f a, f3 &Lv)
{
    Lv = f3{ a.x - q.P.x, a.y - q.P.y, a.z - q.P.z };
    return dot3(Lv, Lv);
}
DEV float point_r2(v4f a) { return a.w * a.w; }
DEV v4f point_rec_a(const ShadeParams &p, uint32_t l) { return ldu(p.point_raw, 2 * l); }
DEV v4f point_rec_b(const ShadeParams &p, uint32_t l) { return ldu(p.point_raw, 2 * l + 1); }
DEV v4f point_rec_a_v(const ShadeParams &p, uint32_t l) { return ((const v4f *)p.point_raw)[2 * l]; }   // per-lane (vector) fetch
DEV v4f point_rec_b_v(const ShadeParams &p, uint32_t l) { return ((const v4f *)p.point_raw)[2 * l + 1]; }
DEV bool point_test(const Px &q, v4f a, f3 &Lv, float &d2)
{
    d2 = point_d2(q, a, Lv);
    return d2 < point_r2(a) && q.occ > 0.0f;
}
DEV void point_eval(const Px &q, v4f a, v4f b, f3 Lv, float d2, bool on, f3 &res)
{
    if (on) {
        float r2 = point_r2(a);
        float inv = 1.0f / sqrtf(fmaxf(d2, 1e-12f));
        f3 L = { Lv.x * inv, Lv.y * inv, Lv.z * inv };
        float qq = d2 / r2, w = clampf(1.0f - qq * qq, 0.0f, 1.0f);
        float att = (w * w) / (d2 + 1e-4f);
        f3 c = brdf_light(q, L, f3{ b.x, b.y, b.z });
        res.x += ((c.x * b.w) * att) * q.occ;
        res.y += ((c.y * b.w) * att) * q.occ;
        res.z += ((c.z * b.w) * att) * q.occ;
    }
}
DEV void add_point(const Px &q, v4f a, v4f b, f3 &res)
{
    f3 Lv;
    float d2;
    const bool on = point_test(q, a, Lv, d2);
    point_eval(q, a, b, Lv, d2, on, res);
}
DEV void add_ambient(const Px &q, const ShadeParams &p, uint32_t, f3 &res)
{
    res.x += (p.ambient[0] * q.albedo.x) * q.occ;
    res.y += (p.ambient[1] * q.albedo.y) * q.occ;
    res.z += (p.ambient[2] * q.albedo.z) * q.occ;
}
DEV void shadow_coords(f3 P, const ShadeParams &p, float &u, float &v, float &zr)
{
    float qx = mrow(p.lvp, 0, P.x, P.y, P.z, 1.0f), qy = mrow(p.lvp, 1, P.x, P.y, P.z, 1.0f);
    zr = mrow(p.lvp, 2, P.x, P.y, P.z, 1.0f);
    u = qx * 0.5f + 0.5f; v = qy * 0.5f + 0.5f;
}
DEV float pcf_result(int cnt) { return (float)cnt / 9.0f; }
DEV float aces1(float x)
{
    return clampf((x * (2.51f * x + 0.03f)) / (x * (2.43f * x + 0.59f) + 0.14f), 0.0f, 1.0f);
}
DEV uint32_t srgb8_direct(float cf)   // the OETF in double, as oracle/mil_oracle.c srgb8(): exact up to the rounding of pow()
{
    double c = (double)cf;
    c = c < 0.0 ? 0.0 : (c > 1.0 ? 1.0 : c);
    double s = c <= 0.0031308 ? 12.92 * c : 1.055 * pow(c, 1.0 / 2.4) - 0.055;
    return (uint32_t)floor(s * 255.0 + 0.5);
}

#else
// ---- fast path: same formulas, light-invariant terms hoisted, hardware rcp/rsq/sqrt ----
// Everything per-pixel is a template over T: float (one pixel per lane) or v2f (two pixels per lane;
// hipcc lowers the v2f arithmetic to v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, one issue slot for
// both pixels, and the two pixels give every wave two independent dependency chains).
DEV float rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float rsq_(float x) { return __builtin_amdgcn_rsqf(x); }
DEV float sqrt_(float x) { return __builtin_amdgcn_sqrtf(x); }
DEV v2f rcp_(v2f x) { return v2f{ rcp_(x.x), rcp_(x.y) }; }
DEV v2f rsq_(v2f x) { return v2f{ rsq_(x.x), rsq_(x.y) }; }
DEV v2f sqrt_(v2f x) { return v2f{ sqrt_(x.x), sqrt_(x.y) }; }
DEV v2f sat(v2f x) { return v2f{ sat(x.x), sat(x.y) }; }
// Saturate folded into the packed multiply / fma that produces the value (the clamp bit of VOP3P; the compiler only folds it
// into scalar-FP32 producers and otherwise spends two v_max per pair).  The s_nop on either side is the wait state the
// compiler keeps between a packed-FP32 write and the VALU instruction that reads it, which it cannot place around inline asm.
DEV v2f sat_mul(v2f a, v2f b)
{
    v2f d;
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, %2 clamp\n\ts_nop 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
DEV v2f sat_fma(v2f a, v2f b, v2f c)
{
    v2f d;
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 clamp\n\ts_nop 0" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
DEV v2f sat_fnma(v2f a, v2f b, v2f c)      // sat(c - a * b)
{
    v2f d;
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, %3 neg_lo:[1,0,0] neg_hi:[1,0,0] clamp\n\ts_nop 0" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
// the same with an inline constant operand (op_sel_hi 0: both halves read the one 32-bit constant): no VGPR pair holds it
DEV v2f sat_half(v2f a)                    // sat(a / 2)
{
    v2f d;
    asm("s_nop 0\n\tv_pk_mul_f32 %0, %1, 0.5 op_sel_hi:[1,0] clamp\n\ts_nop 0" : "=v"(d) : "v"(a));
    return d;
}
DEV v2f sat_one_minus_mul(v2f a, v2f b)    // sat(1 - a * b)
{
    v2f d;
    asm("s_nop 0\n\tv_pk_fma_f32 %0, %1, %2, 1.0 op_sel_hi:[1,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0] clamp\n\ts_nop 0" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
DEV float sat_half(float a) { return sat(a * 0.5f); }
DEV float sat_one_minus_mul(float a, float b) { return sat(__builtin_fmaf(-a, b, 1.0f)); }
DEV float sat_mul(float a, float b) { return sat(a * b); }
DEV float sat_fma(float a, float b, float c) { return sat(__builtin_fmaf(a, b, c)); }
DEV float sat_fnma(float a, float b, float c) { return sat(__builtin_fmaf(-a, b, c)); }
DEV float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
DEV v2f fma_(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
DEV float max_(float a, float b) { return fmaxf(a, b); }
DEV v2f max_(v2f a, v2f b) { return v2f{ fmaxf(a.x, b.x), fmaxf(a.y, b.y) }; }
DEV float med3_(float a, float lo, float hi) { return __builtin_amdgcn_fmed3f(a, lo, hi); }
DEV v2f med3_(v2f a, float lo, float hi) { return v2f{ med3_(a.x, lo, hi), med3_(a.y, lo, hi) }; }
DEV float tof(uint32_t u) { return (float)u; }
DEV v2f tof(v2u u) { return __builtin_convertvector(u, v2f); }
template <typename T> DEV T dot3(V3<T> a, V3<T> b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
DEV float mrow1(const float *m, int r, float x, float y, float z)   // w == 1
{
    return __builtin_fmaf(m[0 + r], x, __builtin_fmaf(m[4 + r], y, __builtin_fmaf(m[8 + r], z, m[12 + r])));
}
typedef PxT<float> Px;
typedef PxT<v2f> Px2;

// PositionFromDepth with the two matrix products and the pixel -> clip map folded on the host
// (mil_api.cpp fill_common, in double).  The folded matrix carries inv_view's translation MINUS the
// camera position, so what comes out is D = P - cameraPosition without the cancellation the literal
// form has near the camera (V = normalize(-D)); P = D + cameraPosition.
//   D = (A . (x, y_global, depth, 1)).xyz / (A . (...)).w
template <typename T> struct PosT { V3<T> P, D; };
template <typename T> DEV PosT<T> pixel_position_t(const ShadeParams &p, T xf, T yf, T depth)
{
    const float *m = p.pos_m;
    T vx = fma_((T)m[0], xf, fma_((T)m[1], yf, fma_((T)m[2], depth, (T)m[3])));
    T vy = fma_((T)m[4], xf, fma_((T)m[5], yf, fma_((T)m[6], depth, (T)m[7])));
    T vz = fma_((T)m[8], xf, fma_((T)m[9], yf, fma_((T)m[10], depth, (T)m[11])));
    T vw = fma_((T)m[12], xf, fma_((T)m[13], yf, fma_((T)m[14], depth, (T)m[15])));
    T iw = rcp_(vw);
    PosT<T> r;
    r.D = V3<T>{ vx * iw, vy * iw, vz * iw };
    r.P = V3<T>{ r.D.x + p.cam[0], r.D.y + p.cam[1], r.D.z + p.cam[2] };
    return r;
}
typedef PosT<float> Pos;
DEV Pos pixel_position(const ShadeParams &p, uint32_t x, uint32_t yg, float depth)
{
    return pixel_position_t<float>(p, (float)x, (float)yg, depth);
}

// U = uint32_t / v2u: the three packed G-buffer texels of the pixel(s)
template <typename T, typename U> DEV void setup_pixel_t(PxT<T> &q, const PosT<T> &pos, U u1, U u2, U u3)
{
    const float k10 = 2.0f / 1023.0f, k8 = 1.0f / 255.0f;
    V3<T> nrm = { fma_(tof(u1 & 0x3FFu), (T)k10, (T)-1.0f), fma_(tof((u1 >> 10) & 0x3FFu), (T)k10, (T)-1.0f),
                  fma_(tof((u1 >> 20) & 0x3FFu), (T)k10, (T)-1.0f) };
    V3<T> albedo = { tof(u2 & 0xFFu) * k8, tof((u2 >> 8) & 0xFFu) * k8, tof((u2 >> 16) & 0xFFu) * k8 };
    q.occ = tof(u2 >> 24) * k8;
    T roughness = max_(tof(u3 & 0xFFu) * k8, (T)0.0001f);
    T m = tof((u3 >> 8) & 0xFFu) * k8;
    q.P = pos.P;
    T rn = rsq_(dot3(nrm, nrm));
    q.N = V3<T>{ nrm.x * rn, nrm.y * rn, nrm.z * rn };
    T rv = -rsq_(dot3(pos.D, pos.D));
    q.V = V3<T>{ pos.D.x * rv, pos.D.y * rv, pos.D.z * rv };
    q.alpha = roughness * roughness;
    q.a2 = q.alpha * q.alpha;
    q.oma2 = 1.0f - q.a2;
    q.NdotV = med3_(dot3(q.N, q.V), 0.0001f, 1.0f);
    q.sV = sqrt_(fma_(fma_(-q.a2, q.NdotV, q.NdotV), q.NdotV, q.a2));
    T om = 1.0f - m;
    const V3<T> F0 = { fma_(albedo.x, m, 0.04f * om), fma_(albedo.y, m, 0.04f * om), fma_(albedo.z, m, 0.04f * om) };
    const T f90 = sat((F0.x + F0.y + F0.z) * 16.5f);
    T ds = om * q.occ * (1.0f / PI_LIT);
    q.dif = V3<T>{ albedo.x * ds, albedo.y * ds, albedo.z * ds };
    const V3<T> tint = { fma_(albedo.x, m, om) * q.occ, fma_(albedo.y, m, om) * q.occ, fma_(albedo.z, m, om) * q.occ };
    q.G0 = V3<T>{ tint.x * F0.x, tint.y * F0.y, tint.z * F0.z };
    q.G1 = V3<T>{ tint.x * (f90 - F0.x), tint.y * (f90 - F0.y), tint.z * (f90 - F0.z) };
}
DEV void setup_pixel(Px &q, const ShadeParams &, Pos pos, uint32_t u1, uint32_t u2, uint32_t u3)
{
    setup_pixel_t<float, uint32_t>(q, pos, u1, u2, u3);
}

// light() with the invariants in q; rad = lightColor * intensity [* attenuation * shadow].
//   colour = dif + tint * (F0 + (f90 - F0) w^5) * D * Vis = dif + (G0 + G1 w^5) * DV
template <typename T> DEV void eval_tail(const PxT<T> &q, T NdotL, T s2, T LdotH, V3<T> col, T scale, V3<T> &res)
{
    // D_GGX's denominator (1 - x^2) + x^2 alpha^2, x = N.H, from s2 = 1 - x^2 (formed without cancellation by the caller,
    // see one_minus_ndoth2): x^2 = 1 - s2, so t = s2 (1 - alpha^2) + alpha^2 >= alpha^2 > 0
    T t = fma_(s2, q.oma2, q.a2);
    T sL = sqrt_(fma_(fma_(-q.a2, NdotL, NdotL), NdotL, q.a2));
    T den = fma_(NdotL, q.sV, q.NdotV * sL);                             // GGXV + GGXL
    // D * Vis = (alpha / t)^2 * 0.5 / (pi den) with ONE reciprocal (a quarter-rate instruction: 16 cycles a wave against 4):
    // u = alpha / (t den), D * Vis = u * (u * den * 0.5 / pi).  t den >= alpha^2 * N.V alpha >= 1e-28 and the product is
    // formed small-to-large, so nothing under- or overflows that the two-reciprocal form kept finite.
    T u = q.alpha * rcp_(t * den);
    T DV = u * (u * (den * (0.5f / PI_LIT)));                            // D * Vis
    T w = 1.0f - LdotH, w2 = w * w, w5 = w2 * w2 * w;
    T cx = fma_(fma_(q.G1.x, w5, q.G0.x), DV, q.dif.x);
    T cy = fma_(fma_(q.G1.y, w5, q.G0.y), DV, q.dif.y);
    T cz = fma_(fma_(q.G1.z, w5, q.G0.z), DV, q.dif.z);
    T sN = scale * NdotL;                                                // radiance = col * scale, scale = attenuation or shadow
    res.x = fma_(cx, col.x * sN, res.x);
    res.y = fma_(cy, col.y * sN, res.y);
    res.z = fma_(cz, col.z * sN, res.z);
}
// general L (directional lights are not normalised by the shader, inc_structs.glsl:12-16)
// 1 - clamp(N.H, 0, 1)^2 for the D_GGX denominator (inc_brdf.glsl:3-5).  Formed as 1 - x * x from a rounded x it carries
// an ABSOLUTE error of ~1e-7 whatever its size, while the denominator goes down to alpha^2 (2e-8 for the roughness-0.01
// spheres of scenes/roughness_test.rs): round 2's census had 532 pixels of the 4K headline frame outside
// 1e-3 + 1e-4 |b| of the float64 result, all of them on such highlights, 129 of them where even the reference-order FP32
// result is inside.  For unit vectors |N - Hn|^2 = 2 - 2 N.Hn =: e, so 1 - x = e / 2 and 1 - x^2 = (e/2)(2 - e/2): the
// difference N - Hn is taken component by component with one rounding (fma), its error is RELATIVE to the angle between
// the two, and a length error eps of N or Hn only adds eps^2.  N.H <= 0 (x clamped to 0, s2 = 1) is e >= 2: the
// saturate of e / 2 does it.  Census with this form: 4-10 pixels outside the bare bar, own 0-1 (tools/fast_emul.py).
template <typename T> DEV T one_minus_ndoth2(const V3<T> &N, const V3<T> &H, T rh)
{
    const V3<T> d = { fma_(-H.x, rh, N.x), fma_(-H.y, rh, N.y), fma_(-H.z, rh, N.z) };
    const T h = sat_half(dot3(d, d));
    return h * (2.0f - h);
}
// general L (directional lights are not normalised by the shader, inc_structs.glsl:12-16)
template <typename T> DEV void eval_light(const PxT<T> &q, V3<T> L, V3<T> col, T scale, V3<T> &res)
{
    V3<T> H = { L.x + q.V.x, L.y + q.V.y, L.z + q.V.z };
    // L = -V (light straight behind the pixel as seen from the camera): H = 0 and normalize(H) is NaN in the literal
    // shader; here |H|^2 is floored, Hn and L.H come out 0 and every factor stays finite
    T rh = rsq_(max_(dot3(H, H), (T)(1e-30f)));
    const T LdotH = sat_mul(dot3(L, H), rh);
    const T s2 = one_minus_ndoth2<T>(q.N, H, rh);
    eval_tail<T>(q, sat_fma(q.N.z, L.z, fma_(q.N.y, L.y, q.N.x * L.x)), s2, LdotH, col, scale, res);
}
// Point lights: L and V are unit vectors here, so L.H = L.(L + V) = 1 + L.V = |H|^2 / 2 and clamp(L.H / |H|, 0, 1) =
// |H|^2 rh / 2 - one multiply where the general form takes a dot product (three packed instructions per pair fewer).
// |H|^2 is the literal dot(H, H) of the literal H = L + V, so nothing cancels that the literal form keeps; the two forms
// differ by the unit-length error of L and V, ~1e-7 absolute, in a term that only enters Fresnel's (1 - L.H)^5.
// (N.H keeps its own form, one_minus_ndoth2: the shortcut N.H = (N.L + N.V) / |H| with |H|^2 = 2 + 2 L.V of round 1
// cancelled when the light is nearly opposite the eye and was dropped for that.)
template <typename T> DEV void eval_light_unit(const PxT<T> &q, V3<T> L, V3<T> col, T scale, V3<T> &res)
{
    V3<T> H = { L.x + q.V.x, L.y + q.V.y, L.z + q.V.z };
    const T hh = max_(dot3(H, H), (T)(1e-30f));
    const T rh = rsq_(hh);
    const T LdotH = sat_mul(hh * 0.5f, rh);
    const T s2 = one_minus_ndoth2<T>(q.N, H, rh);
    eval_tail<T>(q, sat_fma(q.N.z, L.z, fma_(q.N.y, L.y, q.N.x * L.x)), s2, LdotH, col, scale, res);
}

DEV void add_directional(const Px &q, DirRec d, float shadow, bool use_shadow, f3 &res)
{
    eval_light<float>(q, f3{ d.a.x, d.a.y, d.a.z }, f3{ d.b.x, d.b.y, d.b.z }, use_shadow ? shadow : 1.0f, res);
}
DEV void add_directional2(const Px2 &q, DirRec d, v2f shadow, bool use_shadow, V3<v2f> &res)
{
    eval_light<v2f>(q, V3<v2f>{ (v2f)(d.a.x), (v2f)(d.a.y), (v2f)(d.a.z) }, V3<v2f>{ (v2f)(d.b.x), (v2f)(d.b.y), (v2f)(d.b.z) },
                    use_shadow ? shadow : (v2f)(1.0f), res);
}

DEV float point_d2(const Px &q, v4f a, f3 &Lv)
{
    Lv = f3{ a.x - q.P.x, a.y - q.P.y, a.z - q.P.z };
    return __builtin_fmaf(Lv.z, Lv.z, __builtin_fmaf(Lv.y, Lv.y, Lv.x * Lv.x));
}
DEV float point_r2(v4f a) { return a.w; }   // prepared record carries r*r
DEV v4f point_rec_a(const ShadeParams &p, uint32_t l) { return ldu(p.point_prep, 2 * l); }
DEV v4f point_rec_b(const ShadeParams &p, uint32_t l) { return ldu(p.point_prep, 2 * l + 1); }
DEV v4f point_rec_a_v(const ShadeParams &p, uint32_t l) { return ((const v4f *)p.point_prep)[2 * l]; }   // per-lane (vector) fetch
DEV v4f point_rec_b_v(const ShadeParams &p, uint32_t l) { return ((const v4f *)p.point_prep)[2 * l + 1]; }
DEV bool point_test(const Px &q, v4f a, f3 &Lv, float &d2)
{
    d2 = point_d2(q, a, Lv);
    return d2 < a.w && q.occ > 0.0f;
}
DEV void point_eval(const Px &q, v4f a, v4f b, f3 Lv, float d2, bool on, f3 &res)
{
    // Out-of-range lanes get att = 0 by SELECT (the window w can round to a tiny non-zero value just outside the
    // radius, and a culled tile list must not change a bit).  Everything the BRDF forms from a finite pixel is
    // finite or clamped (den >= NdotV * alpha > 0, t >= alpha^2 > 0, |L| <= 1e6 |Lv|), so radiance 0 adds exactly 0.
    float inv = rsq_(d2 + 1e-12f);
    V3<float> L = { Lv.x * inv, Lv.y * inv, Lv.z * inv };
    float qq = d2 * b.w, w = sat(__builtin_fmaf(-qq, qq, 1.0f));
    float att = on ? (w * w) * rcp_(d2 + 1e-4f) : 0.0f;
    eval_light_unit<float>(q, L, V3<float>{ b.x, b.y, b.z }, att, res);
}
DEV void add_point(const Px &q, v4f a, v4f b, f3 &res)
{
    f3 Lv;
    float d2;
    const bool on = point_test(q, a, Lv, d2);
    if (__builtin_amdgcn_ballot_w64(on) == 0) return;           // wave-uniform early out
    point_eval(q, a, b, Lv, d2, on, res);
}
// two pixels per lane: the same arithmetic per pixel, bit for bit (packed ops round like the scalar ones)
DEV v2f point_d2_2(const Px2 &q, v4f a, V3<v2f> &Lv)
{
    Lv = V3<v2f>{ a.x - q.P.x, a.y - q.P.y, a.z - q.P.z };
    return fma_(Lv.z, Lv.z, fma_(Lv.y, Lv.y, Lv.x * Lv.x));
}
// Which lanes' pixels are within the light's radius, as two wave masks formed on the scalar side: the compares write SGPR
// pairs, the "anybody?" test is one s_or + s_cmp, and the per-lane select in point_eval2 reads the masks back (inverse
// ballot: no instruction).  Written with bools, the compiler materialised the ballot argument through v_cndmask +
// v_cmp_ne, and the "pixel is lit" half of the condition through two more compares per light: 4 of the 15 VALU
// instructions a tested-and-rejected light cost.  The lit half now sits in the data: before a point-light loop an unlit
// pixel's position is moved out of every light's reach (far_if_unlit), so the distance compare alone rejects it.
struct PointMasks { unsigned long long m0, m1; };
DEV PointMasks point_masks(v2f d2, float r2)
{
    return PointMasks{ __builtin_amdgcn_ballot_w64(d2.x < r2), __builtin_amdgcn_ballot_w64(d2.y < r2) };
}
// |P - c|^2 overflows to +inf for every light (never < r^2); should such a lane ride along in point_eval2 (another lane's
// pixel is in range), its L = Lv * rsq(inf) = -0, every BRDF term stays finite and its radiance is selected to 0:
// an unlit pixel's sum stays exactly 0, as before.
DEV void far_if_unlit(Px2 &q)
{
    const bool a = q.occ.x > 0.0f, b = q.occ.y > 0.0f;
    q.P.x = v2f{ a ? q.P.x.x : 1e30f, b ? q.P.x.y : 1e30f };
    q.P.y = v2f{ a ? q.P.y.x : 1e30f, b ? q.P.y.y : 1e30f };
    q.P.z = v2f{ a ? q.P.z.x : 1e30f, b ? q.P.z.y : 1e30f };
}
// EXTENSION E4 of a pixel pair: ambient * albedo * occlusion, albedo re-decoded from the texel.  The fast kernels START their
// sum with it (fs_deferred_lighting.glsl:61 starts from 0 and the strict kernels add the extension last, as the oracle does;
// the order is the same in every fast kernel, so they still agree bit for bit): occlusion and the texel are then dead before
// the light loops instead of live across them - four registers in kernels that sit at the 128-VGPR limit.
DEV V3<v2f> ambient_start(const Px2 &q, const ShadeParams &p, v2u w2)
{
    if (!p.has_ambient) return V3<v2f>{ (v2f)(0.0f), (v2f)(0.0f), (v2f)(0.0f) };
    const v2f ka = q.occ * (1.0f / 255.0f);
    return V3<v2f>{ (v2f)(p.ambient[0]) * (tof(w2 & 0xFFu) * ka), (v2f)(p.ambient[1]) * (tof((w2 >> 8) & 0xFFu) * ka),
                    (v2f)(p.ambient[2]) * (tof((w2 >> 16) & 0xFFu) * ka) };
}
DEV void point_eval2(const Px2 &q, v4f b, V3<v2f> Lv, v2f d2, bool on0, bool on1, V3<v2f> &res)
{
    v2f inv = rsq_(d2 + 1e-12f);
    V3<v2f> L = { Lv.x * inv, Lv.y * inv, Lv.z * inv };
    v2f qq = d2 * b.w, w = sat_one_minus_mul(qq, qq);
    v2f att = (w * w) * rcp_(d2 + 1e-4f);
    att = v2f{ on0 ? att.x : 0.0f, on1 ? att.y : 0.0f };          // see point_eval
    eval_light_unit<v2f>(q, L, V3<v2f>{ (v2f)(b.x), (v2f)(b.y), (v2f)(b.z) }, att, res);
}
DEV void add_ambient(const Px &q, const ShadeParams &p, uint32_t u2, f3 &res)
{
    const float k = q.occ * (1.0f / 255.0f);       // albedo * occlusion re-decoded from the texel
    res.x = __builtin_fmaf(p.ambient[0], (float)(u2 & 0xFF) * k, res.x);
    res.y = __builtin_fmaf(p.ambient[1], (float)((u2 >> 8) & 0xFF) * k, res.y);
    res.z = __builtin_fmaf(p.ambient[2], (float)((u2 >> 16) & 0xFF) * k, res.z);
}
DEV void shadow_coords(f3 P, const ShadeParams &p, float &u, float &v, float &zr)
{
    float qx = mrow1(p.lvp, 0, P.x, P.y, P.z), qy = mrow1(p.lvp, 1, P.x, P.y, P.z);
    zr = mrow1(p.lvp, 2, P.x, P.y, P.z);
    u = __builtin_fmaf(qx, 0.5f, 0.5f); v = __builtin_fmaf(qy, 0.5f, 0.5f);
}
DEV float pcf_result(int cnt) { return (float)cnt * (1.0f / 9.0f); }
DEV float aces1(float x)
{
    x = fminf(x, 1.0e9f);   // ACESFilm is 1 (clamped) far below this; keeps x*x finite and 1/den normal
    float num = x * __builtin_fmaf(2.51f, x, 0.03f), den = __builtin_fmaf(x, __builtin_fmaf(2.43f, x, 0.59f), 0.14f);
    return clampf(num * rcp_(den), 0.0f, 1.0f);
}
DEV uint32_t srgb8_direct(float c)
{
    float s = c <= 0.0031308f ? 12.92f * c
                              : __builtin_fmaf(1.055f, __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(c) * (1.0f / 2.4f)), -0.055f);
    return (uint32_t)__builtin_fmaf(s, 255.0f, 0.5f);
}
#endif // MIL_STRICT

// ---------------------------------------------------------------------------------
// EXTENSION E3: PCF.  The integer texel / validity logic is identical in both modes.
// ---------------------------------------------------------------------------------
struct ShadowTap { int tx, ty; float zc; bool inside; };

DEV ShadowTap shadow_tap(f3 P, const ShadeParams &p)
{
    float u, v, zr;
    shadow_coords(P, p, u, v, zr);
    ShadowTap t;
    t.inside = (u >= 0.0f && u <= 1.0f && v >= 0.0f && v <= 1.0f) && !(zr > 1.0f);
    float S = (float)p.shadow_size;
    t.tx = t.inside ? (int)floorf(u * S) : 0;
    t.ty = t.inside ? (int)floorf(v * S) : 0;
    t.zc = zr - p.shadow_bias;
    return t;
}
#if !MIL_STRICT
// the two pixels of a lane at once: the same fma chains as shadow_coords (packed ops round like the scalar ones)
DEV void shadow_tap2(const V3<v2f> &P, const ShadeParams &p, ShadowTap &ta, ShadowTap &tb)
{
    auto row = [&](int r) {
        return fma_((v2f)(p.lvp[r]), P.x, fma_((v2f)(p.lvp[4 + r]), P.y, fma_((v2f)(p.lvp[8 + r]), P.z, (v2f)(p.lvp[12 + r]))));
    };
    const v2f u = fma_(row(0), (v2f)(0.5f), (v2f)(0.5f)), v = fma_(row(1), (v2f)(0.5f), (v2f)(0.5f)), zr = row(2);
    const v2f S = (v2f)((float)p.shadow_size);
    const v2f fu = __builtin_elementwise_floor(u * S), fv = __builtin_elementwise_floor(v * S), zc = zr - p.shadow_bias;
    ta.inside = (u.x >= 0.0f && u.x <= 1.0f && v.x >= 0.0f && v.x <= 1.0f) && !(zr.x > 1.0f);
    tb.inside = (u.y >= 0.0f && u.y <= 1.0f && v.y >= 0.0f && v.y <= 1.0f) && !(zr.y > 1.0f);
    ta.tx = ta.inside ? (int)fu.x : 0; ta.ty = ta.inside ? (int)fv.x : 0; ta.zc = zc.x;
    tb.tx = tb.inside ? (int)fu.y : 0; tb.ty = tb.inside ? (int)fv.y : 0; tb.zc = zc.y;
}
#endif
struct __attribute__((packed, aligned(4))) Row3 { float a, b, c; };
DEV float pcf_direct(const ShadowTap &t, const ShadeParams &p)
{
    if (!t.inside) return 1.0f;
    const int S = (int)p.shadow_size;
    int cnt = 0;
    const bool border = t.tx < 1 || t.ty < 1 || t.tx > S - 2 || t.ty > S - 2;
    if (__builtin_amdgcn_ballot_w64(border) == 0) {
        // interior for every active lane: clamp-to-edge is the identity, a row of taps is 3 consecutive floats
        const uint32_t o = (uint32_t)(t.ty - 1) * (uint32_t)S + (uint32_t)(t.tx - 1);     // < 2^28: the host caps the map at 16384^2
        __builtin_assume(o < (1u << 28));
        const float *q = p.shadow_map + o;
        Row3 r0, r1, r2;
        __builtin_memcpy(&r0, q, 12); __builtin_memcpy(&r1, q + (uint32_t)S, 12); __builtin_memcpy(&r2, q + 2u * (uint32_t)S, 12);
        cnt = (t.zc <= r0.a) + (t.zc <= r0.b) + (t.zc <= r0.c) + (t.zc <= r1.a) + (t.zc <= r1.b) + (t.zc <= r1.c) +
              (t.zc <= r2.a) + (t.zc <= r2.b) + (t.zc <= r2.c);
        return pcf_result(cnt);
    }
#pragma unroll
    for (int dy = -1; dy <= 1; dy++) {
        int yy = min(max(t.ty + dy, 0), S - 1);
#pragma unroll
        for (int dx = -1; dx <= 1; dx++) {
            int xx = min(max(t.tx + dx, 0), S - 1);
            cnt += (t.zc <= p.shadow_map[(size_t)yy * S + xx]) ? 1 : 0;
        }
    }
    return pcf_result(cnt);
}

// ---------------------------------------------------------------------------------
// Per-pixel math of the sky (K6) and transparency-resolve (K7) subpasses, shared by their own kernels and by
// the whole-frame launch (emit_pixel below), so that the fused frame equals the chain of separate calls.
// ---------------------------------------------------------------------------------
#define MIL_SKY_DEPTH 0.99999994f   // largest float below 1: LessOrEqual at z = w - 1e-5 passes only texels at the clear depth
#if !MIL_STRICT
// acos on [0, 1] (cos(gamma) is clamped to >= 0.001): sqrt(1 - x) * P7(x), Abramowitz & Stegun 4.4.46, |error| <= 2e-8 + FP32
// rounding -- a dozen instructions instead of the library routine's forty
DEV float acos01(float x)
{
    float q = -0.0012624911f;
    q = __builtin_fmaf(q, x, 0.0066700901f); q = __builtin_fmaf(q, x, -0.0170881256f); q = __builtin_fmaf(q, x, 0.0308918810f);
    q = __builtin_fmaf(q, x, -0.0501743046f); q = __builtin_fmaf(q, x, 0.0889789874f); q = __builtin_fmaf(q, x, -0.2145988016f);
    q = __builtin_fmaf(q, x, 1.5707963050f);
    return sqrt_(1.0f - x) * q;
}
#endif
template <typename PRM>
DEV void sky_radiance(const PRM &p, uint32_t x, uint32_t yg, float d, float r3[3])   // sky_hosek_frag.glsl:28-60 (yg: frame row)
{
    const float *A = p.hw[0], *B = p.hw[1], *C = p.hw[2], *D = p.hw[3], *E = p.hw[4], *F = p.hw[5], *G = p.hw[6], *H = p.hw[7],
                *I = p.hw[8], *Z = p.hw[9], *sun = p.hw[10];
#if MIL_STRICT
    float cx = ((float)x + 0.5f) / p.res_x, cy = ((float)yg + 0.5f) / p.res_y;
    float px = cx * 2.0f - 1.0f, py = cy * 2.0f - 1.0f;
    float vx = mrow(p.inv_proj, 0, px, py, d, 1.0f), vy = mrow(p.inv_proj, 1, px, py, d, 1.0f);
    float vz = mrow(p.inv_proj, 2, px, py, d, 1.0f), vw = mrow(p.inv_proj, 3, px, py, d, 1.0f);
    vx = vx / vw; vy = vy / vw; vz = vz / vw; vw = vw / vw;
    f3 P = { mrow(p.inv_view, 0, vx, vy, vz, vw), mrow(p.inv_view, 1, vx, vy, vz, vw), mrow(p.inv_view, 2, vx, vy, vz, vw) };
    f3 vd = normalize3(f3{ P.x - p.cam[0], P.y - p.cam[1], P.z - p.cam[2] });                    // :56-58
    const float cg = fmaxf(0.001f, dot3(f3{ sun[0], sun[1], sun[2] }, vd));                      // :46
    const float ct = fmaxf(0.001f, (vd.x * 0.0f + vd.y * 1.0f) + vd.z * 0.0f);                   // :47
    const float gamma = (float)acos((double)cg);                                                 // :49 (exp / pow / acos in double, as the oracle)
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float chi = (1.0f + cg * cg) / (float)pow((double)((1.0f + H[c] * H[c]) - (2.0f * H[c]) * cg), 1.5);   // :39
        const float left = 1.0f + A[c] * (float)exp((double)(B[c] / (ct + 0.01f)));              // :41
        const float right = (((C[c] + D[c] * (float)exp((double)(E[c] * gamma))) + (F[c] * cg) * cg) + G[c] * chi) + I[c] * sqrtf(ct);
        r3[c] = ((left * right) * Z[c]) * 0.05f;                                                 // :51, :58
    }
#else
    const float xf = (float)x, yf = (float)yg;
    const float *m = p.pos_m;
    // direction only: the w divide of PositionFromDepth is a positive scale the normalisation removes
    f3 v = { __builtin_fmaf(m[0], xf, __builtin_fmaf(m[1], yf, __builtin_fmaf(m[2], d, m[3]))),
             __builtin_fmaf(m[4], xf, __builtin_fmaf(m[5], yf, __builtin_fmaf(m[6], d, m[7]))),
             __builtin_fmaf(m[8], xf, __builtin_fmaf(m[9], yf, __builtin_fmaf(m[10], d, m[11]))) };
    const float vw = __builtin_fmaf(m[12], xf, __builtin_fmaf(m[13], yf, __builtin_fmaf(m[14], d, m[15])));
    const float rn = rsq_(dot3(v, v)) * (vw < 0.0f ? -1.0f : 1.0f);
    const f3 vd = { v.x * rn, v.y * rn, v.z * rn };
    const float cg = fmaxf(0.001f, dot3(f3{ sun[0], sun[1], sun[2] }, vd));
    const float ct = fmaxf(0.001f, vd.y);
    const float gamma = acos01(cg);
    const float cg2 = cg * cg, sct = sqrt_(ct), ict = rcp_(ct + 0.01f);
    const float LOG2E = 1.4426950408889634f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float t = __builtin_fmaf(-2.0f * H[c], cg, __builtin_fmaf(H[c], H[c], 1.0f));       // 1 + H^2 - 2 H cos(gamma)
        const float rt = rsq_(t);
        const float chi = (1.0f + cg2) * (rt * rt * rt);                                          // t^-1.5 with one transcendental
        const float left = __builtin_fmaf(A[c], __builtin_amdgcn_exp2f(B[c] * ict * LOG2E), 1.0f);
        const float right = __builtin_fmaf(I[c], sct, __builtin_fmaf(G[c], chi, __builtin_fmaf(F[c], cg2,
                            __builtin_fmaf(D[c], __builtin_amdgcn_exp2f(E[c] * gamma * LOG2E), C[c]))));
        r3[c] = left * right * (Z[c] * 0.05f);
    }
#endif
}
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
DEV bool resolve_untouched(h4 a, float r)     // nothing transparent here: 0 * 0 + hdr * 1 = hdr exactly
{
    return r == 1.0f && (float)a.x == 0.0f && (float)a.y == 0.0f && (float)a.z == 0.0f;
}
DEV v4f resolve_blend(h4 a, float r, v4f h)    // fs_mcguire13_resolve.glsl:8-13 + blend mcguire13/mod.rs:108-120
{
    const float ar = (float)a.x, ag = (float)a.y, ab = (float)a.z, aa = (float)a.w;
    const float den = fminf(5e4f, fmaxf(1e-4f, aa)), om = 1.0f - r;
#if MIL_STRICT
    return v4f{ (ar / den) * om + h.x * r, (ag / den) * om + h.y * r, (ab / den) * om + h.z * r, r * om + h.w * r };
#else
    const float k = rcp_(den) * om;
    return v4f{ __builtin_fmaf(ar, k, h.x * r), __builtin_fmaf(ag, k, h.y * r), __builtin_fmaf(ab, k, h.z * r), __builtin_fmaf(r, om, h.w * r) };
#endif
}

// ---------------------------------------------------------------------------------
// output encode (fs_tonemap.glsl:23-27 + attachment formats)
// ---------------------------------------------------------------------------------
DEV uint32_t ufloat_enc(float x, int mbits)
{
    uint32_t bits = __float_as_uint(x);
    int sh = 23 - mbits;
    if (x >= 6.103515625e-05f) {
        uint32_t r = bits + ((1u << (sh - 1)) - 1u) + ((bits >> sh) & 1u);
        return (r >> sh) - ((127u - 15u) << mbits);
    }
    return (uint32_t)__builtin_rintf(x * (float)(1u << (14 + mbits)));
}
DEV uint32_t pack_r11g11b10f(float r, float g, float b)
{
    return ufloat_enc(r, 6) | (ufloat_enc(g, 6) << 11) | (ufloat_enc(b, 5) << 22);
}
template <int FMT>
DEV uint32_t encode_ldr(float r, float g, float b, const uint8_t *lut11, const uint8_t *lut10)
{
    r = aces1(r); g = aces1(g); b = aces1(b);
    if (FMT == FMT_LDR) return pack_r11g11b10f(r, g, b);
    if (FMT == FMT_SRGB8) return srgb8_direct(b) | (srgb8_direct(g) << 8) | (srgb8_direct(r) << 16) | 0xFF000000u;
    uint32_t cr = ufloat_enc(r, 6), cg = ufloat_enc(g, 6), cb = ufloat_enc(b, 5);
    return (uint32_t)lut10[cb] | ((uint32_t)lut11[cg] << 8) | ((uint32_t)lut11[cr] << 16) | 0xFF000000u;
}
template <int FMT>
DEV void store_pixel(const ShadeParams &p, uint32_t x, uint32_t yo, f3 res, float alpha = 1.0f)   // yo: output row
{
    size_t o = (size_t)yo * p.out_pitch + x;
    if (FMT == FMT_HDR) {
        __builtin_nontemporal_store(v4f{ res.x, res.y, res.z, alpha }, (v4f *)p.out + o);   // fs_deferred_lighting.glsl:66
        // the tonemap subpass of the same pixel in the same launch (mil_extras.ldr_out): what k_tonemap makes of the stored value
        if (p.out2) __builtin_nontemporal_store(encode_ldr<FMT_SRGB8>(res.x, res.y, res.z, nullptr, nullptr), p.out2 + (size_t)yo * p.out2_pitch + x);
    } else {
        __builtin_nontemporal_store(encode_ldr<FMT>(res.x, res.y, res.z, p.lut11, p.lut10), (uint32_t *)p.out + o);
    }
}
// A black pixel (unlit tile) stored through pointers the caller fetched earlier: its encodings are constants
// (ACES(0) = 0: R11G11B10F code 0, sRGB8 0xFF000000), so no uniform is read at the store.
template <int FMT>
DEV void store_zero(void *out, uint32_t out_pitch, uint32_t x, uint32_t y, uint32_t *out2 = nullptr, uint32_t out2_pitch = 0)
{
    const size_t o = (size_t)y * out_pitch + x;
    if (FMT == FMT_HDR) {
        __builtin_nontemporal_store(v4f{ 0.0f, 0.0f, 0.0f, 1.0f }, (v4f *)out + o);
        if (out2) __builtin_nontemporal_store(0xFF000000u, out2 + (size_t)y * out2_pitch + x);
    }
    else __builtin_nontemporal_store(FMT == FMT_LDR ? 0u : 0xFF000000u, (uint32_t *)out + o);
}
// Whole-frame launch (FRAME, mil_shade_frame): what the later hdr-writing subpasses of the reference's render pass do
// to this pixel happens in registers before the one store - sky where the depth is still the clear value
// (render/mod.rs:221-226), then the transparency resolve blend (render/mod.rs:282-294), then the tonemap/pack of FMT
// (render/mod.rs:297-309).  hdr is a transient attachment in the reference (pbr/mod.rs:76-95); here it never exists.
DEV v4f frame_value(const ShadeParams &p, uint32_t x, uint32_t y, RowMap rm, float depth, f3 res, const uint32_t *tile)
{
    v4f h = { res.x, res.y, res.z, 1.0f };
    if (p.has_sky) {
        const bool sky = depth >= MIL_SKY_DEPTH;
        if (__builtin_amdgcn_ballot_w64(sky) != 0) {
            float r3[3];
            sky_radiance(p, x, y + rm.g, depth, r3);
            if (sky) h = v4f{ r3[0], r3[1], r3[2], 1.0f };
        }
    }
    if (p.accum) {
        h4 a;
        float r;
        if (tile) {             // k_tiled staged the tile's accum / revealage texels in LDS (LDS-DMA, with the G-buffer planes)
            const uint32_t c = x & 15u, row = y & 15u;
            typedef uint32_t u2_t __attribute__((ext_vector_type(2)));
            const u2_t aw = *(const u2_t *)(tile + 2 * (row * 16 + c));
            const uint32_t rw = tile[512 + row * 8 + (c >> 1)];
            union { u2_t u; h4 h; } cv; cv.u = aw; a = cv.h;
            union { unsigned short s; _Float16 h; } rv; rv.s = (unsigned short)(rw >> (16 * (c & 1u)));
            r = (float)rv.h;
        } else {
            a = __builtin_nontemporal_load((const h4 *)p.accum + (size_t)y * p.accum_pitch + x);
            r = (float)__builtin_nontemporal_load((const _Float16 *)p.reveal + (size_t)y * p.reveal_pitch + x);
        }
#if MIL_STRICT
        h = resolve_blend(a, r, h);
#else
        if (__builtin_amdgcn_ballot_w64(!resolve_untouched(a, r)) != 0) h = resolve_blend(a, r, h);
#endif
    }
    return h;
}
template <int FMT, bool FRAME>
DEV void emit_pixel(const ShadeParams &p, uint32_t x, uint32_t y, RowMap rm, float depth, f3 res, const uint32_t *tile = nullptr)
{
    if (!FRAME) { store_pixel<FMT>(p, x, y + rm.o, res); return; }
    const v4f h = frame_value(p, x, y, rm, depth, res, tile);
    store_pixel<FMT>(p, x, y + rm.o, f3{ h.x, h.y, h.z }, h.w);
}
#if !MIL_STRICT
// The two pixels of a lane at once (fast mode, no later subpass): the ACES fit and the sRGB OETF of encode_ldr<FMT_SRGB8> on
// the packed pairs the shading already holds - one v_pk instruction for both pixels where the per-pixel form issues two
// (rcp / log / exp stay per pixel).  The same operations in the same order, and packed ops round like scalar ones: the same
// bits as emit_pixel twice (tests: fused == unfused, dual output == mil_tonemap of the stored hdr).  The constants are
// pinned where they are used: as loop invariants the compiler held them in SGPRs across the light loops and spilled.
DEV v2f kpin(float k) { asm volatile("" : "+s"(k)); return (v2f)(k); }
DEV v2u srgb8_pair(v2f x)
{
    x = v2f{ fminf(x.x, 1.0e9f), fminf(x.y, 1.0e9f) };
    const v2f num = x * fma_(kpin(2.51f), x, kpin(0.03f)), den = fma_(x, fma_(kpin(2.43f), x, kpin(0.59f)), kpin(0.14f));
    const v2f c = v2f{ clampf(num.x * rcp_(den.x), 0.0f, 1.0f), clampf(num.y * rcp_(den.y), 0.0f, 1.0f) };
    const v2f lg = v2f{ __builtin_amdgcn_logf(c.x), __builtin_amdgcn_logf(c.y) } * kpin(1.0f / 2.4f);
    const v2f pw = fma_(kpin(1.055f), v2f{ __builtin_amdgcn_exp2f(lg.x), __builtin_amdgcn_exp2f(lg.y) }, kpin(-0.055f));
    const v2f lin = kpin(12.92f) * c;
    const v2f s = v2f{ c.x <= 0.0031308f ? lin.x : pw.x, c.y <= 0.0031308f ? lin.y : pw.y };
    const v2f q = fma_(s, kpin(255.0f), (v2f)(0.5f));
    return v2u{ (uint32_t)q.x, (uint32_t)q.y };
}
template <int FMT>
DEV void emit_pair(const ShadeParams &p, uint32_t x, uint32_t ya, uint32_t yb, RowMap rm, bool inA, bool inB, V3<v2f> r)
{
    if (FMT == FMT_HDR || FMT == FMT_SRGB8) {
        const size_t oa = (size_t)(ya + rm.o) * p.out_pitch + x, ob = (size_t)(yb + rm.o) * p.out_pitch + x;
        if (FMT == FMT_HDR) {
            if (inA) __builtin_nontemporal_store(v4f{ r.x.x, r.y.x, r.z.x, 1.0f }, (v4f *)p.out + oa);
            if (inB) __builtin_nontemporal_store(v4f{ r.x.y, r.y.y, r.z.y, 1.0f }, (v4f *)p.out + ob);
            if (!p.out2) return;
        }
        const v2u cr = srgb8_pair(r.x), cg = srgb8_pair(r.y), cb = srgb8_pair(r.z);
        const v2u w = cb | (cg << 8) | (cr << 16) | 0xFF000000u;
        uint32_t *const o = FMT == FMT_HDR ? p.out2 : (uint32_t *)p.out;
        const uint32_t pitch = FMT == FMT_HDR ? p.out2_pitch : p.out_pitch;
        if (inA) __builtin_nontemporal_store(w.x, o + (size_t)(ya + rm.o) * pitch + x);
        if (inB) __builtin_nontemporal_store(w.y, o + (size_t)(yb + rm.o) * pitch + x);
    } else {
        if (inA) store_pixel<FMT>(p, x, ya + rm.o, f3{ r.x.x, r.y.x, r.z.x });
        if (inB) store_pixel<FMT>(p, x, yb + rm.o, f3{ r.x.y, r.y.y, r.z.y });
    }
}
#endif
// the same store through pointers fetched earlier (see k_tiled's unlit tiles)
template <int FMT>
DEV void store_pixel_to(void *out, uint32_t out_pitch, const uint8_t *lut11, const uint8_t *lut10, uint32_t x, uint32_t y, v4f h,
                        uint32_t *out2 = nullptr, uint32_t out2_pitch = 0)
{
    const size_t o = (size_t)y * out_pitch + x;
    if (FMT == FMT_HDR) {
        __builtin_nontemporal_store(h, (v4f *)out + o);
        if (out2) __builtin_nontemporal_store(encode_ldr<FMT_SRGB8>(h.x, h.y, h.z, nullptr, nullptr), out2 + (size_t)y * out2_pitch + x);
    }
    else __builtin_nontemporal_store(encode_ldr<FMT>(h.x, h.y, h.z, lut11, lut10), (uint32_t *)out + o);
}

// ---------------------------------------------------------------------------------
// K1: streaming kernel.  Block = 64 x 4 threads covering 64 x 16 pixels; each wave owns a
// 64-pixel row segment (one 256-byte load per plane, one contiguous 1 KiB HDR store) on four
// rows, issues all 16 loads first and then shades the four pixels one after the other.
// Handles directional lights, shadow by direct taps, brute-force point lights, ambient.
// ---------------------------------------------------------------------------------
template <int FMT, bool EXT, bool FRAME>
DEV void shade_stream_pixel(const ShadeParams &p, uint32_t x, uint32_t y, RowMap rm, uint32_t u1, uint32_t u2, uint32_t u3, float depth, DirRec d0)
{
    Px q;
    setup_pixel(q, p, pixel_position(p, x, y + rm.g, depth), u1, u2, u3);
    f3 res = { 0.0f, 0.0f, 0.0f };
#if !MIL_STRICT
    if (EXT && p.has_ambient) {        // the fast kernels start their sum with the ambient term (ambient_start)
        const float k = q.occ * (1.0f / 255.0f);
        res = f3{ p.ambient[0] * ((float)(u2 & 0xFF) * k), p.ambient[1] * ((float)((u2 >> 8) & 0xFF) * k), p.ambient[2] * ((float)((u2 >> 16) & 0xFF) * k) };
    }
#endif
    float shadow = 1.0f;
    const bool sh = EXT && p.has_shadow;
    if (sh && p.n_dir > 0) shadow = pcf_direct(shadow_tap(q.P, p), p);
    DirRec d = d0;
    for (uint32_t l = 0; l < p.n_dir; l++) {                 // fs_deferred_lighting.glsl:62-64
        const DirRec dn = dir_rec(p, min(l + 1, p.n_dir - 1)); // next record requested while this one is evaluated
        add_directional(q, d, shadow, sh && l == 0, res);
        d = dn;
    }
    if (EXT) {
        for (uint32_t l = 0; l < p.n_point; l++) add_point(q, point_rec_a(p, l), point_rec_b(p, l), res);
#if MIL_STRICT
        if (p.has_ambient) add_ambient(q, p, u2, res);
#endif
    }
    emit_pixel<FMT, FRAME>(p, x, y, rm, depth, res);
}

template <typename T> DEV T sel4(uint32_t j, T a, T b, T c, T d) { return j == 0 ? a : (j == 1 ? b : (j == 2 ? c : d)); }

template <int FMT, int NPX, bool EXT, bool FRAME>
__global__ __launch_bounds__(256) void k_stream(const ShadeParams p)
{
    // Coordinates are clamped for the loads instead of branched around: the texel loads are the
    // first thing the wave issues (one kernarg fetch away), the bounds test comes afterwards.
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, xc = min(x, p.width - 1), ylast = p.height - 1;
    if (NPX == 1) {
        const uint32_t y = blockIdx.y * 4 + threadIdx.y;
        const size_t i = (size_t)min(y, ylast) * p.pitch + xc;
        const uint32_t u1 = __builtin_nontemporal_load(p.normal + i), u2 = __builtin_nontemporal_load(p.albedo + i);
        const uint32_t u3 = __builtin_nontemporal_load(p.rm + i);
        const float depth = __builtin_nontemporal_load(p.depth + i);
        const DirRec d0 = dir_rec(p, 0);
        if (x >= p.width || y > ylast) return;
        const RowMap rm = row_map(p, blockIdx.y * 4);
#if !MIL_STRICT
        // extensions on: a wave whose 64 pixels all have occlusion 0 (cleared G-buffer) writes its exact zeros
        // and skips shadow taps, point lights and BRDF (the same early-out as k_tiled)
        if (EXT && __builtin_amdgcn_ballot_w64((u2 >> 24) != 0) == 0) {
            emit_pixel<FMT, FRAME>(p, x, y, rm, depth, f3{ 0.0f, 0.0f, 0.0f });
            return;
        }
#endif
        shade_stream_pixel<FMT, EXT, FRAME>(p, x, y, rm, u1, u2, u3, depth, d0);
        return;
    }
    const uint32_t y0 = blockIdx.y * 16 + threadIdx.y;     // this lane: rows y0, y0+4, y0+8, y0+12
    uint32_t u1[4], u2[4], u3[4];
    float dz[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const size_t i = (size_t)min(y0 + 4 * j, ylast) * p.pitch + xc;
        u1[j] = __builtin_nontemporal_load(p.normal + i); u2[j] = __builtin_nontemporal_load(p.albedo + i);
        u3[j] = __builtin_nontemporal_load(p.rm + i); dz[j] = __builtin_nontemporal_load(p.depth + i);
    }
    const DirRec d0 = dir_rec(p, 0);
    if (x >= p.width) return;
    const RowMap rm = row_map(p, blockIdx.y * 16);
#pragma unroll 1
    for (uint32_t j = 0; j < 4; j++) {
        const uint32_t y = y0 + 4 * j;
        if (y > ylast) break;
        shade_stream_pixel<FMT, EXT, FRAME>(p, x, y, rm, sel4(j, u1[0], u1[1], u1[2], u1[3]), sel4(j, u2[0], u2[1], u2[2], u2[3]),
                                     sel4(j, u3[0], u3[1], u3[2], u3[3]), sel4(j, dz[0], dz[1], dz[2], dz[3]), d0);
    }
}

#if !MIL_STRICT
// K1, two pixels per lane: the lane shades (x, y) and (x, y + 4) together in packed FP32 (the template instantiation K2
// uses: v_pk_fma_f32 / v_pk_mul_f32, one issue slot for both pixels).  The one-pixel kernel spends 72 % of its SIMD
// cycles in the VALU pipe beside its HBM stream; halving the issue count leaves the stream as the only bound.  Block =
// 64 x 4 lanes over 64 x 8 pixels: every wave still reads whole 256-byte row segments and writes whole 1 KiB HDR rows.
// A wave whose 128 pixels all have occlusion 0 (cleared G-buffer, render/mod.rs:129-131) stores its exact zeros unshaded.
template <int FMT, bool EXT, bool FRAME>
__global__ __launch_bounds__(256) void k_stream2(const ShadeParams p)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, xc = min(x, p.width - 1), ylast = p.height - 1;
    const uint32_t ya = blockIdx.y * 8 + threadIdx.y, yb = ya + 4;
    const uint32_t ia = min(ya, ylast) * p.pitch + xc, ib = min(yb, ylast) * p.pitch + xc;   // < 2^30 (checked on the host)
    __builtin_assume(ia < (1u << 30)); __builtin_assume(ib < (1u << 30));
    const v2u w2 = { __builtin_nontemporal_load(p.albedo + ia), __builtin_nontemporal_load(p.albedo + ib) };
    const v2u u1 = { __builtin_nontemporal_load(p.normal + ia), __builtin_nontemporal_load(p.normal + ib) };
    const v2u u3 = { __builtin_nontemporal_load(p.rm + ia), __builtin_nontemporal_load(p.rm + ib) };
    const v2f depth = { __builtin_nontemporal_load(p.depth + ia), __builtin_nontemporal_load(p.depth + ib) };
    if (x >= p.width || ya > ylast) return;
    const RowMap rm = row_map(p, blockIdx.y * 8);
    const bool inB = yb <= ylast;
    const bool litA = (w2.x >> 24) != 0, litB = inB && (w2.y >> 24) != 0;
    if (__builtin_amdgcn_ballot_w64(litA || litB) == 0) {
        emit_pixel<FMT, FRAME>(p, x, ya, rm, depth.x, f3{ 0.0f, 0.0f, 0.0f });
        if (inB) emit_pixel<FMT, FRAME>(p, x, yb, rm, depth.y, f3{ 0.0f, 0.0f, 0.0f });
        return;
    }
    Px2 q;
    setup_pixel_t<v2f, v2u>(q, pixel_position_t<v2f>(p, (v2f)((float)x), v2f{ (float)(ya + rm.g), (float)(yb + rm.g) }, depth), u1, w2, u3);
    V3<v2f> r = { (v2f)(0.0f), (v2f)(0.0f), (v2f)(0.0f) };
    if (EXT) r = ambient_start(q, p, w2);
    const bool sh = EXT && p.has_shadow && p.n_dir > 0;
    v2f shadow = { 1.0f, 1.0f };
    if (sh) {
        ShadowTap ta, tb;
        shadow_tap2(q.P, p, ta, tb);
        if (litA) shadow.x = pcf_direct(ta, p);
        if (litB) shadow.y = pcf_direct(tb, p);
    }
    if (EXT && p.n_point) far_if_unlit(q);                   // after the taps (they read P), before the loops (occlusion dies here)
    for (uint32_t l = 0; l < p.n_dir; l++) add_directional2(q, dir_rec(p, l), shadow, sh && l == 0, r);   // fs_deferred_lighting.glsl:62-64
    if (EXT) {
        for (uint32_t l = 0; l < p.n_point; l++) {
            const v4f a = point_rec_a(p, l);
            V3<v2f> Lv;
            const v2f d2 = point_d2_2(q, a, Lv);
            const PointMasks pm = point_masks(d2, a.w);
            if ((pm.m0 | pm.m1) == 0) continue;
            const bool on0 = __builtin_amdgcn_inverse_ballot_w64(pm.m0), on1 = __builtin_amdgcn_inverse_ballot_w64(pm.m1);
            point_eval2(q, point_rec_b(p, l), Lv, d2, on0, on1, r);
        }
    }
    if (!FRAME) emit_pair<FMT>(p, x, ya, yb, rm, true, inB, r);
    else {
        emit_pixel<FMT, FRAME>(p, x, ya, rm, depth.x, f3{ r.x.x, r.y.x, r.z.x });
        if (inB) emit_pixel<FMT, FRAME>(p, x, yb, rm, depth.y, f3{ r.x.y, r.y.y, r.z.y });
    }
}
#endif

// ---------------------------------------------------------------------------------
// K2: tiled kernel.  ONE WAVE PER 16x16 SCREEN TILE, no workgroup barrier anywhere: a
// 256-thread workgroup is four independent waves on four horizontally adjacent tiles (so
// the workgroup reads whole 256-byte row segments).  Lane l owns pixels
// (l & 15, (l >> 4) + 4 j), j = 0..3.
//  1. the tile's 4 planes x 256 texels go HBM -> LDS by 16 LDS-DMA loads per wave
//     (global_load_lds_dword: no staging registers, each texel fetched exactly once), issued
//     together with the first 64 cull records;
//  2. world positions of the lane's 4 pixels -> the tile's AABB of lit pixels by DPP row
//     reductions + v_readlane (no LDS round trip);
//  3. point lights, CHUNK at a time: 64 lights per step (next step's records in flight) are
//     tested against the AABB and survivors appended to the wave's LDS list with ballot +
//     mbcnt prefix, so the list stays in ascending light order and the summation order
//     equals the brute-force loop's (culled == unculled, bit for bit);
//  4. shadow (optional staging): when the tile's texel rectangle (+1 apron) fits, the wave
//     stages it in LDS with clamp-to-edge baked in and the 3x3 taps read LDS;
//  5. the list is consumed 64 entries at a time: lane k fetches entry k's light record into
//     8 VGPRs, and the per-light loop broadcasts record k with v_readlane (no memory latency
//     inside the loop; the radiance half is only broadcast when some lane is in range);
//  6. four passes (j) shade 64 pixels each from the LDS texels: directional lights, then
//     the list.
// ---------------------------------------------------------------------------------
template <int CTRL> DEV float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> DEV int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
// 0xB1 quad_perm[1,0,3,2], 0x4E quad_perm[2,3,0,1], 0x141 row_half_mirror, 0x140 row_mirror:
// after the four steps every lane holds its 16-lane row's result; rows are combined by readlane.
DEV float uniform_f(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
DEV float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
DEV float wave_min(float v)
{
    v = fminf(v, dpp_f<0xB1>(v)); v = fminf(v, dpp_f<0x4E>(v)); v = fminf(v, dpp_f<0x141>(v)); v = fminf(v, dpp_f<0x140>(v));
    return fminf(fminf(lane_f(v, 0), lane_f(v, 16)), fminf(lane_f(v, 32), lane_f(v, 48)));
}
DEV float wave_max(float v)
{
    v = fmaxf(v, dpp_f<0xB1>(v)); v = fmaxf(v, dpp_f<0x4E>(v)); v = fmaxf(v, dpp_f<0x141>(v)); v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
// The tile AABB: three wave-wide minima and three maxima at once.  Each step is ONE v_min/v_max with a DPP operand
// (xor 1, xor 2, half mirror, mirror: every lane holds its 16-lane row's result; row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2 and 3: lane 63 holds the wave's), the six independent chains interleaved so that no
// DPP read follows the write of its register by fewer than five instructions (the hazard needs two wait states);
// 36 + 6 instructions where the builtin form (v_mov_dpp + canonicalising v_max + v_min, readlane x 4) took 134.
DEV void wave_aabb(f3 &lo, f3 &hi)
{
#define MIL_RED_STEP(CTRL)                                             \
    "v_min_f32_dpp %0, %0, %0 " CTRL "\n v_min_f32_dpp %1, %1, %1 " CTRL "\n v_min_f32_dpp %2, %2, %2 " CTRL "\n" \
    "v_max_f32_dpp %3, %3, %3 " CTRL "\n v_max_f32_dpp %4, %4, %4 " CTRL "\n v_max_f32_dpp %5, %5, %5 " CTRL "\n"
    asm volatile("s_nop 1\n"     // the operands may have been written by the instruction just before this block
                 MIL_RED_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 MIL_RED_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 MIL_RED_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 MIL_RED_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 MIL_RED_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 MIL_RED_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 : "+v"(lo.x), "+v"(lo.y), "+v"(lo.z), "+v"(hi.x), "+v"(hi.y), "+v"(hi.z));
#undef MIL_RED_STEP
    lo = f3{ lane_f(lo.x, 63), lane_f(lo.y, 63), lane_f(lo.z, 63) };
    hi = f3{ lane_f(hi.x, 63), lane_f(hi.y, 63), lane_f(hi.z, 63) };
}
DEV int wave_mini(int v)
{
    v = min(v, dpp_i<0xB1>(v)); v = min(v, dpp_i<0x4E>(v)); v = min(v, dpp_i<0x141>(v)); v = min(v, dpp_i<0x140>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
DEV int wave_maxi(int v)
{
    v = max(v, dpp_i<0xB1>(v)); v = max(v, dpp_i<0x4E>(v)); v = max(v, dpp_i<0x141>(v)); v = max(v, dpp_i<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// squared distance from a light centre to the tile AABB, formed with the same operation
// sequence as point_d2 so that  pixel in range  =>  box in range  holds exactly under
// rounding (every step is monotone in |component|).
DEV float box_d2(v4f a, f3 lo, f3 hi)
{
    float dx = fmaxf(fmaxf(lo.x - a.x, a.x - hi.x), 0.0f);
    float dy = fmaxf(fmaxf(lo.y - a.y, a.y - hi.y), 0.0f);
    float dz = fmaxf(fmaxf(lo.z - a.z, a.z - hi.z), 0.0f);
#if MIL_STRICT
    return (dx * dx + dy * dy) + dz * dz;
#else
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
#endif
}

// Per-wave LDS carve of k_tiled (bytes), shared by the launcher and the kernel.
struct TiledLds { uint32_t list, stash, shadow, resolve, pos, per_wave; };
static __host__ __device__ __forceinline__ TiledLds tiled_lds(uint32_t n_point, bool stage, bool resolve)
{
    TiledLds l;
    const uint32_t cap = n_point < (uint32_t)CHUNK ? ((n_point + 63u) & ~63u) : (uint32_t)CHUNK;   // list entries (u16)
    l.list = 4096u;                                                   // after the 4 x 4 x 64 texel dwords
    l.stash = l.list + ((cap * 2u + 2u + 15u) & ~15u);              // + one dump entry behind the list (branch-free append)
    l.shadow = l.stash + (n_point > (uint32_t)CHUNK ? 768u * 4u : 0u);
    l.resolve = l.shadow + (stage ? (uint32_t)SH_CAP * 4u : 0u);
    l.pos = l.resolve + (resolve ? 2560u : 0u);                       // whole frame: 256 accum texels (8 B) + 256 revealage (2 B)
    // fast mode: P - cameraPosition of the tile's 256 pixels, formed once for the box of the lit pixels and read back by the
    // two shading passes (12 floats per lane) instead of being evaluated again
    // (not beside the resolve texels of a whole-frame launch: 10 KB per wave would leave three workgroups per CU where
    // the registers allow four, measured +9 % on mil_shade_frame; there the passes evaluate the positions again)
    l.per_wave = l.pos + ((!MIL_STRICT && (n_point > 0u || stage) && !resolve) ? 3072u : 0u);
    return l;
}

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
DEV v4f bcast4(v4f v, int k)
{
    return v4f{ lane_f(v.x, k), lane_f(v.y, k), lane_f(v.z, k), lane_f(v.w, k) };
}

#if !MIL_STRICT
// A point light's 32-byte record through the SCALAR cache, issued by hand.  The fast kernels walk a tile's light list one
// light at a time with the record in SGPRs (wave-uniform operands of the packed arithmetic).  Broadcasting it out of a
// lane's VGPRs cost 8 v_readlane per light (a third of a rejected light's instructions); a compiler-scheduled scalar load
// is sunk to its first use and exposes the whole scalar-cache latency per light.  srec_load issues s_load_dwordx8 by inline
// asm - the compiler does not know the destination is still in flight - and srec_wait (s_waitcnt lgkmcnt(0)) comes before
// the source touches it and before a loop's back edge.  That the COMPILER does not touch it in between either (copy, spill)
// is proved on the assembly by tools/check_srec.py (tests/test_abi.py runs it).  Loads only: nothing is written through
// the scalar cache.  Same records, same arithmetic, same bits as the broadcast form.
typedef float f8v __attribute__((ext_vector_type(8)));
struct SRec { f8v f; };
DEV SRec srec_load(uint64_t base, uint32_t li)
{
    SRec r;
    const uint64_t addr = base + (uint64_t)li * 32u;
    // early clobber: the destination never overlaps the address pair (a replayed load - XNACK - would read a clobbered address)
    asm volatile("s_load_dwordx8 %0, %1, 0x0 ; SREC_LOAD" : "=&s"(r.f) : "s"(addr));
    return r;
}
DEV void srec_wait() { asm volatile("s_waitcnt lgkmcnt(0) ; SREC_WAIT" ::: "memory"); }
#endif

// The kernel-argument block again, through a pointer the optimiser cannot see through: uniforms read
// through it are re-fetched from the scalar cache in the phase that uses them instead of being held in
// SGPRs across the whole kernel (which spills them into VGPR lanes).
typedef const __attribute__((address_space(4))) ShadeParams *kparams_t;
DEV const ShadeParams &kernarg_params()
{
    kparams_t pp = (kparams_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(pp));
    return *(const ShadeParams *)pp;
}

// STAGE: the LDS-staged PCF taps (MIL_FLAG_SHADOW_LDS) as an instantiation of their own - the rectangle bookkeeping they
// keep live across the passes would otherwise cost every launch its registers.  Without it a wave needs <= 96 VGPRs (or
// spills three dwords per lane, hdr output), which admits five waves per SIMD instead of four (measured -1 ... -3.5 %).
template <int FMT, bool FRAME, bool STAGE>
__global__ __launch_bounds__(256, STAGE ? 4 : 5) void k_tiled(const ShadeParams p_arg)
{
    const ShadeParams &p = kernarg_params();
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t wpb = blockDim.x >> 6;               // waves (= tiles) per workgroup
    // Everything the first phase reads, requested together: uniforms come through an opaque pointer (see kernarg_params),
    // so read one by one behind branches they cost a scalar-memory round trip each - six before the first texel was
    // requested, six more in an unlit tile's store path; a sky tile's wave spent most of its 2.7 us waiting on them.
    const uint32_t k_n_point = p.n_point, k_has_shadow = p.has_shadow, k_n_dir = p.n_dir, k_shadow_lds = p.shadow_lds;
    const uint32_t k_tune = p.tune, k_tiles_x = p.tiles_x, k_tiles_y = p.tiles_y, k_width = p.width, k_height = p.height, k_pitch = p.pitch;
    const uint32_t k_resolve_dma = FRAME ? p.resolve_dma : 0u, k_has_sky = FRAME ? p.has_sky : 0u;
    const uint32_t k_y_offset = p.y_offset, k_band_shift = p.band_shift, k_band_extra = p.band_extra, k_out_band_extra = p.out_band_extra;
    const uint32_t *const k_albedo = p.albedo;
    const uint16_t *const k_accum = FRAME ? p.accum : nullptr;
    uint32_t *const k_stamps = p.tile_stamps;
    asm volatile("" :: "s"(k_n_point), "s"(k_has_shadow), "s"(k_n_dir), "s"(k_shadow_lds), "s"(k_tune), "s"(k_tiles_x), "s"(k_tiles_y),
                 "s"(k_width), "s"(k_height), "s"(k_pitch), "s"(k_resolve_dma), "s"(k_has_sky), "s"(k_albedo), "s"(k_accum), "s"(k_stamps),
                 "s"(k_y_offset), "s"(k_band_shift), "s"(k_band_extra), "s"(k_out_band_extra));
    const bool do_shadow = k_has_shadow && k_n_dir > 0;
    const bool stage = STAGE && do_shadow && k_shadow_lds;
    // whole frame with a resolve: the tile's accum / revealage texels ride the LDS-DMA with the G-buffer planes, so their
    // latency is not exposed at the store (dword granularity: the host sets resolve_dma when pitches / width allow it)
    const bool rdma = FRAME && k_accum && k_resolve_dma;
    const TiledLds L = tiled_lds(k_n_point, stage, rdma);
    unsigned char *s_wave = s_raw + wv * L.per_wave;
    uint32_t *s_tex = (uint32_t *)s_wave;               // [plane: normal, albedo, rm, depth][j][lane]
    uint16_t *s_list = (uint16_t *)(s_wave + L.list);
    float *s_res = (float *)(s_wave + L.stash);
    float *s_shadow = (float *)(s_wave + L.shadow);
    const uint32_t *s_resolve = rdma ? (const uint32_t *)(s_wave + L.resolve) : nullptr;

    // Workgroup -> tile: consecutive workgroups (dealt round-robin over the XCDs by the dispatcher) walk along one
    // tile row (a workgroup reads whole 256-byte row segments).  The ROWS are visited in the order the host
    // computed (mil_api.cpp order_rows): rows that many point lights reach first, cheap rows last, so the
    // expensive waves are not the tail of the launch.  Nothing depends on the order but speed.
    const uint32_t bpr = (k_tiles_x + wpb - 1) / wpb, blk = blockIdx.x;
    const uint32_t ri = blk / bpr, tyi = (k_tune & 1u) || k_tiles_y > MIL_MAX_ROW_ORDER ? ri : (uint32_t)p.row_order[ri];
    const uint32_t txi = (blk - ri * bpr) * wpb + wv;
    if (txi >= k_tiles_x) return;                       // whole wave; nothing below synchronises across waves
    // frame row / output row of this tile's local rows (band dealing: a tile never straddles two bands).  Used by the unlit
    // tile's early store path only; the later phases re-derive it from the kernarg block (two SGPRs held across the whole
    // kernel were two more spilled into VGPR lanes)
    const uint32_t k_band = (tyi * TILE) >> k_band_shift;
    const RowMap rm_e = { k_y_offset + k_band * k_band_extra, k_band * k_out_band_extra };
    (void)rm_e;                                         // strict mode has no early store path
    const uint64_t t_start = k_stamps ? __builtin_amdgcn_s_memrealtime() : 0;   // diagnostics only (100 MHz)

    // ---- 1. HBM -> LDS, coordinates clamped (not branched) so the loads are straight-line ----
    const uint32_t x = txi * TILE + (lane & 15u), y0 = tyi * TILE + (lane >> 4);
    const uint32_t xc = min(x, k_width - 1), ylast = k_height - 1;
    const bool xin = x < k_width;
    {
        uint32_t i[4];       // texel offsets fit 30 bits (checked on the host): scalar plane base + one 32-bit lane offset per row
#pragma unroll
        for (int j = 0; j < 4; j++) { i[j] = min(y0 + 4 * j, ylast) * k_pitch + xc; __builtin_assume(i[j] < (1u << 30)); }
#pragma unroll
        for (int j = 0; j < 4; j++) __builtin_amdgcn_global_load_lds((gptr_t)(k_albedo + i[j]), (lptr_t)(s_tex + (4 + j) * 64), 4, 0, 0);
        if (rdma) {
            uint32_t *dst = (uint32_t *)(s_wave + L.resolve);
#pragma unroll
            for (int k = 0; k < 8; k++) {       // accum: 16 rows x 32 dwords, two rows per step
                const uint32_t row = 2 * k + (lane >> 5), d = lane & 31u;
                const size_t t = (size_t)min(tyi * TILE + row, ylast) * p.accum_pitch + min(txi * TILE + (d >> 1), p.width - 1);
                __builtin_amdgcn_global_load_lds((gptr_t)((const uint32_t *)p.accum + 2 * t + (d & 1u)), (lptr_t)(dst + k * 64), 4, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 2; k++) {       // revealage: 16 rows x 8 dwords (two texels each; width and pitch are even here)
                const uint32_t row = 8 * k + (lane >> 3), d = lane & 7u;
                const size_t t = (size_t)min(tyi * TILE + row, ylast) * p.reveal_pitch + min(txi * TILE + 2 * d, p.width - 2);
                __builtin_amdgcn_global_load_lds((gptr_t)(p.reveal + t), (lptr_t)(dst + 512 + k * 64), 4, 0, 0);
            }
        }
#if !MIL_STRICT
        // The albedo/occlusion plane goes first: a tile without a lit pixel (sky: half of Scene A) writes its zeros and
        // never fetches the other three planes (12 of its 16 B/px); lit tiles pay one more LDS-DMA round trip, which
        // the VALU-bound phase hides (-5 % on the headline frame; MIL_TUNE bit 32 restores the single round).
        if (!(k_tune & 32u)) {
            const bool early_depth = FRAME && k_has_sky;     // a sky texel is told by its depth
            // what an unlit tile's store path reads, fetched while the LDS-DMA is in flight
            void *const e_out = p.out;
            const uint32_t e_out_pitch = p.out_pitch;
            uint32_t *const e_counts = p.tile_counts;
            const uint8_t *const e_lut11 = (FRAME && FMT == FMT_SRGB8_VIA_LDR) ? p.lut11 : nullptr, *const e_lut10 = (FRAME && FMT == FMT_SRGB8_VIA_LDR) ? p.lut10 : nullptr;
            uint32_t *const e_out2 = FMT == FMT_HDR ? p.out2 : nullptr;
            const uint32_t e_out2_pitch = FMT == FMT_HDR ? p.out2_pitch : 0u;
            asm volatile("" :: "s"(e_out), "s"(e_out_pitch), "s"(e_counts), "s"(e_lut11), "s"(e_lut10), "s"(e_out2), "s"(e_out2_pitch));
            if (early_depth) {
#pragma unroll
                for (int j = 0; j < 4; j++) __builtin_amdgcn_global_load_lds((gptr_t)(p.depth + i[j]), (lptr_t)(s_tex + (12 + j) * 64), 4, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool any = false;
#pragma unroll
            for (int j = 0; j < 4; j++) any |= xin && (y0 + 4 * j) <= ylast && (s_tex[(4 + j) * 64 + lane] >> 24) != 0;
            if (__builtin_amdgcn_ballot_w64(any) == 0) {
                if (FRAME) {        // sky / resolve of the four pixels first (their uniforms are read once), then the stores
                    v4f h[4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (xin && (y0 + 4 * j) <= ylast)
                            h[j] = frame_value(p, x, y0 + 4 * j, rm_e, early_depth ? __uint_as_float(s_tex[(12 + j) * 64 + lane]) : 0.0f, f3{ 0.0f, 0.0f, 0.0f }, s_resolve);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (xin && (y0 + 4 * j) <= ylast) store_pixel_to<FMT>(e_out, e_out_pitch, e_lut11, e_lut10, x, y0 + 4 * j + rm_e.o, h[j], e_out2, e_out2_pitch);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (xin && (y0 + 4 * j) <= ylast) store_zero<FMT>(e_out, e_out_pitch, x, y0 + 4 * j + rm_e.o, e_out2, e_out2_pitch);
                }
                if (e_counts && lane == 0) e_counts[tyi * k_tiles_x + txi] = 0;
                if (k_stamps && lane == 0) {
                    uint32_t *st = k_stamps + 4 * (size_t)(tyi * k_tiles_x + txi);
                    st[0] = (uint32_t)t_start; st[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); st[2] = 0; st[3] = 0;
                }
                return;
            }
        }
        if (!(FRAME && k_has_sky) || (k_tune & 32u))
#endif
        {
#pragma unroll
            for (int j = 0; j < 4; j++) __builtin_amdgcn_global_load_lds((gptr_t)(p.depth + i[j]), (lptr_t)(s_tex + (12 + j) * 64), 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) __builtin_amdgcn_global_load_lds((gptr_t)(p.normal + i[j]), (lptr_t)(s_tex + (0 + j) * 64), 4, 0, 0);
#pragma unroll
        for (int j = 0; j < 4; j++) __builtin_amdgcn_global_load_lds((gptr_t)(p.rm + i[j]), (lptr_t)(s_tex + (8 + j) * 64), 4, 0, 0);
    }
    const v4f *cull = (const v4f *)p.point_cull;
    // The first cull records are requested with the texels - as a cache warm-up only: the culling loop loads them again (an
    // L1 hit by then).  Kept in registers for the loop instead, they stayed live across the rounds loop and the compiler
    // parked them in scratch at its entry: 1 KiB written per lit tile's wave, 16.6 MB per headline launch, which is what
    // WRITE_SIZE showed as a 1.10 x write amplification (147.1 MB against 132.7 MB of hdr) in rounds 1 and 2.
    {
        v4f warm = { 0.0f, 0.0f, 0.0f, 0.0f };
        if (lane < p.n_point) warm = cull[lane];
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(warm) : "memory");   // the LDS-DMA writes are ordered only by this wave's vmcnt
    }

    // ---- 2. AABB of lit pixels + shadow texel rectangle ----
    const float INF = __builtin_inff();
    f3 lo = { INF, INF, INF }, hi = { -INF, -INF, -INF };
    int x0 = 0x7FFFFFFF, x1 = -0x7FFFFFFF, yy0 = 0x7FFFFFFF, yy1 = -0x7FFFFFFF;
    bool lit4[4];
    bool have_pos = false;        // wave-uniform: P - cameraPosition of the tile's pixels is parked in LDS (fast mode)
    (void)have_pos;
#pragma unroll
    for (int j = 0; j < 4; j++)   // occlusion 0 => every point-light term is exactly 0
        lit4[j] = xin && (y0 + 4 * j) <= ylast && (s_tex[(4 + j) * 64 + lane] >> 24) != 0;
    // a tile without a lit pixel (sky) has an empty box: no positions, no reductions, no culling
    if ((p.n_point > 0 || stage) && __builtin_amdgcn_ballot_w64(lit4[0] || lit4[1] || lit4[2] || lit4[3]) != 0) {
        const RowMap rm = row_map(p, tyi * TILE);
#if !MIL_STRICT
        f3 P4[4];    // the same packed evaluation as the shading passes (rows j, j + 1 together): identical positions
#pragma unroll
        for (int jj = 0; jj < 4; jj += 2) {
            const PosT<v2f> q2 = pixel_position_t<v2f>(p, (v2f)((float)x), v2f{ (float)(y0 + 4 * jj + rm.g), (float)(y0 + 4 * jj + 4 + rm.g) },
                                                       v2f{ __uint_as_float(s_tex[(12 + jj) * 64 + lane]), __uint_as_float(s_tex[(13 + jj) * 64 + lane]) });
            P4[jj] = f3{ q2.P.x.x, q2.P.y.x, q2.P.z.x }; P4[jj + 1] = f3{ q2.P.x.y, q2.P.y.y, q2.P.z.y };
            if (L.per_wave != L.pos) {
                v2f *park = (v2f *)(s_wave + L.pos) + (jj / 2) * 192;
                park[lane] = q2.D.x; park[64 + lane] = q2.D.y; park[128 + lane] = q2.D.z;
            }
        }
        have_pos = L.per_wave != L.pos;
#endif
#pragma unroll
        for (int j = 0; j < 4; j++) {
#if MIL_STRICT
            const f3 P = pixel_position(p, x, y0 + 4 * j + rm.g, __uint_as_float(s_tex[(12 + j) * 64 + lane])).P;
#else
            const f3 P = P4[j];
#endif
            const bool lit = lit4[j];
            if (lit) {
                lo.x = fminf(lo.x, P.x); lo.y = fminf(lo.y, P.y); lo.z = fminf(lo.z, P.z);
                hi.x = fmaxf(hi.x, P.x); hi.y = fmaxf(hi.y, P.y); hi.z = fmaxf(hi.z, P.z);
                if (stage) {
                    const ShadowTap t = shadow_tap(P, p);
                    if (t.inside) { x0 = min(x0, t.tx); x1 = max(x1, t.tx); yy0 = min(yy0, t.ty); yy1 = max(yy1, t.ty); }
                }
            }
        }
    }
    bool any_lit = false;
    if (p.n_point > 0 && __builtin_amdgcn_ballot_w64(lo.x <= hi.x) != 0) {
        wave_aabb(lo, hi);
        any_lit = lo.x <= hi.x;
    }
    // the reduced box is wave-uniform: held in SGPRs for the culling trips (as VGPRs it was six registers live across the
    // rounds loop, one of them spilled under the five-wave budget)
    const f3 blo = { uniform_f(lo.x), uniform_f(lo.y), uniform_f(lo.z) }, bhi = { uniform_f(hi.x), uniform_f(hi.y), uniform_f(hi.z) };
    bool staged = false;
    int rx0 = 0, ry0 = 0, rw = 0;
    if (stage) {
        x0 = wave_mini(x0); x1 = wave_maxi(x1); yy0 = wave_mini(yy0); yy1 = wave_maxi(yy1);
        if (x0 <= x1) {
            rw = x1 - x0 + 3;
            const int rh = yy1 - yy0 + 3;
            rx0 = x0 - 1; ry0 = yy0 - 1;
            if ((long long)rw * rh <= SH_CAP) {
                staged = true;
                const int S = (int)p.shadow_size;
                for (int yy = (int)(lane >> 4); yy < rh; yy += 4) {
                    const int sy = min(max(ry0 + yy, 0), S - 1);      // clamp-to-edge baked into the staged copy
                    for (int xx = (int)(lane & 15u); xx < rw; xx += 16) {
                        const int sx = min(max(rx0 + xx, 0), S - 1);
                        s_shadow[yy * rw + xx] = p.shadow_map[(size_t)sy * S + sx];
                    }
                }
                __builtin_amdgcn_wave_barrier();   // LDS ops of one wave execute in order; keep the compiler from reordering
            }
        }
    }

    // Rounds: lights are culled in ascending order until they are exhausted or the LDS list (CHUNK entries) could
    // overflow in the next 64-light step; the survivors are shaded; a tile that needed another round parks its partial
    // sums in LDS and goes on culling where it stopped.  Nearly every tile takes one round whatever the light count.
    uint32_t total = 0, next = 0;
    bool first = true, last;
    const uint32_t list_cap = p.n_point < (uint32_t)CHUNK ? ((p.n_point + 63u) & ~63u) : (uint32_t)CHUNK;
#pragma unroll 1
    do {
        // ---- 3. cull lights into the wave's LDS list (ascending order) ----
        const uint32_t end = p.n_point;
        uint32_t cnt = 0;
        if (any_lit) {
            // 256 lights per trip: the four 16-byte records of a lane are requested together (clamped index instead of a
            // branch), so one memory latency is paid per 256 lights, not per 64.  A trip starts only if its survivors
            // are sure to fit the list, or if everything that is left fits.
            uint32_t b0 = next;
#pragma unroll 1
            for (; b0 < end && (cnt + 256 <= list_cap || end - b0 <= list_cap - cnt); b0 += 256) {
                v4f c4[4];
#pragma unroll
                for (int u = 0; u < 4; u++) c4[u] = cull[min(b0 + 64 * u + lane, end - 1)];
#pragma unroll
                for (int u = 0; u < 4; u++) {          // straight-line: rejected lanes write the dump entry behind the list
                    const uint32_t li = b0 + 64 * u + lane;
                    const bool pass = (li < end) & (box_d2(c4[u], blo, bhi) < c4[u].w * c4[u].w);
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
                    const uint32_t pos = cnt + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    s_list[pass ? pos : list_cap] = (uint16_t)li;
                    cnt += (uint32_t)__builtin_popcountll(m);
                }
            }
            next = min(b0, end);
            __builtin_amdgcn_wave_barrier();
        } else {
            next = end;
        }
        last = next >= end;
        total += cnt;

        // ---- 5. strict mode: lane k <- record of list entry kb + k, broadcast by v_readlane in the passes (fast mode: srec_load) ----
#if MIL_STRICT
        auto gather = [&](uint32_t kb, v4f &ga, v4f &gb) {
            const uint32_t li = s_list[min(kb + lane, cnt - 1)];
            ga = point_rec_a_v(p, li); gb = point_rec_b_v(p, li);
        };
        v4f ra = { 0.0f, 0.0f, 0.0f, 0.0f }, rb = ra;
        if (cnt) gather(0, ra, rb);       // stays in registers across the passes unless the list is longer than 64
#endif

        // E3 shadow term of one pixel (integer logic shared by both numeric modes)
        auto shadow_px = [&](const ShadowTap &st) -> float {
            if (!staged) return pcf_direct(st, p);
            if (!st.inside) return 1.0f;
            int c9 = 0;
            const int bx = st.tx - rx0, by = st.ty - ry0;
#pragma unroll
            for (int dy = -1; dy <= 1; dy++)
#pragma unroll
                for (int dx = -1; dx <= 1; dx++) c9 += (st.zc <= s_shadow[(by + dy) * rw + bx + dx]) ? 1 : 0;
            return pcf_result(c9);
        };

#if MIL_STRICT
        // ---- 6. four passes of 64 pixels ----
#pragma unroll 1
        for (uint32_t j = 0; j < 4; j++) {
            const ShadeParams &p = kernarg_params();
            const RowMap rm = row_map(p, tyi * TILE);
            const uint32_t y = y0 + 4 * j;
            const uint32_t u1 = s_tex[(0 + j) * 64 + lane], w2 = s_tex[(4 + j) * 64 + lane], u3 = s_tex[(8 + j) * 64 + lane];
            const float depth = __uint_as_float(s_tex[(12 + j) * 64 + lane]);
            const bool inbj = xin && y <= ylast, litj = inbj && (w2 >> 24) != 0;
            Px q;
            setup_pixel(q, p, pixel_position(p, x, y + rm.g, depth), u1, w2, u3);
            f3 r = { 0.0f, 0.0f, 0.0f };
            if (first) {
                float shadow = 1.0f;
                if (do_shadow && litj) shadow = shadow_px(shadow_tap(q.P, p));
                for (uint32_t l = 0; l < p.n_dir; l++) add_directional(q, dir_rec(p, l), shadow, do_shadow && l == 0, r);   // fs_deferred_lighting.glsl:62-64
            } else {
                r = f3{ s_res[j * 64 + lane], s_res[256 + j * 64 + lane], s_res[512 + j * 64 + lane] };
            }
            if (cnt) {
                uint32_t kb = 0;
#pragma unroll 1
                for (;;) {
                    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, cnt - kb));      // scalar loop bound
#pragma unroll 1
                    for (uint32_t k = 0; k < n; k++) {
                        const v4f a = bcast4(ra, (int)k);
                        f3 Lv;
                        float d2;
                        const bool on = point_test(q, a, Lv, d2);
                        if (__builtin_amdgcn_ballot_w64(on) == 0) continue;      // wave-uniform: nobody in range
                        point_eval(q, a, bcast4(rb, (int)k), Lv, d2, on, r);
                    }
                    kb += 64;
                    if (kb >= cnt) break;
                    gather(kb, ra, rb);
                }
                if (cnt > 64 && j < 3) gather(0, ra, rb);
            }
            if (last) {
                if (p.has_ambient) add_ambient(q, p, w2, r);
                if (inbj) emit_pixel<FMT, FRAME>(p, x, y, rm, depth, r, s_resolve);
            } else {
                s_res[j * 64 + lane] = r.x; s_res[256 + j * 64 + lane] = r.y; s_res[512 + j * 64 + lane] = r.z;
            }
        }
#else
        // ---- 6. two passes of 128 pixels: each lane shades rows ja and ja + 1 of its column together (v2f) ----
#pragma unroll 1
        for (uint32_t jp = 0; jp < 2; jp++) {
            const ShadeParams &p = kernarg_params();
            const RowMap rm = row_map(p, tyi * TILE);
            const uint32_t ja = 2 * jp, ya = y0 + 4 * ja, yb = ya + 4;
            const v2u u1 = { s_tex[(0 + ja) * 64 + lane], s_tex[(1 + ja) * 64 + lane] };
            const v2u w2 = { s_tex[(4 + ja) * 64 + lane], s_tex[(5 + ja) * 64 + lane] };
            const v2u u3 = { s_tex[(8 + ja) * 64 + lane], s_tex[(9 + ja) * 64 + lane] };
            const v2f depth = { __uint_as_float(s_tex[(12 + ja) * 64 + lane]), __uint_as_float(s_tex[(13 + ja) * 64 + lane]) };
            const bool inA = xin && ya <= ylast, inB = xin && yb <= ylast;
            const bool litA = inA && (w2.x >> 24) != 0, litB = inB && (w2.y >> 24) != 0;
            // occlusion 0 (cleared G-buffer, render/mod.rs:129-131: sky / background) multiplies every term of
            // fs_deferred_lighting.glsl:63 and of the extensions: the pixel is exactly 0.  When none of the
            // pass's 128 pixels is lit the wave writes the zeros and skips the shading altogether.
            if (__builtin_amdgcn_ballot_w64(litA || litB) == 0) {
                if (last && inA) emit_pixel<FMT, FRAME>(p, x, ya, rm, depth.x, f3{ 0.0f, 0.0f, 0.0f }, s_resolve);
                if (last && inB) emit_pixel<FMT, FRAME>(p, x, yb, rm, depth.y, f3{ 0.0f, 0.0f, 0.0f }, s_resolve);
                continue;
            }
            Px2 q;
            PosT<v2f> pos;
            if (have_pos) {       // the box phase evaluated this (same function, same inputs): read it back
                // (the lane's address is formed here, per pass: hoisted out of the rounds loop it was held in a VGPR across
                // the whole kernel and spilled to scratch memory under the five-wave register budget)
                uint32_t ln = lane;
                asm volatile("" : "+v"(ln));
                const v2f *park = (const v2f *)(s_wave + L.pos) + jp * 192;
                pos.D = V3<v2f>{ park[ln], park[64 + ln], park[128 + ln] };
                pos.P = V3<v2f>{ pos.D.x + p.cam[0], pos.D.y + p.cam[1], pos.D.z + p.cam[2] };
            } else {
                pos = pixel_position_t<v2f>(p, (v2f)((float)x), v2f{ (float)(ya + rm.g), (float)(yb + rm.g) }, depth);
            }
            setup_pixel_t<v2f, v2u>(q, pos, u1, w2, u3);
            V3<v2f> r = { (v2f)(0.0f), (v2f)(0.0f), (v2f)(0.0f) };
            v2f shadow = { 1.0f, 1.0f };
            if (first) {
                r = ambient_start(q, p, w2);
                if (do_shadow) {
                    ShadowTap ta, tb;
                    shadow_tap2(q.P, p, ta, tb);
                    if (litA) shadow.x = shadow_px(ta);
                    if (litB) shadow.y = shadow_px(tb);
                }
            }
            if (cnt) far_if_unlit(q);                        // after the taps (they read P), before the loops (occlusion dies here)
            if (first) {
                for (uint32_t l = 0; l < p.n_dir; l++) add_directional2(q, dir_rec(p, l), shadow, do_shadow && l == 0, r);   // fs_deferred_lighting.glsl:62-64
            } else {
                r.x = v2f{ s_res[ja * 64 + lane], s_res[(ja + 1) * 64 + lane] };
                r.y = v2f{ s_res[256 + ja * 64 + lane], s_res[256 + (ja + 1) * 64 + lane] };
                r.z = v2f{ s_res[512 + ja * 64 + lane], s_res[512 + (ja + 1) * 64 + lane] };
            }
            if (cnt) {
                auto light = [&](const SRec &rec) {
                    const v4f a = { rec.f[0], rec.f[1], rec.f[2], rec.f[3] };
                    V3<v2f> Lv;
                    const v2f d2 = point_d2_2(q, a, Lv);
                    const PointMasks pm = point_masks(d2, a.w);
                    if ((pm.m0 | pm.m1) == 0) return;         // wave-uniform: nobody in range
                    const bool on0 = __builtin_amdgcn_inverse_ballot_w64(pm.m0), on1 = __builtin_amdgcn_inverse_ballot_w64(pm.m1);
                    point_eval2(q, v4f{ rec.f[4], rec.f[5], rec.f[6], rec.f[7] }, Lv, d2, on0, on1, r);
                };
                const uint64_t recs = (uint64_t)(uintptr_t)p.point_prep;
                uint32_t kb = 0;
#pragma unroll 1
                for (;;) {
                    const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, cnt - kb));      // scalar loop bound
                    const uint32_t lv = s_list[min(kb + lane, cnt - 1)];
                    SRec r0 = srec_load(recs, (uint32_t)__builtin_amdgcn_readlane((int)lv, 0)), r1;
                    srec_wait();
#pragma unroll 1
                    for (uint32_t k = 0; k < n; k += 2) {
                        r1 = srec_load(recs, (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)min(k + 1u, n - 1u)));
                        light(r0);
                        srec_wait();
                        r0 = srec_load(recs, (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)min(k + 2u, n - 1u)));
                        if (k + 1u < n) light(r1);
                        srec_wait();
                    }
                    kb += 64;
                    if (kb >= cnt) break;
                }
            }
            if (last) {
                if (!FRAME) emit_pair<FMT>(p, x, ya, yb, rm, inA, inB, r);
                else {
                    if (inA) emit_pixel<FMT, FRAME>(p, x, ya, rm, depth.x, f3{ r.x.x, r.y.x, r.z.x }, s_resolve);
                    if (inB) emit_pixel<FMT, FRAME>(p, x, yb, rm, depth.y, f3{ r.x.y, r.y.y, r.z.y }, s_resolve);
                }
            } else {
                s_res[ja * 64 + lane] = r.x.x; s_res[(ja + 1) * 64 + lane] = r.x.y;
                s_res[256 + ja * 64 + lane] = r.y.x; s_res[256 + (ja + 1) * 64 + lane] = r.y.y;
                s_res[512 + ja * 64 + lane] = r.z.x; s_res[512 + (ja + 1) * 64 + lane] = r.z.y;
            }
        }
#endif
        __builtin_amdgcn_wave_barrier();   // the list is rewritten by the next round only after these reads were issued
        first = false;
    } while (!last);
    if (p.tile_counts && lane == 0) p.tile_counts[tyi * p.tiles_x + txi] = total;
    if (p.tile_stamps && lane == 0) {                    // diagnostics: wave lifetime and placement
        uint32_t *st = p.tile_stamps + 4 * (size_t)(tyi * p.tiles_x + txi);
        st[0] = (uint32_t)t_start; st[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); st[2] = total;
        st[3] = (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (15 << 11)) & 0xFFFFu) |          // HW_REG_HW_ID[15:0]
                (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 16);             // HW_REG_XCC_ID[3:0]
    }
}

#if !MIL_STRICT
// ---------------------------------------------------------------------------------
// K2, two-kernel form (fast mode, mil_light_ex): k_bin -> k_shade.
//
// k_tiled gives every 16x16 tile one wave that classifies, culls AND shades.  Measured on it (tools/stamps.py): half of the
// waves of Scene A are sky tiles that live 5 us to store zeros while holding a 128-VGPR slot; every wave of a 4096-light
// frame tests all 4096 lights against its own box (11 % of the instructions); a light is evaluated for the 128 pixels of a
// pass when it reaches one (72 % lane efficiency); and the waves of the heaviest tiles (up to 380 us of a 600 us launch)
// start whenever the row order gets to them, so the launch ends on a tail at a quarter of the wave slots.
//
//   k_bin    one workgroup per 64x64 MACRO tile, one wave per 16x16 tile, light on registers.  Reads the occlusion bytes
//            (and the depth of lit tiles only); writes the exact zeros of unlit half tiles; boxes of the lit pixels of
//            each 16x8 HALF tile (the same packed position evaluation the shading uses); culling in two levels - the
//            16 waves test the lights against the macro tile's box together (1024 lights per step, survivors compacted
//            into LDS in ascending order by ballot + mbcnt + a prefix over the waves' counts), then each wave tests the
//            survivors against its two half-tile boxes and appends to their lists in the context's workspace, ascending;
//            every lit half tile is appended to a COST BUCKET (by list length).
//   k_shade  one wave per listed half tile, two pixels per lane (packed FP32), most expensive bucket first: texels
//            straight into registers, directional lights + PCF, then the list, 64 records per block broadcast with
//            v_readlane as in k_tiled.  No classification, no culling, no LDS.
// The sums run in ascending light order as in k_tiled and in the brute-force loop: all three agree bit for bit.
// ---------------------------------------------------------------------------------
constexpr int BIN_TX = 4, BIN_TY = 4, BIN_WAVES = BIN_TX * BIN_TY;
#ifndef MIL_SHADE_WAVES
#define MIL_SHADE_WAVES 4      // waves per SIMD k_shade's registers are allocated for
#endif
constexpr uint32_t COARSE_MIN = 512, COARSE_CAP = 8192;

DEV uint32_t cost_bucket(uint32_t len, bool brute)      // bucket 0 = most expensive; monotone in the list length
{
    if (brute) return 0u;
    const uint32_t c = len < 32u ? (len >> 1) : 16u + min(15u, (len - 32u) >> 5);      // 16 steps of 2, then steps of 32 up to 512+
    return (MIL_WS_BUCKETS - 1u) - c;
}
DEV void store_zero_rt(int fmt, void *out, uint32_t out_pitch, uint32_t x, uint32_t y, uint32_t *out2, uint32_t out2_pitch)
{
    const size_t o = (size_t)y * out_pitch + x;
    if (fmt == FMT_HDR) {
        __builtin_nontemporal_store(v4f{ 0.0f, 0.0f, 0.0f, 1.0f }, (v4f *)out + o);
        if (out2) __builtin_nontemporal_store(0xFF000000u, out2 + (size_t)y * out2_pitch + x);
    }
    else __builtin_nontemporal_store(fmt == FMT_LDR ? 0u : 0xFF000000u, (uint32_t *)out + o);
}
DEV uint32_t wave_incl_prefix(uint32_t v, uint32_t lane)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, o, 64);
        if (lane >= (uint32_t)o) v += t;
    }
    return v;
}

// Workgroup barrier that orders LDS traffic only: __syncthreads() carries a fence that also waits for every outstanding
// global access of the wave (vmcnt(0)) - here that would put the round trip of the rank atomics in front of the barrier
// instead of beside the pool allocation's.
DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(1024) void k_bin(const ShadeParams p, const int fmt)
{
    __shared__ uint16_t s_coarse[COARSE_CAP];
    __shared__ uint16_t s_list[BIN_WAVES][2][MIL_WS_LIST_MAX];     // the two half-tile lists of every wave, built here, copied out compact
    __shared__ float s_box[BIN_WAVES][6];
    __shared__ uint32_t s_cnt[64];
    __shared__ uint32_t s_off[2 * BIN_WAVES];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t macros_x = (p.tiles_x + BIN_TX - 1) / BIN_TX;
    const uint32_t mby = blockIdx.x / macros_x, mbx = blockIdx.x - mby * macros_x;
    const uint32_t txi = mbx * BIN_TX + (wv & 3u), tyi = mby * BIN_TY + (wv >> 2);
    const bool tile_in = txi < p.tiles_x && tyi < p.tiles_y;       // no early return: the workgroup meets at barriers below
    const uint32_t x = txi * TILE + (lane & 15u), y0 = tyi * TILE + (lane >> 4);
    const uint32_t xc = min(x, p.width - 1), ylast = p.height - 1;
    const bool xin = tile_in && x < p.width;
    const RowMap rm = row_map(p, tyi * TILE);

    // ---- classify: occlusion bytes of the tile's 256 texels
    uint32_t off[4];
    bool lit4[4];
#pragma unroll
    for (int j = 0; j < 4; j++) { off[j] = min(y0 + 4 * j, ylast) * p.pitch + xc; __builtin_assume(off[j] < (1u << 30)); }
    // Everything the wave will read is requested up front - occlusion bytes, depths (wasted on a sky tile: 4 of the 8 bytes
    // a texel costs here) and the first 256 cull records - so that the wave pays ONE memory latency before its arithmetic
    // instead of three in a row; the kernel is bound by that latency, not by bandwidth or issue.
    const uint32_t n_point = p.n_point;
    const v4f *cull = (const v4f *)p.point_cull;
    const bool coarse = n_point > COARSE_MIN && n_point <= COARSE_CAP;      // uniform over the launch
    uint32_t u2[4] = { 0u, 0u, 0u, 0u };
    float dz[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    v4f pc[4];
    if (tile_in) {
#pragma unroll
        for (int j = 0; j < 4; j++) u2[j] = p.albedo[off[j]];
#pragma unroll
        for (int j = 0; j < 4; j++) dz[j] = p.depth[off[j]];
    }
#pragma unroll
    for (int u = 0; u < 4; u++)         // the records the first culling step tests (coarse: this wave's share of the first 4096)
        pc[u] = cull[min(coarse ? (uint32_t)u * (BIN_WAVES * 64u) + wv * 64u + lane : 64u * u + lane, n_point - 1)];
#pragma unroll
    for (int j = 0; j < 4; j++) lit4[j] = xin && (y0 + 4 * j) <= ylast && (u2[j] >> 24) != 0;
    const bool anyA = __builtin_amdgcn_ballot_w64(lit4[0] || lit4[1]) != 0, anyB = __builtin_amdgcn_ballot_w64(lit4[2] || lit4[3]) != 0;
    // occlusion 0 multiplies every term (fs_deferred_lighting.glsl:63 and the extensions): an unlit half tile is exactly 0.
    // A whole-frame launch with a sky or a resolve subpass (mil_shade_frame) has more to do for such pixels than a
    // constant: there every half tile inside the frame becomes an item (empty list, cheapest bucket) and k_shade's store
    // path (emit_pixel<FMT, true>) does the rest.
    const bool frame_items = p.has_sky || p.accum != nullptr;
    const bool itemA = tile_in && (anyA || frame_items), itemB = tile_in && (anyB || (frame_items && tyi * TILE + 8u <= ylast));
    if (!frame_items) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (xin && (y0 + 4 * j) <= ylast && !(j < 2 ? anyA : anyB)) store_zero_rt(fmt, p.out, p.out_pitch, x, y0 + 4 * j + rm.o, p.out2, p.out2_pitch);
    }

    // ---- boxes of the lit pixels of the two half tiles
    const float INF = __builtin_inff();
    f3 loA = { INF, INF, INF }, hiA = { -INF, -INF, -INF }, loB = loA, hiB = hiA;
    if (anyA || anyB) {
#pragma unroll
        for (int jj = 0; jj < 4; jj += 2) {
            const PosT<v2f> q2 = pixel_position_t<v2f>(p, (v2f)((float)x), v2f{ (float)(y0 + 4 * jj + rm.g), (float)(y0 + 4 * jj + 4 + rm.g) },
                                                       v2f{ dz[jj], dz[jj + 1] });
            f3 &lo = jj ? loB : loA, &hi = jj ? hiB : hiA;
            if (lit4[jj]) {
                lo.x = fminf(lo.x, q2.P.x.x); lo.y = fminf(lo.y, q2.P.y.x); lo.z = fminf(lo.z, q2.P.z.x);
                hi.x = fmaxf(hi.x, q2.P.x.x); hi.y = fmaxf(hi.y, q2.P.y.x); hi.z = fmaxf(hi.z, q2.P.z.x);
            }
            if (lit4[jj + 1]) {
                lo.x = fminf(lo.x, q2.P.x.y); lo.y = fminf(lo.y, q2.P.y.y); lo.z = fminf(lo.z, q2.P.z.y);
                hi.x = fmaxf(hi.x, q2.P.x.y); hi.y = fmaxf(hi.y, q2.P.y.y); hi.z = fmaxf(hi.z, q2.P.z.y);
            }
        }
        if (anyA) wave_aabb(loA, hiA);
        if (anyB) wave_aabb(loB, hiB);
    }

    // ---- level 1: the macro tile's box against all lights, by the 16 waves together
    uint32_t coarse_n = 0;
    if (coarse) {
        if (lane == 0) {
            s_box[wv][0] = fminf(loA.x, loB.x); s_box[wv][1] = fminf(loA.y, loB.y); s_box[wv][2] = fminf(loA.z, loB.z);
            s_box[wv][3] = fmaxf(hiA.x, hiB.x); s_box[wv][4] = fmaxf(hiA.y, hiB.y); s_box[wv][5] = fmaxf(hiA.z, hiB.z);
        }
        __syncthreads();
        f3 lo = { INF, INF, INF }, hi = { -INF, -INF, -INF };
        if (lane < (uint32_t)BIN_WAVES) {
            lo = f3{ s_box[lane][0], s_box[lane][1], s_box[lane][2] };
            hi = f3{ s_box[lane][3], s_box[lane][4], s_box[lane][5] };
        }
        wave_aabb(lo, hi);
        if (lo.x <= hi.x) {                                               // the same for every wave of the workgroup
#pragma unroll 1
            for (uint32_t base = 0; base < n_point; base += 4u * BIN_WAVES * 64u) {
                unsigned long long m4[4];
                uint32_t li4[4];
                v4f c4[4];
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    li4[t] = base + (uint32_t)t * (BIN_WAVES * 64u) + wv * 64u + lane;
                    c4[t] = base == 0 ? pc[t] : cull[min(li4[t], n_point - 1)];
                }
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const bool pass = li4[t] < n_point && box_d2(c4[t], lo, hi) < c4[t].w * c4[t].w;
                    m4[t] = __builtin_amdgcn_ballot_w64(pass);
                    if (lane == 0) s_cnt[t * BIN_WAVES + wv] = (uint32_t)__builtin_popcountll(m4[t]);
                }
                __syncthreads();
                const uint32_t cnt = s_cnt[lane];
                const uint32_t incl = wave_incl_prefix(cnt, lane), excl = incl - cnt;     // (step, wave) order = ascending light index
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const uint32_t o = coarse_n + (uint32_t)__builtin_amdgcn_readlane((int)excl, t * BIN_WAVES + (int)wv) +
                                       __builtin_amdgcn_mbcnt_hi((uint32_t)(m4[t] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m4[t], 0u));
                    if ((m4[t] >> lane) & 1ull) s_coarse[o] = (uint16_t)li4[t];
                }
                coarse_n += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                __syncthreads();
            }
        }
    }

    // ---- level 2: this wave's two half-tile boxes against the survivors (or against all lights), lists in ascending order, in LDS
    const uint32_t src_n = coarse ? coarse_n : n_point, cap = p.ws_list_cap;
    const uint32_t idA = 2u * (tyi * p.tiles_x + txi);
    uint32_t lenA = 0, lenB = 0, cntU = 0;
    uint16_t *const listA = s_list[wv][0], *const listB = s_list[wv][1];
    if (anyA || anyB) {
#pragma unroll 1
        for (uint32_t b0 = 0; b0 < src_n; b0 += 256u) {
            uint32_t li4[4];
            v4f c4[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {                 // four records per lane requested together: one latency per 256 candidates
                const uint32_t k = min(b0 + 64u * u + lane, src_n - 1);
                li4[u] = coarse ? (uint32_t)s_coarse[k] : k;
                c4[u] = (!coarse && b0 == 0) ? pc[u] : cull[li4[u]];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool valid = b0 + 64u * u + lane < src_n;
                const float r2 = c4[u].w * c4[u].w;
                const bool pA = valid && anyA && box_d2(c4[u], loA, hiA) < r2, pB = valid && anyB && box_d2(c4[u], loB, hiB) < r2;
                const unsigned long long mA = __builtin_amdgcn_ballot_w64(pA), mB = __builtin_amdgcn_ballot_w64(pB);
                const uint32_t posA = lenA + __builtin_amdgcn_mbcnt_hi((uint32_t)(mA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mA, 0u));
                const uint32_t posB = lenB + __builtin_amdgcn_mbcnt_hi((uint32_t)(mB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mB, 0u));
                if (pA && posA < cap) listA[posA] = (uint16_t)li4[u];
                if (pB && posB < cap) listB[posB] = (uint16_t)li4[u];
                lenA += (uint32_t)__builtin_popcountll(mA); lenB += (uint32_t)__builtin_popcountll(mB);
                cntU += (uint32_t)__builtin_popcountll(mA | mB);
            }
        }
    }
    // ---- the macro tile's 32 lists go to the pool back to back: ONE returning atomic per workgroup.  The kernel is bound by
    // latency, and a returning atomic on a word that all XCDs share takes 4-8 us: (a) the pool is cut into MIL_WS_POOL_PARTS
    // partitions with a counter word (its own 128-byte line) each and a workgroup asks the partition of the XCD it runs on
    // first (1.5-2.7 us; a full partition sends it on to the next one, so nothing depends on which XCD that is); (b) the
    // returning adds that give the two items their rank in the launch order (k_order) are issued BEFORE the barrier, so
    // their round trip runs under the allocation's.
    const bool bruteA = lenA > cap, bruteB = lenB > cap;          // more lights reach the half tile than a list holds: brute force there
    const uint32_t eA = (itemA && !bruteA) ? lenA : 0u, eB = (itemB && !bruteB) ? lenB : 0u;
    const uint32_t tile = tyi * p.tiles_x + txi, shard = tile % MIL_WS_SHARDS;
    const uint32_t bktA = cost_bucket(lenA, bruteA), bktB = cost_bucket(lenB, bruteB);
    uint32_t rankA = 0, rankB = 0;
    if (lane == 0) {
        s_off[2 * wv] = eA; s_off[2 * wv + 1] = eB;
        // [shard][bucket]: a shard's 32 words are one 128-byte line, so the hot buckets' words of different shards lie in
        // different lines (laid out [bucket][shard] the 64 words of the one bucket most items fall into shared two lines and
        // the adds serialised there: k_bin 108 us instead of 58)
        if (itemA) rankA = atomicAdd(&p.ws_hist[shard * MIL_WS_BUCKETS + bktA], 1u);
        if (itemB) rankB = atomicAdd(&p.ws_hist[shard * MIL_WS_BUCKETS + bktB], 1u);
    }
    lds_barrier();
    if (wv == 0) {
        const uint32_t v = lane < 2u * BIN_WAVES ? s_off[lane] : 0u;
        const uint32_t incl = wave_incl_prefix(v, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t base = MIL_WS_NONE;
        if (total) {
            const uint32_t part_cap = p.ws_pool_cap / MIL_WS_POOL_PARTS;
            const uint32_t xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));           // HW_REG_XCC_ID[3:0]
#pragma unroll 1
            for (uint32_t k = 0; k < MIL_WS_POOL_PARTS && base == MIL_WS_NONE; k++) {
                const uint32_t part = (xcc + k) % MIL_WS_POOL_PARTS;
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&p.ws_pool_used[part * 32u], total);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if ((uint64_t)b + total <= (uint64_t)part_cap) base = part * part_cap + b;
            }
            // no partition has room (mil_set_list_pool): the macro tile's half tiles are shaded by brute force
            if (base == MIL_WS_NONE && lane == 0) atomicAdd(p.ws_pool_used + MIL_WS_POOL_PARTS * 32u, 1u);
        }
        if (lane < 2u * BIN_WAVES) s_off[lane] = base != MIL_WS_NONE ? base + (incl - v) : MIL_WS_NONE;
    }
    lds_barrier();
    if (tile_in) {
        const uint32_t offA = s_off[2 * wv], offB = s_off[2 * wv + 1];
        if (offA != MIL_WS_NONE) {
            for (uint32_t i = lane; i < eA; i += 64u) p.ws_pool[offA + i] = listA[i];
            for (uint32_t i = lane; i < eB; i += 64u) p.ws_pool[offB + i] = listB[i];
        }
        // ---- meta of both half tiles: list length, cost bucket, brute-force flag; offset of the list
        if (lane == 0) {
            const bool bA = bruteA || (offA == MIL_WS_NONE && eA), bB = bruteB || (offB == MIL_WS_NONE && eB);
            p.ws_meta[idA] = itemA ? make_uint2((bA ? MIL_WS_BRUTE : lenA) | (bktA << 16), bA ? 0u : offA) : make_uint2(MIL_WS_NONE, 0u);
            p.ws_meta[idA + 1] = itemB ? make_uint2((bB ? MIL_WS_BRUTE : lenB) | (bktB << 16), bB ? 0u : offB) : make_uint2(MIL_WS_NONE, 0u);
            if (itemA) p.ws_rank[idA] = rankA;
            if (itemB) p.ws_rank[idA + 1] = rankB;
            if (p.tile_counts) p.tile_counts[tile] = cntU;
        }
    }
}

// k_order: a counting sort of the items by cost bucket.  Every workgroup turns the histogram (2048 words, 8 KB)
// into exclusive segment starts in LDS (bucket-major: bucket 0, the most expensive, first), then walks its share of the
// half tiles: an item's slot is its segment's start + the rank k_bin gave it.  No atomics.  ws_order = one launch order,
// longest lists first; ws_total = their number.
constexpr uint32_t ORDER_THREADS = 1024, ORDER_KEYS = MIL_WS_BUCKETS * MIL_WS_SHARDS;
static_assert(ORDER_KEYS == 2 * ORDER_THREADS, "two histogram words per thread");
__global__ __launch_bounds__(1024) void k_order(const ShadeParams p)
{
    __shared__ uint32_t s_start[ORDER_KEYS];
    __shared__ uint32_t s_wave[ORDER_THREADS / 64];
    const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    // keys in bucket-major order (key = bucket * 64 + shard); the table itself is [shard][bucket] (see k_bin)
    uint2 c;
    c.x = p.ws_hist[((2 * t) % MIL_WS_SHARDS) * MIL_WS_BUCKETS + (2 * t) / MIL_WS_SHARDS];
    c.y = p.ws_hist[((2 * t + 1) % MIL_WS_SHARDS) * MIL_WS_BUCKETS + (2 * t + 1) / MIL_WS_SHARDS];
    const uint32_t mine = c.x + c.y;
    const uint32_t incl = wave_incl_prefix(mine, lane);
    if (lane == 63u) s_wave[wv] = incl;
    __syncthreads();
    uint32_t before = 0, all = 0;
#pragma unroll
    for (uint32_t k = 0; k < ORDER_THREADS / 64; k++) { const uint32_t v = s_wave[k]; before += k < wv ? v : 0u; all += v; }
    const uint32_t excl = before + incl - mine;
    s_start[2 * t] = excl; s_start[2 * t + 1] = excl + c.x;
    if (blockIdx.x == 0 && t == 0) *p.ws_total = all;
    __syncthreads();
    const uint32_t n_items = 2u * p.tiles_x * p.tiles_y;
    for (uint32_t i = blockIdx.x * ORDER_THREADS + t; i < n_items; i += gridDim.x * ORDER_THREADS) {
        const uint32_t m = p.ws_meta[i].x;
        if (m == MIL_WS_NONE) continue;
        const uint32_t key = ((m >> 16) & (MIL_WS_BUCKETS - 1u)) * MIL_WS_SHARDS + (i >> 1) % MIL_WS_SHARDS;
        p.ws_order[s_start[key] + p.ws_rank[i]] = i;
    }
}

template <int FMT, bool FRAME>
__global__ __launch_bounds__(256, MIL_SHADE_WAVES) void k_shade(const ShadeParams p)
{
    // four independent waves per workgroup, one item each (no barrier, no LDS): single-wave workgroups are admitted to a CU
    // in smaller numbers than its wave slots
    const uint32_t lane = threadIdx.x & 63u, w = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (w >= *(const uint32_t *)p.ws_total) return;       // beyond the last listed half tile (the grid is the upper bound)
    const uint64_t t_start = p.tile_stamps ? __builtin_amdgcn_s_memrealtime() : 0;      // diagnostics only (100 MHz)
    const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.ws_order[w]);
    const uint2 meta2 = p.ws_meta[item];
    const uint32_t meta = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta2.x), list_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)meta2.y);
    const uint32_t tile = item >> 1, tyi = tile / p.tiles_x, txi = tile - tyi * p.tiles_x;
    const uint32_t x = txi * TILE + (lane & 15u), ya = tyi * TILE + (item & 1u) * 8u + (lane >> 4), yb = ya + 4;
    const uint32_t xc = min(x, p.width - 1), ylast = p.height - 1;
    const uint32_t ia = min(ya, ylast) * p.pitch + xc, ib = min(yb, ylast) * p.pitch + xc;
    __builtin_assume(ia < (1u << 30)); __builtin_assume(ib < (1u << 30));
    const v2u w2 = { __builtin_nontemporal_load(p.albedo + ia), __builtin_nontemporal_load(p.albedo + ib) };
    const v2u u1 = { __builtin_nontemporal_load(p.normal + ia), __builtin_nontemporal_load(p.normal + ib) };
    const v2u u3 = { __builtin_nontemporal_load(p.rm + ia), __builtin_nontemporal_load(p.rm + ib) };
    const v2f depth = { __builtin_nontemporal_load(p.depth + ia), __builtin_nontemporal_load(p.depth + ib) };
    const bool brute = (meta & MIL_WS_BRUTE) != 0;
    const uint32_t len = brute ? 0u : (meta & 0xFFFFu);
    const uint16_t *const list = p.ws_pool + list_off;
    uint32_t lv = len ? list[min(lane, len - 1u)] : 0u;      // lane k: index of list entry k, requested with the texels
    const RowMap rm = row_map(p, tyi * TILE);
    const bool inA = x < p.width && ya <= ylast, inB = x < p.width && yb <= ylast;
    const bool litA = inA && (w2.x >> 24) != 0, litB = inB && (w2.y >> 24) != 0;
    if (FRAME && __builtin_amdgcn_ballot_w64(litA || litB) == 0) {      // a whole-frame item without a lit pixel: sky / resolve of exact zeros
        if (inA) emit_pixel<FMT, true>(p, x, ya, rm, depth.x, f3{ 0.0f, 0.0f, 0.0f });
        if (inB) emit_pixel<FMT, true>(p, x, yb, rm, depth.y, f3{ 0.0f, 0.0f, 0.0f });
        return;
    }

    Px2 q;
    setup_pixel_t<v2f, v2u>(q, pixel_position_t<v2f>(p, (v2f)((float)x), v2f{ (float)(ya + rm.g), (float)(yb + rm.g) }, depth), u1, w2, u3);
    V3<v2f> r = ambient_start(q, p, w2);
    const bool do_shadow = p.has_shadow && p.n_dir > 0;
    v2f shadow = { 1.0f, 1.0f };
    if (do_shadow) {
        ShadowTap ta, tb;
        shadow_tap2(q.P, p, ta, tb);
        if (litA) shadow.x = pcf_direct(ta, p);
        if (litB) shadow.y = pcf_direct(tb, p);
    }
    far_if_unlit(q);                                         // after the taps (they read P), before the loops (occlusion dies here)
    for (uint32_t l = 0; l < p.n_dir; l++) add_directional2(q, dir_rec(p, l), shadow, do_shadow && l == 0, r);   // fs_deferred_lighting.glsl:62-64
    if (len) {
        // Records through the scalar cache: lane k holds the index of list entry kb + k; one v_readlane brings an index to an
        // SGPR and one s_load_dwordx8 fetches its record WHILE the light before it is evaluated (no VALU broadcasts).  The
        // compiler sinks a scalar load to its first use, so the load is issued by hand (srec_load) and waited for by hand
        // (srec_wait) - always before the record is touched and before the loop's back edge.
        auto light = [&](const SRec &rec) {
            const v4f a = { rec.f[0], rec.f[1], rec.f[2], rec.f[3] };
            V3<v2f> Lv;
            const v2f d2 = point_d2_2(q, a, Lv);
            const PointMasks pm = point_masks(d2, a.w);
            if ((pm.m0 | pm.m1) == 0) return;             // wave-uniform: nobody in range
            const bool on0 = __builtin_amdgcn_inverse_ballot_w64(pm.m0), on1 = __builtin_amdgcn_inverse_ballot_w64(pm.m1);
            point_eval2(q, v4f{ rec.f[4], rec.f[5], rec.f[6], rec.f[7] }, Lv, d2, on0, on1, r);
        };
        const uint64_t recs = (uint64_t)(uintptr_t)p.point_prep;
        uint32_t kb = 0;
#pragma unroll 1
        for (;;) {
            const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(64u, len - kb));      // scalar loop bound
            SRec r0 = srec_load(recs, (uint32_t)__builtin_amdgcn_readlane((int)lv, 0)), r1;
            srec_wait();
#pragma unroll 1
            for (uint32_t k = 0; k < n; k += 2) {
                r1 = srec_load(recs, (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)min(k + 1u, n - 1u)));
                light(r0);
                srec_wait();
                r0 = srec_load(recs, (uint32_t)__builtin_amdgcn_readlane((int)lv, (int)min(k + 2u, n - 1u)));
                if (k + 1u < n) light(r1);
                srec_wait();
            }
            kb += 64;
            if (kb >= len) break;
            lv = list[min(kb + lane, len - 1u)];
        }
    } else if (brute) {
                                   // every light, as the brute-force kernel (same order, same bits)
#pragma unroll 1
        for (uint32_t l = 0; l < p.n_point; l++) {
            const v4f a = point_rec_a(p, l);
            V3<v2f> Lv;
            const v2f d2 = point_d2_2(q, a, Lv);
            const PointMasks pm = point_masks(d2, a.w);
            if ((pm.m0 | pm.m1) == 0) continue;
            const bool on0 = __builtin_amdgcn_inverse_ballot_w64(pm.m0), on1 = __builtin_amdgcn_inverse_ballot_w64(pm.m1);
            point_eval2(q, point_rec_b(p, l), Lv, d2, on0, on1, r);
        }
    }
    if (!FRAME) emit_pair<FMT>(p, x, ya, yb, rm, inA, inB, r);
    else {
        if (inA) emit_pixel<FMT, FRAME>(p, x, ya, rm, depth.x, f3{ r.x.x, r.y.x, r.z.x });
        if (inB) emit_pixel<FMT, FRAME>(p, x, yb, rm, depth.y, f3{ r.x.y, r.y.y, r.z.y });
    }
    if (p.tile_stamps && lane == 0) {                     // diagnostics (MIL_TUNE bit 2048): per ITEM start, end, list length, launch index
        uint32_t *st = p.tile_stamps + 4 * (size_t)item;
        st[0] = (uint32_t)t_start; st[1] = (uint32_t)__builtin_amdgcn_s_memrealtime(); st[2] = brute ? 0xFFFFu : len; st[3] = w;

    }
}
#endif

// ---------------------------------------------------------------------------------
// K4: tonemap + pack (fs_tonemap.glsl:23-27).  Block = 64 x 4 pixels.
// ---------------------------------------------------------------------------------
template <int FMT, int ROWS>
__global__ __launch_bounds__(256) void k_tonemap(const TonemapParams p)
{
    // ROWS rows per lane (y, y + 4, ...): all loads are issued before the first encode, so a wave keeps ROWS KiB in flight
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y0 = blockIdx.y * (4 * ROWS) + threadIdx.y;
    if (x >= p.width || y0 >= p.height) return;
    v4f h[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++) h[j] = __builtin_nontemporal_load((const v4f *)p.hdr + (size_t)min(y0 + 4 * j, p.height - 1) * p.hdr_pitch + x);
#pragma unroll
    for (int j = 0; j < ROWS; j++)
        if (y0 + 4 * j < p.height)
            __builtin_nontemporal_store(encode_ldr<FMT>(h[j].x, h[j].y, h[j].z, p.lut11, p.lut10), p.out + (size_t)(y0 + 4 * j) * p.out_pitch + x);
}

#if !MIL_STRICT
// Brute-force count of in-radius (pixel, point light) pairs: the FLOP-roofline denominator.
__global__ __launch_bounds__(256) void k_count_pairs(const ShadeParams p, unsigned long long *count)
{
    uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    unsigned long long n = 0;
    if (x < p.width && y < p.height) {
        size_t i = (size_t)y * p.pitch + x;
        Px q;
        setup_pixel(q, p, pixel_position(p, x, y + row_map(p, blockIdx.y * 4).g, p.depth[i]), p.normal[i], p.albedo[i], p.rm[i]);
        if (q.occ > 0.0f)
            for (uint32_t l = 0; l < p.n_point; l++) {
                v4f a = ldu(p.point_prep, 2 * l);
                f3 Lv;
                n += point_d2(q, a, Lv) < a.w ? 1u : 0u;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o, 64);
    if (((threadIdx.y * 64 + threadIdx.x) & 63u) == 0 && n) atomicAdd(count, n);
}
#endif

// ---------------------------------------------------------------------------------
// K5 (next row f-1): FXAA, renderer/shaders/fs_fxaa.glsl:22-91 (draw: render/mod.rs:312-329), from
// the ldr attachment (B10G11R11_UFLOAT) to the swapchain image (B8G8R8A8_SRGB).  Sampler: NEAREST,
// CLAMP_TO_EDGE (render/fxaa/mod.rs:76-88).  One pixel per lane, block 64 x 4; the nine taps are plain
// dword loads (neighbouring lanes share lines in L1).  A B10G11R11 channel is a bit-field of an FP16
// number (same 5-bit exponent and bias, shorter mantissa), so a texel decodes EXACTLY with one shift/mask
// and one v_cvt_f32_f16 per channel, denormals included.
// ---------------------------------------------------------------------------------
DEV f3 fxaa_decode(uint32_t u)
{
    typedef _Float16 h1;
    union { unsigned short s; h1 h; } r, g, b;
    r.s = (unsigned short)((u & 0x7FFu) << 4);
    g.s = (unsigned short)((u >> 7) & 0x7FF0u);
    b.s = (unsigned short)((u >> 17) & 0x7FE0u);
    return f3{ (float)r.h, (float)g.h, (float)b.h };
}
DEV f3 fxaa_texel(const FxaaParams &p, int ix, int iy)
{
    ix = min(max(ix, 0), (int)p.width - 1); iy = min(max(iy, 0), (int)p.height - 1);
    return fxaa_decode(p.ldr[(size_t)iy * p.ldr_pitch + ix]);
}
#if MIL_STRICT
DEV f3 fxaa_tap(const FxaaParams &p, float u, float v)     // texture(tex, uv): floor(uv * size), clamp to edge
{
    float fx = floorf(u * (float)p.width), fy = floorf(v * (float)p.height);
    int ix = fx < 0.0f ? 0 : (fx > (float)(p.width - 1) ? (int)p.width - 1 : (int)fx);
    int iy = fy < 0.0f ? 0 : (fy > (float)(p.height - 1) ? (int)p.height - 1 : (int)fy);
    return fxaa_decode(p.ldr[(size_t)iy * p.ldr_pitch + ix]);
}
DEV float fxaa_luma(f3 c) { return (c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f; }
#else
DEV float fxaa_luma(f3 c) { return __builtin_fmaf(c.z, 0.114f, __builtin_fmaf(c.y, 0.587f, c.x * 0.299f)); }
#endif

__global__ __launch_bounds__(256) void k_fxaa(const FxaaParams p)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, yy = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.width || yy >= p.row_count) return;
    const uint32_t y = p.row_begin + yy;
    const float K1 = (float)(1.0 / 3.0 - 0.5), K2 = (float)(2.0 / 3.0 - 0.5), KRED = 0.03125f, REDUCE_MIN = 0.0078125f, SPAN = 8.0f;
    f3 a, b;
    float lmin, lmax;
#if MIL_STRICT
    const float ivx = 1.0f / (float)p.width, ivy = 1.0f / (float)p.height;      // fs_fxaa.glsl:23
    const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
    const f3 nw = fxaa_tap(p, (fx + -1.0f) * ivx, (fy + -1.0f) * ivy), ne = fxaa_tap(p, (fx + 1.0f) * ivx, (fy + -1.0f) * ivy);
    const f3 sw = fxaa_tap(p, (fx + -1.0f) * ivx, (fy + 1.0f) * ivy), se = fxaa_tap(p, (fx + 1.0f) * ivx, (fy + 1.0f) * ivy);
    const float umx = fx * ivx, umy = fy * ivy;
    const f3 m = fxaa_tap(p, umx, umy);
    const float lNW = fxaa_luma(nw), lNE = fxaa_luma(ne), lSW = fxaa_luma(sw), lSE = fxaa_luma(se), lM = fxaa_luma(m);
    lmin = fminf(lM, fminf(fminf(lNW, lNE), fminf(lSW, lSE)));                   // :49-50
    lmax = fmaxf(lM, fmaxf(fmaxf(lNW, lNE), fmaxf(lSW, lSE)));
    float dx = -((lNW + lNE) - (lSW + lSE)), dy = ((lNW + lSW) - (lNE + lSE));    // :53-54
    const float reduce = fmaxf((((lNW + lNE) + lSW) + lSE) * KRED, REDUCE_MIN);   // :56-57
    const float rcpmin = 1.0f / (fminf(fabsf(dx), fabsf(dy)) + reduce);           // :59
    dx = fminf(SPAN, fmaxf(-SPAN, dx * rcpmin)) * ivx;                             // :60-62
    dy = fminf(SPAN, fmaxf(-SPAN, dy * rcpmin)) * ivy;
    const f3 t0 = fxaa_tap(p, umx + dx * K1, umy + dy * K1), t1 = fxaa_tap(p, umx + dx * K2, umy + dy * K2);   // :64
    const f3 t2 = fxaa_tap(p, umx + dx * -0.5f, umy + dy * -0.5f), t3 = fxaa_tap(p, umx + dx * 0.5f, umy + dy * 0.5f);   // :65
    a = f3{ 0.5f * (t0.x + t1.x), 0.5f * (t0.y + t1.y), 0.5f * (t0.z + t1.z) };
    b = f3{ a.x * 0.5f + 0.25f * (t2.x + t3.x), a.y * 0.5f + 0.25f * (t2.y + t3.y), a.z * 0.5f + 0.25f * (t2.z + t3.z) };
#else
    // The five fixed taps are the integer neighbours ((x + 0.5 +- 1) / W * W floors to x +- 1); the four
    // directional taps are formed in pixel units, without the round trip through normalised coordinates.
    const int xi = (int)x, yi = (int)y;
    const f3 nw = fxaa_texel(p, xi - 1, yi - 1), ne = fxaa_texel(p, xi + 1, yi - 1);
    const f3 sw = fxaa_texel(p, xi - 1, yi + 1), se = fxaa_texel(p, xi + 1, yi + 1);
    const f3 m = fxaa_decode(p.ldr[(size_t)y * p.ldr_pitch + x]);
    const float lNW = fxaa_luma(nw), lNE = fxaa_luma(ne), lSW = fxaa_luma(sw), lSE = fxaa_luma(se), lM = fxaa_luma(m);
    lmin = fminf(lM, fminf(fminf(lNW, lNE), fminf(lSW, lSE)));
    lmax = fmaxf(lM, fmaxf(fmaxf(lNW, lNE), fmaxf(lSW, lSE)));
    float dx = (lSW + lSE) - (lNW + lNE), dy = (lNW + lSW) - (lNE + lSE);
    const float reduce = fmaxf((lNW + lNE + lSW + lSE) * KRED, REDUCE_MIN);
    const float rcpmin = __builtin_amdgcn_rcpf(fminf(fabsf(dx), fabsf(dy)) + reduce);
    dx = __builtin_amdgcn_fmed3f(dx * rcpmin, -SPAN, SPAN);
    dy = __builtin_amdgcn_fmed3f(dy * rcpmin, -SPAN, SPAN);
    a = m; b = m;
    // |dir| < 1 pixel: the four taps at dir * {-1/2, -1/6, 1/6, 1/2} all stay inside the centre texel, so
    // rgbA = rgbB = M exactly (0.5 M + 0.25 (M + M)).  Smooth regions are most of a frame: when it holds for
    // the whole wave the taps and the blend are skipped.
    if (__builtin_amdgcn_ballot_w64(fmaxf(fabsf(dx), fabsf(dy)) > 0.99f) != 0) {
        const float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
        auto tap = [&](float k) {
            return fxaa_texel(p, (int)floorf(__builtin_fmaf(dx, k, fx)), (int)floorf(__builtin_fmaf(dy, k, fy)));
        };
        const f3 t0 = tap(K1), t1 = tap(K2), t2 = tap(-0.5f), t3 = tap(0.5f);
        a = f3{ 0.5f * (t0.x + t1.x), 0.5f * (t0.y + t1.y), 0.5f * (t0.z + t1.z) };
        b = f3{ __builtin_fmaf(0.25f, t2.x + t3.x, a.x * 0.5f), __builtin_fmaf(0.25f, t2.y + t3.y, a.y * 0.5f),
                __builtin_fmaf(0.25f, t2.z + t3.z, a.z * 0.5f) };
    }
#endif
    const float lB = fxaa_luma(b);
    const f3 o = (lB < lmin || lB > lmax) ? a : b;                                 // :67-71
    __builtin_nontemporal_store(srgb8_direct(sat(o.z)) | (srgb8_direct(sat(o.y)) << 8) | (srgb8_direct(sat(o.x)) << 16) | 0xFF000000u,
                                p.out + (size_t)yy * p.out_pitch + x);
}

#if !MIL_STRICT
// Fast FXAA: each lane owns ONE column and R output rows (y .. y + R - 1; R = 2 or 4); it decodes the R + 2 texels of
// its column in rows y - 1 .. y + R once, and the diagonal neighbours' lumas come from lanes l - 1 / l + 1 by DPP wave
// shifts instead of four more loads + decodes each.  Lanes 0 and 63 are halo lanes (columns x0 - 1, x0 + 62), so
// a wave produces 62 columns.  Per-pixel arithmetic is float2 (packed, row pairs).  Same texel picks as k_fxaa.
template <int CTRL> DEV float wave_shift(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int R>
__global__ __launch_bounds__(256) void k_fxaa2(const FxaaParams p)
{
    const int lane = (int)threadIdx.x;
    const int col = (int)(blockIdx.x * 62) + lane - 1;                 // may be -1 or >= width: clamp to edge
    // the wave's rows are wave-uniform: row clamps and row base addresses are scalar work, the lane adds its column
    const uint32_t yy = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.y * 4 + threadIdx.y) * R));   // first of the R rows, strip-local
    if (yy >= p.row_count) return;                                      // whole wave
    const int y = (int)(p.row_begin + yy);
    const int xc = min(max(col, 0), (int)p.width - 1), H1 = (int)p.height - 1;
    f3 c[R + 2];
    float L[R + 2];
    uint32_t u[R + 2];
#pragma unroll
    for (int r = 0; r < R + 2; r++) u[r] = (p.ldr + (size_t)min(max(y - 1 + r, 0), H1) * p.ldr_pitch)[(uint32_t)xc];
#pragma unroll
    for (int r = 0; r < R + 2; r++) {
        c[r] = fxaa_decode(u[r]);
        L[r] = fxaa_luma(c[r]);
    }
    // A pixel whose edge direction is shorter than one pixel keeps its centre texel (below), which is an R11G11B10F
    // code: its sRGB8 value comes from the two tables of every code <= 1.0 (exact, the tonemap's VIA_LDR tables)
    // instead of three log/exp evaluations.  Per pixel, not per wave, so a strip equals the rows of the whole frame.
    uint32_t wl[R];
#pragma unroll
    for (int e = 0; e < R; e++) {
        const uint32_t t = u[1 + e];
        wl[e] = (uint32_t)p.lut10[min(t >> 22, 0x1E0u)] | ((uint32_t)p.lut11[min((t >> 11) & 0x7FFu, 0x3C0u)] << 8) |
                ((uint32_t)p.lut11[min(t & 0x7FFu, 0x3C0u)] << 16) | 0xFF000000u;
    }
    float Lw[R + 2], Le[R + 2];                                          // the same rows, one column to the west / east
#pragma unroll
    for (int r = 0; r < R + 2; r++) { Lw[r] = wave_shift<0x138>(L[r]); Le[r] = wave_shift<0x130>(L[r]); }   // wave_shr:1 / wave_shl:1
    const float KRED = 0.03125f, REDUCE_MIN = 0.0078125f, SPAN = 8.0f;
    const float K1 = (float)(1.0 / 3.0 - 0.5), K2 = (float)(2.0 / 3.0 - 0.5);
    const bool live = lane >= 1 && lane <= 62 && col < (int)p.width;
    float dxs[R], dys[R], mns[R], mxs[R];
    bool edge[R];
#pragma unroll
    for (int j = 0; j < R; j += 2) {                                     // output rows j, j + 1 as one float2
        const v2f lNW = { Lw[j], Lw[j + 1] }, lNE = { Le[j], Le[j + 1] }, lSW = { Lw[j + 2], Lw[j + 3] }, lSE = { Le[j + 2], Le[j + 3] };
        const v2f lM = { L[j + 1], L[j + 2] };
        mns[j] = fminf(lM.x, fminf(fminf(lNW.x, lNE.x), fminf(lSW.x, lSE.x))); mns[j + 1] = fminf(lM.y, fminf(fminf(lNW.y, lNE.y), fminf(lSW.y, lSE.y)));
        mxs[j] = fmaxf(lM.x, fmaxf(fmaxf(lNW.x, lNE.x), fmaxf(lSW.x, lSE.x))); mxs[j + 1] = fmaxf(lM.y, fmaxf(fmaxf(lNW.y, lNE.y), fmaxf(lSW.y, lSE.y)));
        v2f dx = (lSW + lSE) - (lNW + lNE), dy = (lNW + lSW) - (lNE + lSE);
        const v2f red = max_((lNW + lNE + lSW + lSE) * KRED, (v2f)(REDUCE_MIN));
        const v2f rcpmin = rcp_(v2f{ fminf(fabsf(dx.x), fabsf(dy.x)), fminf(fabsf(dx.y), fabsf(dy.y)) } + red);
        dx = med3_(dx * rcpmin, -SPAN, SPAN);
        dy = med3_(dy * rcpmin, -SPAN, SPAN);
        dxs[j] = dx.x; dxs[j + 1] = dx.y; dys[j] = dy.x; dys[j + 1] = dy.y;
        edge[j] = fmaxf(fabsf(dx.x), fabsf(dy.x)) > 0.99f; edge[j + 1] = fmaxf(fabsf(dx.y), fabsf(dy.y)) > 0.99f;
    }
    bool any_edge = false;
#pragma unroll
    for (int e = 0; e < R; e++) any_edge = any_edge || edge[e];
    if (__builtin_amdgcn_ballot_w64(live && any_edge) != 0) {
#pragma unroll
        for (int e = 0; e < R; e++) {
            if (__builtin_amdgcn_ballot_w64(live && edge[e]) == 0) continue;   // this row of the wave has no edge
            const float ddx = dxs[e], ddy = dys[e];
            const float fx = (float)col + 0.5f, fy = (float)(y + e) + 0.5f;
            auto tap = [&](float k) {          // pixel offsets fit 30 bits (checked on the host): one 24-bit mad + scalar base
                const int ix = min(max((int)floorf(__builtin_fmaf(ddx, k, fx)), 0), (int)p.width - 1);
                const int iy = min(max((int)floorf(__builtin_fmaf(ddy, k, fy)), 0), H1);
                const uint32_t off = __umul24((uint32_t)iy, p.ldr_pitch) + (uint32_t)ix;
                __builtin_assume(off < (1u << 30));
                return fxaa_decode(p.ldr[off]);
            };
            const f3 t0 = tap(K1), t1 = tap(K2), t2 = tap(-0.5f), t3 = tap(0.5f);
            const f3 a = { 0.5f * (t0.x + t1.x), 0.5f * (t0.y + t1.y), 0.5f * (t0.z + t1.z) };
            const f3 b = { __builtin_fmaf(0.25f, t2.x + t3.x, a.x * 0.5f), __builtin_fmaf(0.25f, t2.y + t3.y, a.y * 0.5f),
                           __builtin_fmaf(0.25f, t2.z + t3.z, a.z * 0.5f) };
            const float lB = fxaa_luma(b);
            const f3 o = (lB < mns[e] || lB > mxs[e]) ? a : b;
            const uint32_t w = srgb8_direct(sat(o.z)) | (srgb8_direct(sat(o.y)) << 8) | (srgb8_direct(sat(o.x)) << 16) | 0xFF000000u;
            if (edge[e]) wl[e] = w;
        }
    }
    if (!live) return;
#pragma unroll
    for (int e = 0; e < R; e++) {
        if (yy + e >= p.row_count) break;
        __builtin_nontemporal_store(wl[e], (p.out + (size_t)(yy + e) * p.out_pitch) + (uint32_t)col);
    }
}
#endif

// ---------------------------------------------------------------------------------
// K6 (next row f-4): Hosek-Wilkie sky subpass as a screen-space pass, renderer/shaders/sky_hosek_frag.glsl:28-60
// (draw render/mod.rs:221-226; depth test LessOrEqual at z = w - 1e-5, render/hosek/mod.rs:58-64,
// sky_hosek_vert.glsl:19-20): texels still at the clear depth get hdr = (radiance * 0.05, 1), the others keep
// hdr.  The interpolated mesh position of the fragment lies on the pixel's view ray, so the view direction is
// the normalised PositionFromDepth ray.  Block 64 x 4; a wave without a sky texel returns after one load.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sky(const SkyParams p)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.width || y >= p.height) return;
    const float d = p.depth[(size_t)y * p.depth_pitch + x];
    const bool sky = d >= MIL_SKY_DEPTH;
    if (__builtin_amdgcn_ballot_w64(sky) == 0) return;
    float r3[3];
    sky_radiance(p, x, y + p.y_offset, d, r3);
    if (sky) __builtin_nontemporal_store(v4f{ r3[0], r3[1], r3[2], 1.0f }, (v4f *)p.hdr + (size_t)y * p.hdr_pitch + x);
}

// ---------------------------------------------------------------------------------
// K7 (next row f-2): transparency resolve, renderer/shaders/fs_mcguire13_resolve.glsl:8-13 with the blend of
// render/mcguire13/mod.rs:108-120 (draw render/mod.rs:282-294): hdr = src * (1 - r) + hdr * r, src =
// (accum.rgb / clamp(accum.a, 1e-4, 5e4), r).  Streaming: 8 + 2 + 16 B read, 16 B written per pixel; a wave
// whose 64 texels all have revealage 1 and accum 0 (nothing transparent there) leaves hdr untouched:
// 0 * 0 + hdr * 1 = hdr exactly for finite hdr.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(const ResolveParams p)
{
    const uint32_t x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= p.width || y >= p.height) return;
    const h4 a = __builtin_nontemporal_load((const h4 *)p.accum + (size_t)y * p.accum_pitch + x);
    const float r = (float)__builtin_nontemporal_load((const _Float16 *)p.reveal + (size_t)y * p.reveal_pitch + x);
#if !MIL_STRICT
    if (__builtin_amdgcn_ballot_w64(!resolve_untouched(a, r)) == 0) return;
#endif
    v4f *dst = (v4f *)p.hdr + (size_t)y * p.hdr_pitch + x;
    const v4f o = resolve_blend(a, r, *dst);
    __builtin_nontemporal_store(o, dst);
}

// ---------------------------------------------------------------------------------
// K8 (next row f-2, second half): the fragment shader of the transparency accumulate draw,
// renderer/shaders/fs_mcguire13_accumulation.glsl:100-126 (draw render/mod.rs:229-281), over a fragment list:
// one fragment per lane, 52 B read + 20 B written per fragment, all 16-byte accesses of consecutive lanes
// contiguous.  Unlike light(): no NdotL on the diffuse term, no intensity, no occlusion, roughness unclamped.
// ---------------------------------------------------------------------------------
DEV DirRec accum_dir_rec(const AccumParams &p, uint32_t l)
{
    DirRec d;
    if (l < DIR_INLINE) {
        const float *r = p.dir_inline[l];
        d.a = v4f{ r[0], r[1], r[2], r[3] }; d.b = v4f{ r[4], r[5], r[6], r[7] };
    } else {
        d.a = ldu(p.dir_raw, 2 * l); d.b = ldu(p.dir_raw, 2 * l + 1);
    }
    return d;
}
__global__ __launch_bounds__(256) void k_accumulate(const AccumParams p)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.count) return;
    const v4f pz = __builtin_nontemporal_load((const v4f *)p.pos_z + i), no = __builtin_nontemporal_load((const v4f *)p.nrm_op + i);
    const v4f ar = __builtin_nontemporal_load((const v4f *)p.alb_rough + i);
    const float metallic = __builtin_nontemporal_load(p.metallic + i), opacity = no.w;
    const f3 albedo = { ar.x, ar.y, ar.z };
    const float alpha = ar.w * ar.w;                                                  // :104
    f3 lighting = { 0.0f, 0.0f, 0.0f };
#if MIL_STRICT
    const f3 N = normalize3(f3{ no.x, no.y, no.z });                                  // :106
    const f3 V = normalize3(f3{ p.cam[0] - pz.x, p.cam[1] - pz.y, p.cam[2] - pz.z }); // :107
    for (uint32_t l = 0; l < p.n_dir; l++) {                                          // :110-119
        const DirRec d = accum_dir_rec(p, l);
        const f3 L = { d.a.x, d.a.y, d.a.z };
        const f3 H = normalize3(f3{ L.x + V.x, L.y + V.y, L.z + V.z });
        const float NdotV = clampf(dot3(N, V), 0.0001f, 1.0f), NdotL = clampf(dot3(N, L), 0.0f, 1.0f);
        const float NdotH = clampf(dot3(N, H), 0.0f, 1.0f), LdotH = clampf(dot3(L, H), 0.0f, 1.0f);
        const f3 spec = brdf_specular(alpha, albedo, metallic, NdotV, NdotL, NdotH, LdotH);
        lighting.x = lighting.x + (albedo.x / PI_LIT + (spec.x * d.b.x) * NdotL);
        lighting.y = lighting.y + (albedo.y / PI_LIT + (spec.y * d.b.y) * NdotL);
        lighting.z = lighting.z + (albedo.z / PI_LIT + (spec.z * d.b.z) * NdotL);
    }
    const float f = 1.0f - pz.w;                                                      // w10, :84-87
    const float w = opacity * fmaxf(0.01f, ((3000.0f * f) * f) * f);
    const v4f acc = { (lighting.x * opacity) * w, (lighting.y * opacity) * w, (lighting.z * opacity) * w, opacity * w };
#else
    // light-invariant terms hoisted as in setup_pixel_t; the n_dir diffuse terms are added once at the end
    const f3 n = { no.x, no.y, no.z }, D = { p.cam[0] - pz.x, p.cam[1] - pz.y, p.cam[2] - pz.z };
    const float rn = rsq_(dot3(n, n)), rv = rsq_(dot3(D, D));
    const f3 N = { n.x * rn, n.y * rn, n.z * rn }, V = { D.x * rv, D.y * rv, D.z * rv };
    const float a2 = alpha * alpha, oma2 = 1.0f - a2;
    const float NdotV = med3_(dot3(N, V), 0.0001f, 1.0f);
    const float sV = sqrt_(fma_(fma_(-a2, NdotV, NdotV), NdotV, a2));
    const float om = 1.0f - metallic;
    const f3 F0 = { fma_(albedo.x, metallic, 0.04f * om), fma_(albedo.y, metallic, 0.04f * om), fma_(albedo.z, metallic, 0.04f * om) };
    const float f90 = sat((F0.x + F0.y + F0.z) * 16.5f);
    const f3 dF = { f90 - F0.x, f90 - F0.y, f90 - F0.z };
    for (uint32_t l = 0; l < p.n_dir; l++) {
        const DirRec d = accum_dir_rec(p, l);
        const f3 L = { d.a.x, d.a.y, d.a.z }, H = { L.x + V.x, L.y + V.y, L.z + V.z };
        const float rh = rsq_(max_(dot3(H, H), 1e-30f));                              // L = -V: see eval_light
        const float NdotL = sat(dot3(N, L)), LdotH = sat(dot3(L, H) * rh);
        const float k = alpha * rcp_(fma_(one_minus_ndoth2<float>(N, H, rh), oma2, a2));   // see eval_tail; alpha = 0 at N.H = 1 is 0 / 0 as in the shader (:104 does not clamp)
        const float sL = sqrt_(fma_(fma_(-a2, NdotL, NdotL), NdotL, a2));
        const float den = fma_(NdotL, sV, NdotV * sL);
        const float t = ((k * k) * ((0.5f / PI_LIT) * rcp_(den))) * NdotL;            // D * Vis * NdotL
        const float w1 = 1.0f - LdotH, w2 = w1 * w1, w5 = w2 * w2 * w1;
        lighting.x = fma_(fma_(dF.x, w5, F0.x) * d.b.x, t, lighting.x);
        lighting.y = fma_(fma_(dF.y, w5, F0.y) * d.b.y, t, lighting.y);
        lighting.z = fma_(fma_(dF.z, w5, F0.z) * d.b.z, t, lighting.z);
    }
    const float nd = (float)p.n_dir * (1.0f / PI_LIT);
    lighting = f3{ fma_(albedo.x, nd, lighting.x), fma_(albedo.y, nd, lighting.y), fma_(albedo.z, nd, lighting.z) };
    const float f = 1.0f - pz.w;
    const float aw = opacity * (opacity * max_(0.01f, 3000.0f * f * f * f));
    const v4f acc = { lighting.x * aw, lighting.y * aw, lighting.z * aw, aw };
#endif
    __builtin_nontemporal_store(acc, (v4f *)p.accum + i);
    __builtin_nontemporal_store(opacity, p.reveal + i);
}

// ---------------------------------------------------------------------------------
// launchers (host).  Each returns the status of its own hipLaunchKernel call: hipGetLastError would also return (and
// clear) errors that other HIP users of the calling thread left behind.
// ---------------------------------------------------------------------------------
template <typename P> static inline hipError_t launch_k(void (*kernel)(P), dim3 grid, dim3 block, size_t lds, hipStream_t s, const P &p)
{
    void *args[] = { (void *)&p };
    return hipLaunchKernel((const void *)kernel, grid, block, args, lds, s);
}
hipError_t launch_stream(int fmt, const ShadeParams &p, bool frame, hipStream_t s)
{
    const bool ext = p.has_shadow || p.n_point || p.has_ambient;
#if !MIL_STRICT
    if (!(p.tune & 256u) && p.rows_per_block != 4) {      // two pixels per lane, packed (MIL_TUNE bit 256: the one-pixel kernel)
        dim3 block(64, 4), grid((p.width + 63) / 64, (p.height + 7) / 8);
#define MIL_LAUNCH_STREAM2(F)                                                                  \
        if (frame) return launch_k(k_stream2<F, true, true>, grid, block, 0, s, p);         \
        else if (ext) return launch_k(k_stream2<F, true, false>, grid, block, 0, s, p);     \
        else return launch_k(k_stream2<F, false, false>, grid, block, 0, s, p);
        switch (fmt) {
        case FMT_HDR: MIL_LAUNCH_STREAM2(FMT_HDR); break;
        case FMT_LDR: MIL_LAUNCH_STREAM2(FMT_LDR); break;
        case FMT_SRGB8: MIL_LAUNCH_STREAM2(FMT_SRGB8); break;
        default: MIL_LAUNCH_STREAM2(FMT_SRGB8_VIA_LDR); break;
        }
#undef MIL_LAUNCH_STREAM2
    }
#endif
    const bool four = p.rows_per_block == 4 && !frame;
    dim3 block(64, 4), grid((p.width + 63) / 64, four ? (p.height + 15) / 16 : (p.height + 3) / 4);
#define MIL_LAUNCH_STREAM(F)                                                                     \
    if (frame) return launch_k(k_stream<F, 1, true, true>, grid, block, 0, s, p);             \
    else if (four) { if (ext) return launch_k(k_stream<F, 4, true, false>, grid, block, 0, s, p);  \
                     else return launch_k(k_stream<F, 4, false, false>, grid, block, 0, s, p); }   \
    else { if (ext) return launch_k(k_stream<F, 1, true, false>, grid, block, 0, s, p);       \
           else return launch_k(k_stream<F, 1, false, false>, grid, block, 0, s, p); }
    switch (fmt) {
    case FMT_HDR: MIL_LAUNCH_STREAM(FMT_HDR); break;
    case FMT_LDR: MIL_LAUNCH_STREAM(FMT_LDR); break;
    case FMT_SRGB8: MIL_LAUNCH_STREAM(FMT_SRGB8); break;
    default: MIL_LAUNCH_STREAM(FMT_SRGB8_VIA_LDR); break;
    }
#undef MIL_LAUNCH_STREAM
    return hipErrorInvalidValue;    // not reached: every case returns
}
hipError_t launch_tiled(int fmt, const ShadeParams &p, bool frame, hipStream_t s)
{
    const uint32_t wpb = (p.tune & 2u) ? 1 : ((p.tune & 4u) ? 2 : 4);
    dim3 block(64 * wpb), grid(((p.tiles_x + wpb - 1) / wpb) * p.tiles_y);
    const size_t lds = (size_t)wpb * tiled_lds(p.n_point, p.has_shadow && p.shadow_lds && p.n_dir, frame && p.accum && p.resolve_dma).per_wave;
    const bool stage = p.has_shadow && p.shadow_lds && p.n_dir;
#define MIL_LAUNCH_TILED(F)                                                             \
    if (frame) return stage ? launch_k(k_tiled<F, true, true>, grid, block, lds, s, p) : launch_k(k_tiled<F, true, false>, grid, block, lds, s, p);   \
    else return stage ? launch_k(k_tiled<F, false, true>, grid, block, lds, s, p) : launch_k(k_tiled<F, false, false>, grid, block, lds, s, p);
    switch (fmt) {
    case FMT_HDR: MIL_LAUNCH_TILED(FMT_HDR); break;
    case FMT_LDR: MIL_LAUNCH_TILED(FMT_LDR); break;
    case FMT_SRGB8: MIL_LAUNCH_TILED(FMT_SRGB8); break;
    default: MIL_LAUNCH_TILED(FMT_SRGB8_VIA_LDR); break;
    }
#undef MIL_LAUNCH_TILED
    return hipErrorInvalidValue;    // not reached: every case returns
}
hipError_t launch_tonemap(int fmt, const TonemapParams &p, hipStream_t s)
{
    const uint32_t rows = p.rows == 1 || p.rows == 4 ? p.rows : 2;
    dim3 block(64, 4), grid((p.width + 63) / 64, (p.height + 4 * rows - 1) / (4 * rows));
#define MIL_LAUNCH_TONEMAP(F)                                                               \
    if (rows == 1) return launch_k(k_tonemap<F, 1>, grid, block, 0, s, p);               \
    else if (rows == 2) return launch_k(k_tonemap<F, 2>, grid, block, 0, s, p);          \
    else return launch_k(k_tonemap<F, 4>, grid, block, 0, s, p);
    switch (fmt) {
    case FMT_LDR: MIL_LAUNCH_TONEMAP(FMT_LDR); break;
    case FMT_SRGB8: MIL_LAUNCH_TONEMAP(FMT_SRGB8); break;
    default: MIL_LAUNCH_TONEMAP(FMT_SRGB8_VIA_LDR); break;
    }
#undef MIL_LAUNCH_TONEMAP
    return hipErrorInvalidValue;    // not reached: every case returns
}
hipError_t launch_fxaa(const FxaaParams &p, hipStream_t s)
{
#if MIL_STRICT
    dim3 block(64, 4), grid((p.width + 63) / 64, (p.row_count + 3) / 4);
    return launch_k(k_fxaa, grid, block, 0, s, p);
#else
    if (p.rows == 2) {
        dim3 block(64, 4), grid((p.width + 61) / 62, (p.row_count + 7) / 8);
        return launch_k(k_fxaa2<2>, grid, block, 0, s, p);
    } else {
        dim3 block(64, 4), grid((p.width + 61) / 62, (p.row_count + 15) / 16);
        return launch_k(k_fxaa2<4>, grid, block, 0, s, p);
    }
#endif
}
hipError_t launch_sky(const SkyParams &p, hipStream_t s)
{
    dim3 block(64, 4), grid((p.width + 63) / 64, (p.height + 3) / 4);
    return launch_k(k_sky, grid, block, 0, s, p);
}
hipError_t launch_resolve(const ResolveParams &p, hipStream_t s)
{
    dim3 block(64, 4), grid((p.width + 63) / 64, (p.height + 3) / 4);
    return launch_k(k_resolve, grid, block, 0, s, p);
}
hipError_t launch_accumulate(const AccumParams &p, hipStream_t s)
{
    return launch_k(k_accumulate, dim3((p.count + 255) / 256), dim3(256), 0, s, p);
}
#if !MIL_STRICT
hipError_t launch_bin(int fmt, const ShadeParams &p, hipStream_t s)
{
    const uint32_t macros = ((p.tiles_x + BIN_TX - 1) / BIN_TX) * ((p.tiles_y + BIN_TY - 1) / BIN_TY);
    void *args[] = { (void *)&p, (void *)&fmt };
    return hipLaunchKernel((const void *)k_bin, dim3(macros), dim3(64 * BIN_WAVES), args, 0, s);
}
hipError_t launch_order(const ShadeParams &p, hipStream_t s)
{
    const uint32_t items = 2u * p.tiles_x * p.tiles_y, wgs = (items + ORDER_THREADS - 1) / ORDER_THREADS;
    return launch_k(k_order, dim3(wgs < 64u ? (wgs ? wgs : 1u) : 64u), dim3(ORDER_THREADS), 0, s, p);
}
hipError_t launch_shade(int fmt, const ShadeParams &p, bool frame, hipStream_t s)
{
    const dim3 grid((2u * p.tiles_x * p.tiles_y + 3u) / 4u), block(256);        // upper bound: every half tile listed
#define MIL_LAUNCH_SHADE(F)                                                   \
    if (frame) return launch_k(k_shade<F, true>, grid, block, 0, s, p);     \
    else return launch_k(k_shade<F, false>, grid, block, 0, s, p);
    switch (fmt) {
    case FMT_HDR: MIL_LAUNCH_SHADE(FMT_HDR); break;
    case FMT_LDR: MIL_LAUNCH_SHADE(FMT_LDR); break;
    case FMT_SRGB8: MIL_LAUNCH_SHADE(FMT_SRGB8); break;
    default: MIL_LAUNCH_SHADE(FMT_SRGB8_VIA_LDR); break;
    }
#undef MIL_LAUNCH_SHADE
    return hipErrorInvalidValue;    // not reached: every case returns
}
hipError_t launch_count_pairs(const ShadeParams &p, unsigned long long *count, hipStream_t s)
{
    dim3 block(64, 4), grid((p.width + 63) / 64, (p.height + 3) / 4);
    void *args[] = { (void *)&p, (void *)&count };
    return hipLaunchKernel((const void *)k_count_pairs, grid, block, args, 0, s);
}
#endif

} // namespace MIL_NS
} // namespace mil
