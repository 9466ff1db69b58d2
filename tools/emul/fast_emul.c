/*
 * CPU emulation of the FAST kernels' per-pixel arithmetic (csrc/mil_device.hpp, MIL_STRICT=0): same fma chains in float,
 * v_rcp / v_rsq / v_sqrt modelled as the correctly rounded result optionally perturbed by +-1 ulp (the hardware bound).
 * A development tool for the numerics of eval_light / eval_tail: lets tools/fast_emul.py run the tolerance census of
 * tests/test_gpu_configs.py::test_headline_4k_bare_tolerance_census on the CPU for a candidate formula before a GPU run.
 * Not the product, not the oracle.  gcc -O2 -fopenmp -ffp-contract=off -shared -fPIC.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define PI_LIT 3.14159f
typedef struct { float x, y, z; } f3;

static inline uint32_t rng(uint32_t *s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static int g_perturb = 0;
static inline float ulp_jitter(float v, uint32_t *s)
{
    if (!g_perturb) return v;
    uint32_t r = rng(s) % 3u;
    union { float f; uint32_t u; } c; c.f = v;
    if (r == 1) c.u += 1; else if (r == 2) c.u -= 1;
    return c.f;
}
static inline float rcp_(float x, uint32_t *s) { return ulp_jitter((float)(1.0 / (double)x), s); }
static inline float rsq_(float x, uint32_t *s) { return ulp_jitter((float)(1.0 / sqrt((double)x)), s); }
static inline float sqrt_(float x, uint32_t *s) { return ulp_jitter((float)sqrt((double)x), s); }
static inline float sat(float x) { return x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x); }
static inline float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }

typedef struct {
    f3 N, V, P;
    float occ, alpha, a2, am1, NdotV, sV;
    f3 G0, G1, dif;
} Px;

static void eval_light(const Px *q, f3 L, f3 col, float scale, f3 *res, int variant, uint32_t *s)
{
    f3 H = { L.x + q->V.x, L.y + q->V.y, L.z + q->V.z };
    float rh = rsq_(fmaxf(dot3(H, H), 1e-30f), s);
    float NdotL = sat(fmaf(q->N.z, L.z, fmaf(q->N.y, L.y, q->N.x * L.x)));
    float LdotH = sat(dot3(L, H) * rh);
    float t;
    if (variant == 0) {
        float NdotH = sat(dot3(q->N, H) * rh);
        float x2 = NdotH * NdotH;
        t = fmaxf(fmaf(x2, q->am1, 1.0f), q->a2);
    } else {
        f3 d = { fmaf(-H.x, rh, q->N.x), fmaf(-H.y, rh, q->N.y), fmaf(-H.z, rh, q->N.z) };
        float e = dot3(d, d);
        float h = sat(e * 0.5f);
        float s2 = h * (2.0f - h);
        t = fmaf(s2, -q->am1, q->a2);
    }
    float sL = sqrt_(fmaf(fmaf(-q->a2, NdotL, NdotL), NdotL, q->a2), s);
    float den = fmaf(NdotL, q->sV, q->NdotV * sL);
    float u = q->alpha * rcp_(t * den, s);
    float DV = u * (u * (den * (0.5f / PI_LIT)));
    float w = 1.0f - LdotH, w2 = w * w, w5 = w2 * w2 * w;
    float cx = fmaf(fmaf(q->G1.x, w5, q->G0.x), DV, q->dif.x);
    float cy = fmaf(fmaf(q->G1.y, w5, q->G0.y), DV, q->dif.y);
    float cz = fmaf(fmaf(q->G1.z, w5, q->G0.z), DV, q->dif.z);
    float sN = scale * NdotL;
    res->x = fmaf(cx, col.x * sN, res->x);
    res->y = fmaf(cy, col.y * sN, res->y);
    res->z = fmaf(cz, col.z * sN, res->z);
}

/* pos_m: the host-folded 4x4 (row-major [r*4+c]) of fill_common; dir: n_dir x 8 prepared {dir, 0, col*I, 0};
 * point: n_point x 8 prepared {pos, r^2, col*I, 1/r^2} */
void emul_shade(const uint32_t *normal, const uint32_t *albedo_occ, const uint32_t *rough_metal, const float *depth,
                int width, int height, const float *m, const float *cam, const float *dir, int n_dir,
                const float *point, int n_point, const float *ambient, float *hdr, int variant, int perturb)
{
    g_perturb = perturb;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < height; y++) {
        uint32_t seed = 12345u + (uint32_t)y * 7919u;
        for (int x = 0; x < width; x++) {
            size_t i = (size_t)y * width + x;
            uint32_t u1 = normal[i], u2 = albedo_occ[i], u3 = rough_metal[i];
            float xf = (float)x, yf = (float)y, dz = depth[i];
            float vx = fmaf(m[0], xf, fmaf(m[1], yf, fmaf(m[2], dz, m[3])));
            float vy = fmaf(m[4], xf, fmaf(m[5], yf, fmaf(m[6], dz, m[7])));
            float vz = fmaf(m[8], xf, fmaf(m[9], yf, fmaf(m[10], dz, m[11])));
            float vw = fmaf(m[12], xf, fmaf(m[13], yf, fmaf(m[14], dz, m[15])));
            float iw = rcp_(vw, &seed);
            f3 D = { vx * iw, vy * iw, vz * iw };
            Px q;
            q.P = (f3){ D.x + cam[0], D.y + cam[1], D.z + cam[2] };
            const float k10 = 2.0f / 1023.0f, k8 = 1.0f / 255.0f;
            f3 nrm = { fmaf((float)(u1 & 0x3FF), k10, -1.0f), fmaf((float)((u1 >> 10) & 0x3FF), k10, -1.0f), fmaf((float)((u1 >> 20) & 0x3FF), k10, -1.0f) };
            f3 albedo = { (float)(u2 & 0xFF) * k8, (float)((u2 >> 8) & 0xFF) * k8, (float)((u2 >> 16) & 0xFF) * k8 };
            q.occ = (float)(u2 >> 24) * k8;
            float roughness = fmaxf((float)(u3 & 0xFF) * k8, 0.0001f);
            float mt = (float)((u3 >> 8) & 0xFF) * k8;
            float rn = rsq_(dot3(nrm, nrm), &seed);
            q.N = (f3){ nrm.x * rn, nrm.y * rn, nrm.z * rn };
            float rv = -rsq_(dot3(D, D), &seed);
            q.V = (f3){ D.x * rv, D.y * rv, D.z * rv };
            q.alpha = roughness * roughness;
            q.a2 = q.alpha * q.alpha;
            q.am1 = q.a2 - 1.0f;
            float nv = dot3(q.N, q.V);
            q.NdotV = nv < 0.0001f ? 0.0001f : (nv > 1.0f ? 1.0f : nv);
            q.sV = sqrt_(fmaf(fmaf(-q.a2, q.NdotV, q.NdotV), q.NdotV, q.a2), &seed);
            float om = 1.0f - mt;
            f3 F0 = { fmaf(albedo.x, mt, 0.04f * om), fmaf(albedo.y, mt, 0.04f * om), fmaf(albedo.z, mt, 0.04f * om) };
            float f90 = sat((F0.x + F0.y + F0.z) * 16.5f);
            float ds = om * q.occ * (1.0f / PI_LIT);
            q.dif = (f3){ albedo.x * ds, albedo.y * ds, albedo.z * ds };
            f3 tint = { fmaf(albedo.x, mt, om) * q.occ, fmaf(albedo.y, mt, om) * q.occ, fmaf(albedo.z, mt, om) * q.occ };
            q.G0 = (f3){ tint.x * F0.x, tint.y * F0.y, tint.z * F0.z };
            q.G1 = (f3){ tint.x * (f90 - F0.x), tint.y * (f90 - F0.y), tint.z * (f90 - F0.z) };
            f3 r = { 0, 0, 0 };
            if (q.occ > 0.0f) {
                for (int l = 0; l < n_dir; l++) {
                    const float *d = dir + 8 * l;
                    eval_light(&q, (f3){ d[0], d[1], d[2] }, (f3){ d[4], d[5], d[6] }, 1.0f, &r, variant, &seed);
                }
                for (int l = 0; l < n_point; l++) {
                    const float *a = point + 8 * l;
                    f3 Lv = { a[0] - q.P.x, a[1] - q.P.y, a[2] - q.P.z };
                    float d2 = fmaf(Lv.z, Lv.z, fmaf(Lv.y, Lv.y, Lv.x * Lv.x));
                    if (!(d2 < a[3])) continue;
                    float inv = rsq_(d2 + 1e-12f, &seed);
                    f3 L = { Lv.x * inv, Lv.y * inv, Lv.z * inv };
                    float qq = d2 * a[7], w = sat(fmaf(-qq, qq, 1.0f));
                    float att = (w * w) * rcp_(d2 + 1e-4f, &seed);
                    eval_light(&q, L, (f3){ a[4], a[5], a[6] }, att, &r, variant, &seed);
                }
                if (ambient) {
                    float k = q.occ * (1.0f / 255.0f);
                    r.x = fmaf(ambient[0], (float)(u2 & 0xFF) * k, r.x);
                    r.y = fmaf(ambient[1], (float)((u2 >> 8) & 0xFF) * k, r.y);
                    r.z = fmaf(ambient[2], (float)((u2 >> 16) & 0xFF) * k, r.z);
                }
            }
            float *o = hdr + i * 4;
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = 1.0f;
        }
    }
}
