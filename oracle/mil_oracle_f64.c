/*
 * Float64 C restatement of the lighting shader: the same evaluation as oracle/mil_oracle.py shade()
 * (fs_deferred_lighting.glsl:40-67, inc_brdf.glsl:2-63, + the build-defined extensions), written in C with OpenMP so
 * that whole BASELINE.json frames (3840x2160, 256 point lights) can be checked in seconds where the NumPy form
 * needs tens of minutes.  TEST INFRASTRUCTURE ONLY: loaded by tests/ (and tools/) as the checker; the product
 * (renderer_amd/, libmil.so) never does.
 *
 * PARITY: pinned exactly as far as oracle/mil_oracle.py is (the reference holds no fixture for its render path; the
 * shader text is executed by oracle/glsl_exec.py and both restatements are compared with that), and
 * tests/test_oracle_kat.py::test_f64_c_oracle_equals_numpy holds this file to the NumPy form to 1e-12 relative.
 * UNPINNED: texel format conversions, GLSL built-in precision, the extensions (see mil_oracle.py).
 *
 * Arithmetic is double on the float32-quantised inputs.  `cond` (optional) receives the conditioning bound the
 * fast-path tolerance may add (mil_oracle.py brdf_light / shade, want_cond=True).
 *
 * All citations are relative to /root/reference/.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PI_LIT 3.14159 /* renderer/shaders/inc_brdf.glsl:6,46 */

typedef struct { double x, y, z; } d3;

static inline double clampd(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
static inline double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline d3 normalize3(d3 v)
{
    double n = sqrt(dot3(v, v));
    d3 r = { v.x / n, v.y / n, v.z / n };
    return r;
}
/* column-major m[c*4+r] (float32 uniforms, promoted) */
static inline double mrow(const float *m, int r, double x, double y, double z, double w)
{
    return (double)m[0 + r] * x + (double)m[4 + r] * y + (double)m[8 + r] * z + (double)m[12 + r] * w;
}

/* inc_brdf.glsl:49-63 light() with specular :34-43, D_GGX :2-8, V_SmithGGXCorrelated :14-19, fresnel :25-32,
 * diffuse :45-47.  cond_spec (optional, 3 doubles): |specular part| x d(ln D)/d(NdotH) (mil_oracle.py brdf_light). */
static inline d3 brdf_light(d3 N, d3 L, d3 V, d3 lc, double alpha, d3 albedo, double m, double *cond_spec)
{
    d3 Hs = { L.x + V.x, L.y + V.y, L.z + V.z };
    d3 H = normalize3(Hs);
    double NdotV = clampd(dot3(N, V), 0.0001, 1.0), NdotL = clampd(dot3(N, L), 0.0, 1.0);
    double NdotH = clampd(dot3(N, H), 0.0, 1.0), LdotH = clampd(dot3(L, H), 0.0, 1.0);
    double om = 1.0 - m;
    d3 F0 = { 0.04 * om + albedo.x * m, 0.04 * om + albedo.y * m, 0.04 * om + albedo.z * m };
    double den = (1.0 - NdotH * NdotH) + (NdotH * alpha) * (NdotH * alpha);
    double k = alpha / den, D = k * k * (1.0 / PI_LIT);
    double a2 = alpha * alpha;
    double GGXV = NdotL * sqrt((NdotV - a2 * NdotV) * NdotV + a2), GGXL = NdotV * sqrt((NdotL - a2 * NdotL) * NdotL + a2);
    double Vis = 0.5 / (GGXV + GGXL);
    double f90 = clampd((F0.x + F0.y + F0.z) * (50.0 * 0.33), 0.0, 1.0);
    double w = 1.0 - LdotH, pw = w * w * w * w * w;
    d3 F = { F0.x + (f90 - F0.x) * pw, F0.y + (f90 - F0.y) * pw, F0.z + (f90 - F0.z) * pw };
    double DV = D * Vis;
    d3 spec = { DV * F.x, DV * F.y, DV * F.z };
    d3 sp = { spec.x * om + spec.x * albedo.x * m, spec.y * om + spec.y * albedo.y * m, spec.z * om + spec.z * albedo.z * m };
    d3 col = { albedo.x / PI_LIT * om + sp.x, albedo.y / PI_LIT * om + sp.y, albedo.z / PI_LIT * om + sp.z };
    d3 r = { col.x * lc.x * NdotL, col.y * lc.y * NdotL, col.z * lc.z * NdotL };
    if (cond_spec) {
        double cond = 4.0 * NdotH * NdotH * fabs(1.0 - a2) / den;
        cond_spec[0] = fabs(sp.x * lc.x * NdotL) * cond;
        cond_spec[1] = fabs(sp.y * lc.y * NdotL) * cond;
        cond_spec[2] = fabs(sp.z * lc.z * NdotL) * cond;
    }
    return r;
}

/* EXTENSION E3: 3x3 PCF for directional light 0 (mil_oracle.py pcf_shadow) */
static inline double pcf_shadow(d3 P, const float *lvp, const float *map, int S, float bias)
{
    double qx = (double)lvp[0] * P.x + (double)lvp[4] * P.y + (double)lvp[8] * P.z + (double)lvp[12];
    double qy = (double)lvp[1] * P.x + (double)lvp[5] * P.y + (double)lvp[9] * P.z + (double)lvp[13];
    double zr = (double)lvp[2] * P.x + (double)lvp[6] * P.y + (double)lvp[10] * P.z + (double)lvp[14];
    double u = qx * 0.5 + 0.5, v = qy * 0.5 + 0.5;
    if (!(u >= 0.0 && u <= 1.0 && v >= 0.0 && v <= 1.0) || zr > 1.0) return 1.0;
    long tx = (long)floor(u * S), ty = (long)floor(v * S);
    double zc = zr - (double)bias;
    int cnt = 0;
    for (int dy = -1; dy <= 1; dy++)
        for (int dx = -1; dx <= 1; dx++) {
            long xx = tx + dx, yy = ty + dy;
            xx = xx < 0 ? 0 : (xx > S - 1 ? S - 1 : xx);
            yy = yy < 0 ? 0 : (yy > S - 1 ? S - 1 : yy);
            cnt += (zc <= (double)map[(size_t)yy * S + xx]) ? 1 : 0;
        }
    return (double)cnt / 9.0;
}

/* fs_deferred_lighting.glsl:40-67 main() + extensions; same argument meaning as milo_shade (mil_oracle.c).
 * hdr: height x width x 4 doubles; cond: height x width x 3 doubles or NULL. */
void milo_shade_f64(const uint32_t *normal, const uint32_t *albedo_occ, const uint32_t *rough_metal, const float *depth,
                    int width, int height, int pitch_px, int y_offset, int full_height,
                    const float *frame_ubo, const float *dir_lights, int n_dir,
                    const float *point_lights, int n_point, const float *ambient,
                    const float *shadow_lvp, const float *shadow_map, int shadow_size, float shadow_bias,
                    double *hdr, double *cond, int nthreads)
{
    const float *invProj = frame_ubo + 32, *invView = frame_ubo + 48;
    const d3 cam = { frame_ubo[64], frame_ubo[65], frame_ubo[66] };
    const double cam_len = sqrt(dot3(cam, cam));
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            size_t i = (size_t)y * pitch_px + x;
            uint32_t u1 = normal[i], u2 = albedo_occ[i], u3 = rough_metal[i];
            d3 nrm = { (double)(u1 & 0x3FF) / 1023.0 * 2.0 - 1.0, (double)((u1 >> 10) & 0x3FF) / 1023.0 * 2.0 - 1.0,
                       (double)((u1 >> 20) & 0x3FF) / 1023.0 * 2.0 - 1.0 };                                  /* :48 */
            d3 albedo = { (double)(u2 & 0xFF) / 255.0, (double)((u2 >> 8) & 0xFF) / 255.0, (double)((u2 >> 16) & 0xFF) / 255.0 };
            double occlusion = (double)(u2 >> 24) / 255.0;
            double roughness = clampd((double)(u3 & 0xFF) / 255.0, 0.0001, 1.0);                              /* :51 */
            double metallic = (double)((u3 >> 8) & 0xFF) / 255.0;
            /* PositionFromDepth :30-38 */
            double cx = ((double)x + 0.5) / (double)width, cy = ((double)(y + y_offset) + 0.5) / (double)full_height;
            double px = cx * 2.0 - 1.0, py = cy * 2.0 - 1.0, pz = (double)depth[i];
            double vx = mrow(invProj, 0, px, py, pz, 1.0), vy = mrow(invProj, 1, px, py, pz, 1.0);
            double vz = mrow(invProj, 2, px, py, pz, 1.0), vw = mrow(invProj, 3, px, py, pz, 1.0);
            vx /= vw; vy /= vw; vz /= vw; vw = 1.0;
            d3 P = { mrow(invView, 0, vx, vy, vz, vw), mrow(invView, 1, vx, vy, vz, vw), mrow(invView, 2, vx, vy, vz, vw) };
            double alpha = roughness * roughness;                                                             /* :56 */
            d3 N = normalize3(nrm);
            d3 toCam = { cam.x - P.x, cam.y - P.y, cam.z - P.z };
            d3 V = normalize3(toCam);
            double r[3] = { 0, 0, 0 }, cb[3] = { 0, 0, 0 }, cs[3];
            for (int l = 0; l < n_dir; l++) {                                                                 /* :62-64 */
                const float *lt = dir_lights + 8 * l;
                d3 L = { lt[0], lt[1], lt[2] }, col = { lt[4], lt[5], lt[6] };
                d3 c = brdf_light(N, L, V, col, alpha, albedo, metallic, cond ? cs : NULL);
                double s = (l == 0 && shadow_lvp) ? pcf_shadow(P, shadow_lvp, shadow_map, shadow_size, shadow_bias) : 1.0;
                double k = (double)lt[3] * occlusion;
                r[0] += c.x * k * s; r[1] += c.y * k * s; r[2] += c.z * k * s;
                if (cond) { cb[0] += cs[0] * k; cb[1] += cs[1] * k; cb[2] += cs[2] * k; }
            }
            for (int l = 0; l < n_point; l++) {                                                               /* EXTENSION E1 */
                const float *lt = point_lights + 8 * l;
                d3 Lv = { (double)lt[0] - P.x, (double)lt[1] - P.y, (double)lt[2] - P.z };
                double d2 = dot3(Lv, Lv), r2 = (double)(lt[3] * lt[3]);      /* r*r as the FP32 paths form it */
                if (d2 < r2 && occlusion > 0.0) {
                    double inv = 1.0 / sqrt(fmax(d2, 1e-12));
                    d3 L = { Lv.x * inv, Lv.y * inv, Lv.z * inv }, col = { lt[4], lt[5], lt[6] };
                    double q = d2 / r2, w = clampd(1.0 - q * q, 0.0, 1.0);
                    double att = (w * w) / (d2 + 1e-4);
                    d3 c = brdf_light(N, L, V, col, alpha, albedo, metallic, cond ? cs : NULL);
                    double k = (double)lt[7] * att * occlusion;
                    double c3[3] = { c.x * k, c.y * k, c.z * k };
                    for (int ch = 0; ch < 3; ch++) r[ch] += c3[ch];
                    if (cond) {
                        /* + the hot spot next to a light: d(att)/att = 2 d dP / (d^2 + 1e-4) (mil_oracle.py shade) */
                        double scale = sqrt(dot3(P, P)) + cam_len, hot = 2.0 * sqrt(d2) * scale / (d2 + 1e-4);
                        for (int ch = 0; ch < 3; ch++) cb[ch] += cs[ch] * k + fabs(c3[ch]) * hot;
                    }
                }
            }
            if (ambient) {                                                                                    /* EXTENSION E4 */
                r[0] += (double)ambient[0] * albedo.x * occlusion;
                r[1] += (double)ambient[1] * albedo.y * occlusion;
                r[2] += (double)ambient[2] * albedo.z * occlusion;
            }
            double *o = hdr + ((size_t)y * width + x) * 4;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = 1.0;                                                /* :66 */
            if (cond) {
                double *cc = cond + ((size_t)y * width + x) * 3;
                cc[0] = cb[0]; cc[1] = cb[1]; cc[2] = cb[2];
            }
        }
    }
}
