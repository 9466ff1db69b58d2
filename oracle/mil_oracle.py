"""Float64 NumPy restatement of the reference's deferred-lighting + tonemap path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``renderer_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg do, and there only as the checker.

PARITY: the reference holds no test, fixture or golden image for its render path
 and neither its Rust host nor its GLSL can be built in
this image (no rustc, no GLSL compiler, no Vulkan ICD).  The shader arithmetic is
pinned to the reference's own source: ``oracle/glsl_exec.py`` executes the .glsl
files and ``tests/test_glsl_reference.py`` compares (the FP32 C oracle bit for bit,
this float64 one to FP32 rounding).  UNPINNED: texel format conversions, GLSL
built-in precision, blending, cgmath (driver / third-party, restated from their
published definitions) and the build-defined extensions.  Also pinned by the
hand-derived known answers in ``tests/test_oracle_kat.py``.

Every function cites the reference lines (relative to /root/reference/) it
follows.  Arithmetic is float64 on the float32-quantised inputs, so this is the
"infinitely precise" evaluation of what the GLSL text says; the C oracle is the
FP32 evaluation in GLSL operator order.

Sections marked EXTENSION are build-defined: the
reference has no point lights, shadow maps or ambient term.  They vanish
exactly when disabled.
"""
import numpy as np

PI_LITERAL = 3.14159          # renderer/shaders/inc_brdf.glsl:6,46 (literal, not pi)
MAX_LIGHTS = 100              # renderer/shaders/inc_structs.glsl:1

OUT_HDR = 0                   # R32G32B32A32_SFLOAT   (render/pbr/mod.rs:32)
OUT_LDR = 1                   # B10G11R11_UFLOAT_PACK32 (render/pbr/mod.rs:161-173)
OUT_SRGB8 = 2                 # B8G8R8A8_SRGB, direct float OETF (render/renderer.rs:78-83)
OUT_SRGB8_VIA_LDR = 3         # tonemap -> R11G11B10F -> sRGB8 (the reference chain minus FXAA)


# --------------------------------------------------------------------------
# G-buffer decode  (written by renderer/shaders/fs_deferred_geometry.glsl:47-49,
# formats render/pbr/mod.rs:150-160)
# --------------------------------------------------------------------------
def decode_a2b10g10r10(u):
    """A2B10G10R10_UNORM_PACK32: R=bits[9:0] G=[19:10] B=[29:20] A=[31:30]."""
    u = np.asarray(u, dtype=np.uint32)
    r = (u & 0x3FF).astype(np.float64) / 1023.0
    g = ((u >> 10) & 0x3FF).astype(np.float64) / 1023.0
    b = ((u >> 20) & 0x3FF).astype(np.float64) / 1023.0
    a = ((u >> 30) & 0x3).astype(np.float64) / 3.0
    return r, g, b, a


def decode_r8g8b8a8(u):
    """R8G8B8A8_UNORM: byte0=R ... byte3=A, value c/255."""
    u = np.asarray(u, dtype=np.uint32)
    return tuple(((u >> s) & 0xFF).astype(np.float64) / 255.0 for s in (0, 8, 16, 24))


def encode_a2b10g10r10(r, g, b, a=0):
    q = lambda v, m: np.clip(np.rint(np.asarray(v, np.float64) * m), 0, m).astype(np.uint32)
    return (q(r, 1023) | (q(g, 1023) << 10) | (q(b, 1023) << 20) | (q(a, 3) << 30)).astype(np.uint32)


def encode_r8g8b8a8(r, g, b, a):
    q = lambda v: np.clip(np.rint(np.asarray(v, np.float64) * 255.0), 0, 255).astype(np.uint32)
    return (q(r) | (q(g) << 8) | (q(b) << 16) | (q(a) << 24)).astype(np.uint32)


# --------------------------------------------------------------------------
# std140 uniform blocks (renderer/src/render/ubo.rs:28-63)
# --------------------------------------------------------------------------
def mat4_from_columns(f16):
    """16 floats, column-major (cgmath Matrix4 = 4 column vectors) -> M[row, col]."""
    return np.asarray(f16, dtype=np.float64).reshape(4, 4).T


def unpack_frame_ubo(ubo_f32):
    """FrameMatrixData, 272 B = 68 floats: view@0 projection@16 invProjection@32
    invView@48 cameraPosition@64 (fs_deferred_lighting.glsl:16-22)."""
    f = np.asarray(ubo_f32, dtype=np.float32).reshape(68)
    return dict(view=mat4_from_columns(f[0:16]), projection=mat4_from_columns(f[16:32]),
                inv_projection=mat4_from_columns(f[32:48]), inv_view=mat4_from_columns(f[48:64]),
                camera_position=f[64:67].astype(np.float64))


# --------------------------------------------------------------------------
# fs_deferred_lighting.glsl:30-38  PositionFromDepth
# --------------------------------------------------------------------------
def position_from_depth(depth, width, full_height, y_offset, inv_projection, inv_view):
    """depth: (h, w) float; pixel (x, y) has gl_FragCoord.xy = (x+0.5, y+y_offset+0.5)."""
    h, w = depth.shape
    xs = (np.arange(w, dtype=np.float64) + 0.5) / float(width)
    ys = (np.arange(h, dtype=np.float64) + float(y_offset) + 0.5) / float(full_height)
    cx = np.broadcast_to(xs[None, :] * 2.0 - 1.0, (h, w))
    cy = np.broadcast_to(ys[:, None] * 2.0 - 1.0, (h, w))
    clip = np.stack([cx, cy, depth.astype(np.float64), np.ones((h, w))], axis=-1)
    vs = clip @ inv_projection.T
    vs = vs / vs[..., 3:4]
    ws = vs @ inv_view.T
    return ws[..., :3]


def _normalize(v):
    return v / np.sqrt(np.sum(v * v, axis=-1, keepdims=True))


def _dot(a, b):
    return np.sum(a * b, axis=-1)


# --------------------------------------------------------------------------
# inc_brdf.glsl:49-63 light(), :34-43 specular(), :2-8 D_GGX, :14-19 V_Smith,
# :25-32 fresnel / F_Schlick, :45-47 diffuse
# --------------------------------------------------------------------------
def brdf_light(N, L, V, light_color, alpha, albedo, metallic, want_cond=False):
    """All (...,3) / (...) float64 arrays.  Returns vec3 (and, if asked, the
    per-element conditioning factor of D w.r.t. a perturbation of NdotH)."""
    H = _normalize(L + V)
    NdotV = np.clip(_dot(N, V), 0.0001, 1.0)
    NdotL = np.clip(_dot(N, L), 0.0, 1.0)
    NdotH = np.clip(_dot(N, H), 0.0, 1.0)
    LdotH = np.clip(_dot(L, H), 0.0, 1.0)
    m = metallic[..., None]
    F0 = 0.04 * (1.0 - m) + albedo * m                       # mix(vec3(0.04), albedo, metallic)
    den = (1.0 - NdotH * NdotH) + (NdotH * alpha) ** 2       # D_GGX
    k = alpha / den
    D = k * k * (1.0 / PI_LITERAL)
    a2 = alpha * alpha                                       # V_SmithGGXCorrelated
    GGXV = NdotL * np.sqrt((NdotV - a2 * NdotV) * NdotV + a2)
    GGXL = NdotV * np.sqrt((NdotL - a2 * NdotL) * NdotL + a2)
    Vis = 0.5 / (GGXV + GGXL)
    f90 = np.clip(_dot(F0, np.full(3, 50.0 * 0.33)), 0.0, 1.0)
    F = F0 + (f90[..., None] - F0) * ((1.0 - LdotH) ** 5)[..., None]
    spec = (D * Vis)[..., None] * F
    diff = albedo / PI_LITERAL
    color = diff * (1.0 - m) + (spec * (1.0 - m) + (spec * albedo) * m)
    out = (color * light_color) * NdotL[..., None]
    if want_cond:
        # d(ln D)/d(NdotH) = 4*NdotH*(1-alpha^2)/den : how many times a relative
        # error of NdotH is amplified in D.  Used by the FP32 parity tests to
        # widen the tolerance where the reference's own formula is
        # ill-conditioned in FP32 (smooth surfaces at mirror alignment).
        cond = 4.0 * NdotH * NdotH * np.abs(1.0 - a2) / den
        spec_part = (spec * (1.0 - m) + (spec * albedo) * m) * light_color * NdotL[..., None]
        return out, cond, spec_part
    return out


# --------------------------------------------------------------------------
# EXTENSION E3: 3x3 PCF on directional light 0
# --------------------------------------------------------------------------
def pcf_shadow(P, light_view_proj, shadow_map, bias):
    S = shadow_map.shape[0]
    M = light_view_proj
    q = P @ M[:3, :3].T + M[:3, 3]
    u = q[..., 0] * 0.5 + 0.5
    v = q[..., 1] * 0.5 + 0.5
    zr = q[..., 2]
    inside = (u >= 0.0) & (u <= 1.0) & (v >= 0.0) & (v <= 1.0) & ~(zr > 1.0)
    tx = np.floor(np.where(inside, u, 0.0) * S).astype(np.int64)
    ty = np.floor(np.where(inside, v, 0.0) * S).astype(np.int64)
    zc = zr - bias
    cnt = np.zeros(P.shape[:-1], dtype=np.float64)
    sm = shadow_map.astype(np.float64)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            xx = np.clip(tx + dx, 0, S - 1)
            yy = np.clip(ty + dy, 0, S - 1)
            cnt += (zc <= sm[yy, xx])
    return np.where(inside, cnt / 9.0, 1.0)


# --------------------------------------------------------------------------
# fs_deferred_lighting.glsl:40-67 main()  (+ extensions)
# --------------------------------------------------------------------------
def shade(normal_u32, albedo_occ_u32, rough_metal_u32, depth_f32, frame_ubo, dir_lights,
          width=None, full_height=None, y_offset=0, point_lights=None, ambient=None,
          shadow=None, want_cond=False):
    """Returns hdr (h, w, 4) float64; alpha = 1 (fs_deferred_lighting.glsl:66).

    dir_lights   : (n, 8) float32 rows {direction.xyz, intensity, color.xyz, pad}
                   (inc_structs.glsl:12-16; std140 stride 32)
    point_lights : (n, 8) float32 rows {position.xyz, radius, color.xyz, intensity}  EXTENSION E1
    ambient      : 3 floats or None                                                  EXTENSION E4
    shadow       : dict(light_view_proj=16 floats column-major, map=(S,S) f32, bias) EXTENSION E3
    """
    h, w = depth_f32.shape
    width = w if width is None else width
    full_height = h if full_height is None else full_height
    fr = unpack_frame_ubo(frame_ubo)
    b1 = decode_a2b10g10r10(normal_u32)
    b2 = decode_r8g8b8a8(albedo_occ_u32)
    b3 = decode_r8g8b8a8(rough_metal_u32)
    normal = np.stack(b1[:3], axis=-1) * 2.0 - 1.0           # :48
    albedo = np.stack(b2[:3], axis=-1)                       # :49
    occlusion = b2[3]                                        # :50
    roughness = np.clip(b3[0], 0.0001, 1.0)                  # :51
    metallic = b3[1]                                         # :52
    P = position_from_depth(depth_f32, width, full_height, y_offset,
                            fr['inv_projection'], fr['inv_view'])   # :53
    alpha = roughness * roughness                            # :56
    N = _normalize(normal)                                   # :58
    V = _normalize(fr['camera_position'] - P)                # :59

    result = np.zeros((h, w, 3))
    cond_bound = np.zeros((h, w, 3))
    dl = np.asarray(dir_lights, dtype=np.float32).reshape(-1, 8).astype(np.float64)
    sh = None
    if shadow is not None:
        sh = pcf_shadow(P, mat4_from_columns(shadow['light_view_proj']),
                        np.asarray(shadow['map'], np.float32), float(np.float32(shadow['bias'])))
    for i in range(dl.shape[0]):                             # :62-64
        L = np.broadcast_to(dl[i, 0:3], P.shape)
        r = brdf_light(N, L, V, dl[i, 4:7], alpha, albedo, metallic, want_cond)
        if want_cond:
            r, cond, sp = r
            cond_bound += np.abs(sp) * (cond * dl[i, 3] * occlusion)[..., None]
        c = r * dl[i, 3] * occlusion[..., None]
        if i == 0 and sh is not None:
            c = c * sh[..., None]
        result += c
    if point_lights is not None:                             # EXTENSION E1
        pl = np.asarray(point_lights, dtype=np.float32).reshape(-1, 8)
        for i in range(pl.shape[0]):
            pos = pl[i, 0:3].astype(np.float64)
            r2 = float(np.float32(pl[i, 3]) * np.float32(pl[i, 3]))   # r*r as the FP32 paths form it
            Lv = pos - P
            d2 = _dot(Lv, Lv)
            on = (d2 < r2) & (occlusion > 0.0)
            if not on.any():
                continue
            L = Lv / np.sqrt(np.maximum(d2, 1e-12))[..., None]
            qq = d2 / r2
            wdw = np.clip(1.0 - qq * qq, 0.0, 1.0)
            att = (wdw * wdw) / (d2 + 1e-4)
            r = brdf_light(N, L, V, pl[i, 4:7].astype(np.float64), alpha, albedo, metallic, want_cond)
            if want_cond:
                r, cond, sp = r
                cond_bound += np.where(on[..., None], np.abs(sp) * (cond * float(pl[i, 7]) * att * occlusion)[..., None], 0.0)
            c = r * float(pl[i, 7]) * att[..., None] * occlusion[..., None]
            if want_cond:
                # 1/(d^2 + 1e-4) next to a light: d(att)/att = 2 d dP / (d^2 + 1e-4), and the position reconstructed from an
                # FP32 depth carries an absolute error of a few ulps of |P| + |camera| (a hot spot centimetres from a light)
                scale = np.sqrt(_dot(P, P)) + np.sqrt(np.sum(fr['camera_position'] ** 2))
                cond_bound += np.where(on[..., None], np.abs(c) * (2.0 * np.sqrt(d2) * scale / (d2 + 1e-4))[..., None], 0.0)
            result += np.where(on[..., None], c, 0.0)
    if ambient is not None:                                  # EXTENSION E4
        amb = np.asarray(ambient, dtype=np.float32).astype(np.float64)
        result += (amb * albedo) * occlusion[..., None]
    hdr = np.concatenate([result, np.ones((h, w, 1))], axis=-1)
    if want_cond:
        return hdr, cond_bound
    return hdr


def pairs_in_radius(depth_f32, albedo_occ_u32, frame_ubo, point_lights, width=None,
                    full_height=None, y_offset=0):
    """Number of (pixel, point light) pairs with d^2 < r^2 and occlusion > 0:
    the pair count the FLOP roofline uses."""
    h, w = depth_f32.shape
    fr = unpack_frame_ubo(frame_ubo)
    P = position_from_depth(depth_f32, w if width is None else width,
                            h if full_height is None else full_height, y_offset,
                            fr['inv_projection'], fr['inv_view'])
    occ = (np.asarray(albedo_occ_u32, np.uint32) >> 24) > 0
    pl = np.asarray(point_lights, dtype=np.float32).reshape(-1, 8)
    n = 0
    for i in range(pl.shape[0]):
        Lv = pl[i, 0:3].astype(np.float64) - P
        r2 = float(np.float32(pl[i, 3]) * np.float32(pl[i, 3]))
        n += int(np.count_nonzero((_dot(Lv, Lv) < r2) & occ))
    return n


# --------------------------------------------------------------------------
# fs_tonemap.glsl:14-27
# --------------------------------------------------------------------------
def aces_film(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(invalid='ignore', over='ignore'):
        y = (x * (2.51 * x + 0.03)) / (x * (2.43 * x + 0.59) + 0.14)
    y = np.where(np.isnan(y), 0.0, y)     # clamp(NaN) -> max(NaN,0)=0
    return np.clip(y, 0.0, 1.0)


# --------------------------------------------------------------------------
# B10G11R11_UFLOAT_PACK32 (render/pbr/mod.rs:161-173): R=bits[10:0] 5e6m,
# G=[21:11] 5e6m, B=[31:22] 5e5m; bias 15; round-to-nearest-even.
# --------------------------------------------------------------------------
def _encode_ufloat(x, mbits):
    x = np.clip(np.asarray(x, dtype=np.float64), 0.0, 65024.0 if mbits == 6 else 64512.0)
    out = np.zeros(x.shape, dtype=np.uint32)
    with np.errstate(divide='ignore'):
        e = np.floor(np.log2(np.where(x > 0, x, 1.0))).astype(np.int64)
    # guard log2 rounding at exact powers of two
    e = np.where(np.ldexp(1.0, e) > x, e - 1, e)
    e = np.where(np.ldexp(1.0, e + 1) <= x, e + 1, e)
    normal = x >= 2.0 ** -14
    scale = float(1 << mbits)
    mant = np.rint(np.where(normal, x / np.ldexp(1.0, e) - 1.0, 0.0) * scale).astype(np.int64)
    code_n = ((e + 15) << mbits) + mant          # mant == 2^mbits carries into the exponent
    code_d = np.rint(x * 2.0 ** 14 * scale).astype(np.int64)
    out = np.where(normal, code_n, code_d).astype(np.uint32)
    return out


def _decode_ufloat(code, mbits):
    code = np.asarray(code, dtype=np.int64)
    e = code >> mbits
    m = (code & ((1 << mbits) - 1)).astype(np.float64)
    scale = float(1 << mbits)
    return np.where(e == 0, m / scale * 2.0 ** -14, (1.0 + m / scale) * np.ldexp(1.0, e - 15))


def encode_r11g11b10f(rgb):
    rgb = np.asarray(rgb, dtype=np.float64)
    r = _encode_ufloat(rgb[..., 0], 6)
    g = _encode_ufloat(rgb[..., 1], 6)
    b = _encode_ufloat(rgb[..., 2], 5)
    return (r | (g << 11) | (b << 22)).astype(np.uint32)


def decode_r11g11b10f(u):
    u = np.asarray(u, dtype=np.uint32).astype(np.int64)
    return np.stack([_decode_ufloat(u & 0x7FF, 6), _decode_ufloat((u >> 11) & 0x7FF, 6),
                     _decode_ufloat((u >> 22) & 0x3FF, 5)], axis=-1)


# --------------------------------------------------------------------------
# B8G8R8A8_SRGB swapchain encode (render/renderer.rs:78-83,108)
# --------------------------------------------------------------------------
def srgb_oetf(c):
    c = np.clip(np.asarray(c, dtype=np.float64), 0.0, 1.0)
    return np.where(c <= 0.0031308, 12.92 * c, 1.055 * np.power(c, 1.0 / 2.4) - 0.055)


def encode_b8g8r8a8_srgb(rgb):
    q = np.floor(srgb_oetf(rgb) * 255.0 + 0.5).astype(np.uint32)
    return (q[..., 2] | (q[..., 1] << 8) | (q[..., 0] << 16) | np.uint32(0xFF000000)).astype(np.uint32)


def srgb8_luts():
    """sRGB8 code for every R11G11B10F channel code <= 1.0 (961 / 481 entries)."""
    l11 = np.floor(srgb_oetf(_decode_ufloat(np.arange(0x3C1), 6)) * 255.0 + 0.5).astype(np.uint8)
    l10 = np.floor(srgb_oetf(_decode_ufloat(np.arange(0x1E1), 5)) * 255.0 + 0.5).astype(np.uint8)
    return l11, l10


def tonemap(hdr, fmt):
    """fs_tonemap.glsl:23-27 followed by the attachment encode named by fmt."""
    ldr = aces_film(np.asarray(hdr)[..., :3])
    if fmt == OUT_LDR:
        return encode_r11g11b10f(ldr)
    if fmt == OUT_SRGB8:
        return encode_b8g8r8a8_srgb(ldr)
    if fmt == OUT_SRGB8_VIA_LDR:
        return encode_b8g8r8a8_srgb(decode_r11g11b10f(encode_r11g11b10f(ldr)))
    raise ValueError(fmt)


# --------------------------------------------------------------------------
# NEXT ROW f-1: FXAA (renderer/shaders/fs_fxaa.glsl:22-91, drawn at render/mod.rs:312-329;
# sampler NEAREST + CLAMP_TO_EDGE, render/fxaa/mod.rs:76-88) from the ldr attachment to the
# B8G8R8A8_SRGB swapchain image (render/renderer.rs:78-83).
# --------------------------------------------------------------------------
def fxaa(ldr_u32, row_begin=0, row_count=None, return_float=False):
    """ldr_u32: (h, w) B10G11R11_UFLOAT codes of the WHOLE frame.  Returns the packed swapchain
    texels of rows [row_begin, row_begin + row_count), or the float rgb when return_float."""
    ldr_u32 = np.asarray(ldr_u32, dtype=np.uint32)
    h, w = ldr_u32.shape
    row_count = h - row_begin if row_count is None else row_count
    img = decode_r11g11b10f(ldr_u32)                                   # (h, w, 3) float64
    f32 = lambda v: np.float64(np.float32(v))
    inv = np.array([f32(1.0 / f32(w)), f32(1.0 / f32(h))])             # 1.0 / resolution, an FP32 value in the shader
    ys, xs = np.meshgrid(np.arange(row_begin, row_begin + row_count), np.arange(w), indexing='ij')
    frag = np.stack([xs + 0.5, ys + 0.5], -1)                          # gl_FragCoord.xy

    def tex(uv):                                                       # nearest, clamp to edge
        ix = np.clip(np.floor(uv[..., 0] * w), 0, w - 1).astype(np.int64)
        iy = np.clip(np.floor(uv[..., 1] * h), 0, h - 1).astype(np.int64)
        return img[iy, ix]

    luma_w = np.array([f32(0.299), f32(0.587), f32(0.114)])
    luma = lambda c: c @ luma_w
    nw, ne = tex((frag + [-1.0, -1.0]) * inv), tex((frag + [1.0, -1.0]) * inv)       # :24-28, :37-41
    sw, se = tex((frag + [-1.0, 1.0]) * inv), tex((frag + [1.0, 1.0]) * inv)
    um = frag * inv
    m = tex(um)
    l_nw, l_ne, l_sw, l_se, l_m = luma(nw), luma(ne), luma(sw), luma(se), luma(m)
    lmin = np.minimum(l_m, np.minimum(np.minimum(l_nw, l_ne), np.minimum(l_sw, l_se)))  # :49-50
    lmax = np.maximum(l_m, np.maximum(np.maximum(l_nw, l_ne), np.maximum(l_sw, l_se)))
    d = np.stack([-((l_nw + l_ne) - (l_sw + l_se)), (l_nw + l_sw) - (l_ne + l_se)], -1)  # :53-54
    reduce = np.maximum((l_nw + l_ne + l_sw + l_se) * (0.25 * (1.0 / 8.0)), 1.0 / 128.0)  # :56-57
    rcpmin = 1.0 / (np.minimum(np.abs(d[..., 0]), np.abs(d[..., 1])) + reduce)           # :59
    d = np.clip(d * rcpmin[..., None], -8.0, 8.0) * inv                                   # :60-62
    k1, k2 = f32(1.0 / 3.0 - 0.5), f32(2.0 / 3.0 - 0.5)               # glslang folds in double, emits floats
    a = 0.5 * (tex(um + d * k1) + tex(um + d * k2))                   # :64
    b = a * 0.5 + 0.25 * (tex(um + d * -0.5) + tex(um + d * 0.5))     # :65
    l_b = luma(b)
    rgb = np.where(((l_b < lmin) | (l_b > lmax))[..., None], a, b)    # :67-71
    return rgb if return_float else encode_b8g8r8a8_srgb(rgb)


# --------------------------------------------------------------------------
# NEXT ROW f-4: Hosek-Wilkie sky subpass (renderer/shaders/sky_hosek_frag.glsl:28-60, sky_hosek_vert.glsl,
# draw render/mod.rs:221-226, pipeline render/hosek/mod.rs:46-76).  See oracle/mil_oracle.c for why the
# pass covers exactly the texels still at the clear depth and why the view direction is the pixel's ray.
# params: (11, 4) or flat 44 floats = A B C D E F G H I Z sun_direction at 16-byte stride (hosek/mod.rs:222-256).
# --------------------------------------------------------------------------
SKY_DEPTH_PASS = float(np.float32(0.99999994))


def sky_hosek(depth_f32, frame_ubo, params, hdr, y_offset=0, full_height=None):
    depth = np.asarray(depth_f32, dtype=np.float32)
    h, w = depth.shape
    full_height = h if full_height is None else full_height
    f = unpack_frame_ubo(frame_ubo)
    prm = np.asarray(params, np.float64).reshape(11, 4)[:, :3]
    A, B, C, D, E, F, G, H, I, Z, sun = prm
    P = position_from_depth(depth.astype(np.float64), w, full_height, y_offset, f['inv_projection'], f['inv_view'])
    vd = _normalize(P - f['camera_position'])
    cg = np.maximum(0.001, vd @ sun)[..., None]
    ct = np.maximum(0.001, vd[..., 1])[..., None]
    gamma = np.arccos(cg)
    chi = (1.0 + cg * cg) / np.power(1.0 + H * H - 2.0 * H * cg, 1.5)
    rad = (1.0 + A * np.exp(B / (ct + 0.01))) * (C + D * np.exp(E * gamma) + F * cg * cg + G * chi + I * np.sqrt(ct))
    out = np.array(hdr, dtype=np.float64, copy=True)
    sky = depth >= np.float32(SKY_DEPTH_PASS)
    out[sky, :3] = (rad * Z * 0.05)[sky]
    out[sky, 3] = 1.0
    return out


# --------------------------------------------------------------------------
# NEXT ROW f-2: transparency resolve (renderer/shaders/fs_mcguire13_resolve.glsl:8-13; blend state
# render/mcguire13/mod.rs:108-120: src * (1 - srcAlpha) + dst * srcAlpha for colour and alpha; draw
# render/mod.rs:282-294).  accum RGBA16F, reveal R16F (mcguire13/mod.rs:22-23).
# --------------------------------------------------------------------------
def transparency_resolve(accum_f16, reveal_f16, hdr):
    accum = np.asarray(accum_f16, np.float16).astype(np.float64)
    r = np.asarray(reveal_f16, np.float16).astype(np.float64)
    hdr = np.asarray(hdr, np.float64)
    src = accum[..., :3] / np.minimum(5e4, np.maximum(np.float64(np.float32(1e-4)), accum[..., 3]))[..., None]
    out = np.empty_like(hdr)
    out[..., :3] = src * (1.0 - r)[..., None] + hdr[..., :3] * r[..., None]
    out[..., 3] = r * (1.0 - r) + hdr[..., 3] * r
    return out


# --------------------------------------------------------------------------
# NEXT ROW f-2, second half: the fragment shader of the transparency accumulate draw over a fragment list
# (renderer/shaders/fs_mcguire13_accumulation.glsl:90-127; draw render/mod.rs:229-281).  Planes as in
# oracle/mil_oracle.c milo_transparency_accumulate: pos_z = (in_wsPosition, gl_FragCoord.z), nrm_op = (in_normal,
# opacity), alb_rough = (albedo, roughness), metallic; the material factor x texel products (:91-96) are the caller's.
# --------------------------------------------------------------------------
def transparency_accumulate(pos_z, nrm_op, alb_rough, metallic, camera_position, dir_lights, want_cond=False):
    """want_cond: also return sum |specular term| x the conditioning of D_GGX w.r.t. NdotH, scaled like accum (see brdf_light)."""
    pos_z, nrm_op, alb_rough = (np.asarray(a, np.float64).reshape(-1, 4) for a in (pos_z, nrm_op, alb_rough))
    metallic = np.asarray(metallic, np.float64).reshape(-1)
    cam = np.asarray(camera_position, np.float64).reshape(3)
    albedo, opacity = alb_rough[:, :3], nrm_op[:, 3]
    alpha = alb_rough[:, 3] ** 2                                     # :104 (no clamp, unlike fs_deferred_lighting.glsl:51)
    N, V = _normalize(nrm_op[:, :3]), _normalize(cam - pos_z[:, :3])
    m = metallic[:, None]
    lighting, cond_bound = np.zeros_like(albedo), np.zeros_like(albedo)
    for l in np.asarray(dir_lights, np.float64).reshape(-1, 8):      # :110-119
        L = np.broadcast_to(l[0:3], N.shape)
        H = _normalize(L + V)
        NdotV, NdotL = np.clip(_dot(N, V), 0.0001, 1.0), np.clip(_dot(N, L), 0.0, 1.0)
        NdotH, LdotH = np.clip(_dot(N, H), 0.0, 1.0), np.clip(_dot(L, H), 0.0, 1.0)
        F0 = 0.04 * (1.0 - m) + albedo * m
        k = alpha / ((1.0 - NdotH * NdotH) + (NdotH * alpha) ** 2)
        D = k * k * (1.0 / PI_LITERAL)
        a2 = alpha * alpha
        Vis = 0.5 / (NdotL * np.sqrt((NdotV - a2 * NdotV) * NdotV + a2) + NdotV * np.sqrt((NdotL - a2 * NdotL) * NdotL + a2))
        f90 = np.clip(_dot(F0, np.full(3, 50.0 * 0.33)), 0.0, 1.0)
        F = F0 + (f90[:, None] - F0) * ((1.0 - LdotH) ** 5)[:, None]
        spec = (D * Vis)[:, None] * F * l[4:7] * NdotL[:, None]
        lighting = lighting + (albedo / PI_LITERAL + spec)
        cond_bound += np.abs(spec) * (4.0 * NdotH * NdotH * np.abs(1.0 - a2) / ((1.0 - NdotH * NdotH) + (NdotH * alpha) ** 2))[:, None]
    f = 1.0 - pos_z[:, 3]
    w = opacity * np.maximum(0.01, 3000.0 * f * f * f)               # w10, :84-87
    accum = np.concatenate([lighting * opacity[:, None], opacity[:, None]], 1) * w[:, None]
    if want_cond:
        return accum, opacity.copy(), np.concatenate([cond_bound * (opacity * w)[:, None], np.zeros((len(w), 1))], 1)
    return accum, opacity.copy()
