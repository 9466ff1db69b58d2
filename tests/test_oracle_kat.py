"""Pins the CPU oracles (oracle/mil_oracle.py float64, oracle/mil_oracle.c FP32) with the
hand-derived known answers of SURVEY.md 8(c) — the reference itself holds no golden
vector for this path (tests/test_glsl_reference.py pins the oracles to the executed shader text) — and checks the two
restatements agree."""
import os

import numpy as np
import pytest

from oracle import mil_oracle as mo
from renderer_amd import synth, ubo

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def identity_frame(cam=(0.0, 0.0, 0.0)):
    eye = np.eye(4, dtype=np.float32)
    return ubo.frame_matrix_data(eye, eye, cam)


def one_pixel(normal=(0, 1, 0), albedo=(1, 1, 1), occ=1.0, rough=1.0, metal=0.0, depth=0.5):
    n = np.asarray(normal, np.float64) * 0.5 + 0.5
    return (mo.encode_a2b10g10r10(n[0], n[1], n[2]).reshape(1, 1),
            mo.encode_r8g8b8a8(albedo[0], albedo[1], albedo[2], occ).reshape(1, 1),
            mo.encode_r8g8b8a8(rough, metal, 0, 0).reshape(1, 1),
            np.full((1, 1), depth, np.float32))


def kat_setup():
    """Pixel at world (0,0,0.5) (identity matrices, 1x1 frame), camera on +y above it: V = N = L = (0,1,0)."""
    frame = identity_frame(cam=(0.0, 5.0, 0.5))
    lights = ubo.directional_lights([((0, 1, 0), 1.0, (1, 1, 1))])
    return frame, lights


def both(coracle, planes, frame, lights, **kw):
    a = mo.shade(*planes, frame, lights, **kw)
    b = coracle.shade(*planes, frame, lights, **kw)
    return a[0, 0, :3], b[0, 0, :3].astype(np.float64)


def test_kat1_dielectric(coracle):
    # D = 1/3.14159, V = 0.25, F = 0.04 (f90 = clamp(16.5*0.12) = 1 but (1-LdotH)^5 = 0)
    # spec = 0.31831015*0.25*0.04 ; diffuse = 1/3.14159 ; the 10-bit normal (0.5,1,0.5)->(512,1023,512)
    # is not exactly (0,1,0): N.y = 0.99999905, so allow 1e-5.
    frame, lights = kat_setup()
    a, b = both(coracle, one_pixel(), frame, lights)
    np.testing.assert_allclose(a, 0.32149325, atol=2e-5)
    np.testing.assert_allclose(b, 0.32149325, atol=2e-5)


def test_kat2_metal(coracle):
    frame, lights = kat_setup()
    px = one_pixel(albedo=(1.0, 0.5, 0.25), metal=1.0)
    a, b = both(coracle, px, frame, lights)
    alb = np.array([255, 128, 64]) / 255.0            # 8-bit albedo actually stored
    exp = 0.07957754 * alb * alb                      # D*V*F0 * albedo (mix(spec, spec*albedo, 1))
    np.testing.assert_allclose(a, exp, atol=2e-5)
    np.testing.assert_allclose(b, exp, atol=2e-5)


def test_kat3_backfacing_is_exactly_zero(coracle):
    frame, _ = kat_setup()
    lights = ubo.directional_lights([((0.6, -0.8, 0), 3.0, (1, 1, 1))])   # N.L < 0 (L = -V exactly is the NaN case, inc_brdf.glsl:50)
    a, b = both(coracle, one_pixel(), frame, lights)
    assert np.all(a == 0.0) and np.all(b == 0.0)


def test_kat4_position_from_depth_identity():
    W, H = 8, 4
    eye = np.eye(4)
    depth = np.random.RandomState(0).rand(H, W).astype(np.float32)
    P = mo.position_from_depth(depth, W, H, 0, eye, eye)
    xs = 2 * (np.arange(W) + 0.5) / W - 1
    ys = 2 * (np.arange(H) + 0.5) / H - 1
    np.testing.assert_allclose(P[..., 0], np.broadcast_to(xs, (H, W)), atol=1e-15)
    np.testing.assert_allclose(P[..., 1], np.broadcast_to(ys[:, None], (H, W)), atol=1e-15)
    np.testing.assert_allclose(P[..., 2], depth, atol=0)
    # strips use global coordinates
    P2 = mo.position_from_depth(depth[2:], W, H, 2, eye, eye)
    np.testing.assert_array_equal(P2, P[2:])


def test_kat5_aces(coracle):
    np.testing.assert_allclose(mo.aces_film(np.array([0.0, 1.0, 10.0])), [0.0, 2.54 / 3.16, 1.0], atol=1e-12)
    hdr = np.array([[[0.0, 1.0, 10.0, 1.0]]], np.float32)
    code = coracle.tonemap(hdr, mo.OUT_LDR)[0, 0]
    rgb = mo.decode_r11g11b10f(code)
    np.testing.assert_allclose(rgb, [0.0, 0.80379747, 1.0], atol=1 / 128)


def test_kat6_cleared_pixel_is_zero(coracle):
    cam = synth.sweep_camera(16, 9)
    frame = cam.frame_matrix_data()
    z = np.zeros((9, 16), np.uint32)
    planes = (z, z, z, np.ones((9, 16), np.float32))
    lights = ubo.default_directional_lights()
    a = mo.shade(*planes, frame, lights)
    b = coracle.shade(*planes, frame, lights)
    assert np.all(a[..., :3] == 0.0) and np.all(b[..., :3] == 0.0)
    assert np.all(a[..., 3] == 1.0) and np.all(b[..., 3] == 1.0)
    assert np.all(np.isfinite(b))


def test_kat7_layouts():
    assert ubo.FRAME_UBO_BYTES == 272 and ubo.frame_matrix_data(np.eye(4), np.eye(4), (0, 0, 0)).nbytes == 272
    assert ubo.default_directional_lights().strides[0] == 32
    assert ubo.push_constants(1920, 1080, 2).nbytes == 12
    assert int(mo.encode_a2b10g10r10(1, 0, 0, 0)) == 0x000003FF
    assert int(mo.encode_a2b10g10r10(0, 1, 0, 0)) == 0x3FF << 10
    assert int(mo.encode_r8g8b8a8(1, 0, 0, 0)) == 0xFF and int(mo.encode_r8g8b8a8(0, 0, 0, 1)) == 0xFF000000
    f = synth.sweep_camera(16, 9).frame_matrix_data()
    view, proj = mo.mat4_from_columns(f[0:16]), mo.mat4_from_columns(f[16:32])
    np.testing.assert_allclose(view @ mo.mat4_from_columns(f[48:64]), np.eye(4), atol=1e-5)
    np.testing.assert_allclose(proj @ mo.mat4_from_columns(f[32:48]), np.eye(4), atol=1e-5)
    assert proj[3, 2] == -1.0                          # cgmath perspective: m[2][3] = -1 (column-major)
    # default reference camera (main.rs:70-78): fov 90 deg => f = 1
    p = ubo.PerspectiveCamera(aspect_ratio=16 / 9).projection_matrix()
    np.testing.assert_allclose([p[1, 1], p[0, 0]], [1.0, 9 / 16], rtol=1e-6)
    dl = ubo.default_directional_lights()
    np.testing.assert_allclose(np.linalg.norm(dl[:, 0:3], axis=1), 1.0, atol=1e-6)


def test_r11g11b10f_codec(coracle):
    assert coracle.pack_r11g11b10f(1.0, 1.0, 1.0) == (0x3C0 | (0x3C0 << 11) | (0x1E0 << 22))
    assert coracle.pack_r11g11b10f(0.0, 0.0, 0.0) == 0
    assert coracle.pack_r11g11b10f(0.5, 0.25, 0.125) == ((14 << 6) | ((13 << 6) << 11) | ((12 << 5) << 22))
    rng = np.random.RandomState(1)
    x = np.concatenate([rng.rand(4000, 3), rng.rand(2000, 3) * 1e-4, rng.rand(2000, 3) * 2e-6]).astype(np.float32)
    a = mo.encode_r11g11b10f(x)
    b = np.array([coracle.pack_r11g11b10f(*row) for row in x], np.uint32)
    np.testing.assert_array_equal(a, b)
    # decode(encode(x)) is the nearest representable value: error <= half a step
    dec = mo.decode_r11g11b10f(a)
    step = np.maximum(2.0 ** np.floor(np.log2(np.maximum(x, 2.0 ** -14))), 2.0 ** -14) / np.array([64, 64, 32])
    assert np.all(np.abs(dec - x) <= step / 2 + 1e-12)
    # exact ties round to even
    assert coracle.pack_r11g11b10f(1.0 - 2 ** -8, 0, 0) == 0x3C0          # tie 0x3BF.5 -> even 0x3C0
    assert coracle.pack_r11g11b10f(0.5 + 2 ** -8, 0, 0) == (14 << 6)      # tie 0x380.5 -> even 0x380


def test_srgb_luts_and_formats(coracle):
    l11, l10 = mo.srgb8_luts()
    assert l11[0] == 0 and l11[0x3C0] == 255 and l10[0x1E0] == 255 and len(l11) == 961 and len(l10) == 481
    assert np.all(np.diff(l11.astype(int)) >= 0)
    rng = np.random.RandomState(2)
    hdr = np.concatenate([rng.rand(32, 32, 4) * 4, rng.rand(32, 32, 4) * 0.05], 0).astype(np.float32)
    for fmt in (mo.OUT_LDR, mo.OUT_SRGB8_VIA_LDR):
        a, b = mo.tonemap(hdr, fmt), coracle.tonemap(hdr, fmt)
        # float64 vs FP32 ACES can straddle a rounding boundary: at most one code step, rarely
        d = a != b
        assert d.mean() < 2e-3
    a, b = mo.tonemap(hdr, mo.OUT_SRGB8), coracle.tonemap(hdr, mo.OUT_SRGB8)
    ch = lambda u: np.stack([(u >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(int)
    assert np.abs(ch(a) - ch(b)).max() <= 1 and np.all(ch(b)[..., 3] == 255)


def _scene(w=96, h=54, n_point=24, shadow=128):
    cam = synth.sweep_camera(w, h)
    gb = synth.to_numpy(synth.scene_sweep(w, h, cam))
    frame = cam.frame_matrix_data()
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(n_point, 11)
    sh = synth.shadow_setup(dl[0, 0:3], shadow)
    sh = dict(sh, map=sh['map'].numpy())
    return gb, frame, dl, pl, sh


def test_two_restatements_agree(coracle):
    """float64 NumPy vs FP32 C on Scene A with every extension on.  Tolerance: the HDR bar
    of BASELINE.md plus the conditioning allowance where the reference's own D_GGX is
    ill-conditioned in FP32 (smooth surfaces at mirror alignment), and <=0.1% of pixels may
    differ by a PCF tap flip."""
    gb, frame, dl, pl, sh = _scene()
    planes = (gb['normal'], gb['albedo_occ'], gb['rough_metal'], gb['depth'])
    amb = (0.03, 0.03, 0.03)
    a, cond = mo.shade(*planes, frame, dl, point_lights=pl, ambient=amb, shadow=sh, want_cond=True)
    b, pairs = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=amb, shadow=sh, return_pairs=True)
    tol = 1e-3 + 1e-4 * np.abs(a[..., :3]) + 8 * 2.0 ** -24 * cond
    bad = np.abs(b[..., :3] - a[..., :3]) > tol
    assert bad.any(-1).mean() < 1e-3, bad.sum()
    assert abs(pairs - mo.pairs_in_radius(gb['depth'], gb['albedo_occ'], frame, pl)) <= 2
    # disabled extensions vanish exactly
    c = coracle.shade(*planes, frame, dl)
    d = coracle.shade(*planes, frame, dl, point_lights=pl[:0], ambient=None, shadow=None)
    np.testing.assert_array_equal(c, d)


def test_f64_c_oracle_equals_numpy(coracle):
    """oracle/mil_oracle_f64.c (what the full-size GPU tests check the fast path against) is the NumPy float64 oracle
    to rounding: result and conditioning bound, Scene A and noise, every extension on, strips included."""
    for kind, (w, h) in (('sweep', (96, 54)), ('noise', (64, 36))):
        cam = synth.sweep_camera(w, h)
        gb = synth.to_numpy(synth.scene_sweep(w, h, cam) if kind == 'sweep' else synth.scene_noise(w, h, 5))
        frame, dl, pl = cam.frame_matrix_data(), ubo.default_directional_lights(), synth.point_light_field(40, 3)
        sh = synth.shadow_setup(dl[0, 0:3], 128)
        sh = dict(sh, map=sh['map'].numpy())
        planes = (gb['normal'], gb['albedo_occ'], gb['rough_metal'], gb['depth'])
        kw = dict(point_lights=pl, ambient=(0.03, 0.03, 0.03), shadow=sh, want_cond=True)
        a, ca = mo.shade(*planes, frame, dl, **kw)
        b, cb = coracle.shade_f64(*planes, frame, dl, **kw)
        assert np.array_equal(np.isfinite(a), np.isfinite(b))
        fin = np.isfinite(a)
        assert np.all(np.abs(a - b)[fin] <= 1e-12 * np.abs(a[fin]) + 1e-300)
        finc = np.isfinite(ca)
        assert np.all(np.abs(ca - cb)[finc] <= 1e-11 * np.abs(ca[finc]) + 1e-300)
        y0, y1 = 16, 32
        part = coracle.shade_f64(*[p[y0:y1] for p in planes], frame, dl, point_lights=pl, shadow=sh, y_offset=y0, full_height=h)
        full = coracle.shade_f64(*planes, frame, dl, point_lights=pl, shadow=sh)
        np.testing.assert_array_equal(part, full[y0:y1])
        assert coracle.shade_f64(*planes, frame, dl).shape == (h, w, 4)


def test_strip_equals_monolithic(coracle):
    gb, frame, dl, pl, sh = _scene(64, 48, 8, 64)
    planes = (gb['normal'], gb['albedo_occ'], gb['rough_metal'], gb['depth'])
    full = coracle.shade(*planes, frame, dl, point_lights=pl, shadow=sh)
    for y0, y1 in ((0, 16), (16, 48)):
        part = coracle.shade(*[p[y0:y1] for p in planes], frame, dl, point_lights=pl, shadow=sh,
                             y_offset=y0, full_height=48)
        np.testing.assert_array_equal(part, full[y0:y1])


def test_metamorphic_light_split_and_permutation(coracle):
    gb, frame, dl, pl, sh = _scene(48, 27, 0, 32)
    planes = (gb['normal'], gb['albedo_occ'], gb['rough_metal'], gb['depth'])
    both_ = mo.shade(*planes, frame, dl)[..., :3]
    one = mo.shade(*planes, frame, dl[:1])[..., :3] + mo.shade(*planes, frame, dl[1:])[..., :3]
    np.testing.assert_allclose(both_, one, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(both_, mo.shade(*planes, frame, dl[::-1])[..., :3], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize('name', ['sweep_96x54', 'noise_64x36'])
def test_golden_fixtures(coracle, name):
    """The committed fixtures (tests/golden/make_golden.py) still come out of both oracles."""
    z = np.load(os.path.join(GOLDEN, name + '.npz'))
    planes = (z['normal'], z['albedo_occ'], z['rough_metal'], z['depth'])
    sh = dict(light_view_proj=z['shadow_lvp'], map=z['shadow_map'], bias=float(z['shadow_bias']))
    kw = dict(point_lights=z['point_lights'], ambient=z['ambient'], shadow=sh)
    b = coracle.shade(*planes, z['frame'], z['dir_lights'], **kw)
    np.testing.assert_array_equal(b.view(np.uint32), z['hdr_c32'].view(np.uint32))
    a = mo.shade(*planes, z['frame'], z['dir_lights'], **kw)
    np.testing.assert_allclose(a, z['hdr_f64'], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(coracle.tonemap(z['hdr_c32'], 1), z['ldr_c32'])
    np.testing.assert_array_equal(coracle.tonemap(z['hdr_c32'], 3), z['srgb8_via_ldr_c32'])
    b0 = coracle.shade(*planes, z['frame'], z['dir_lights'])
    np.testing.assert_array_equal(b0.view(np.uint32), z['hdr_dir_only_c32'].view(np.uint32))


# ------------------------------------------------------------------ next row f-1: FXAA (fs_fxaa.glsl:22-91)
def _gray(v):
    return [v, v, v]


def _ch(u):
    return np.stack([(u >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(int)


def test_kat8_fxaa_hand_derived(coracle):
    """3x3 gray frame, centre pixel M = 0 with NW .5, NE .25, SW = SE = 0, W .25, E .5 (all exact in B10G11R11):
    dir = (-(NW+NE), NW-NE) = (-.75, .25); dirReduce = max(.75 * .03125, 1/128) = .0234375;
    rcpDirMin = 1 / (.25 + .0234375) = 3.657143; dir * rcpDirMin = (-2.742857, .914286), inside +-8.
    Taps (pixel units from (1.5, 1.5)): *(-1/6) -> (1.957, 1.348) = M, *(1/6) -> (1.043, 1.652) = M,
    *(-.5) -> (2.871, 1.043) = E, *(.5) -> (0.129, 1.957) = W.  rgbA = 0, rgbB = 0 * .5 + .25 * (.5 + .25) = .1875,
    lumaB = .1875 in [0, .5] -> rgbB.  sRGB: 1.055 * .1875^(1/2.4) - .055 = .47021 -> round(119.90) = 120 = 0x78."""
    img = np.array([[_gray(.5), _gray(.125), _gray(.25)], [_gray(.25), _gray(0), _gray(.5)], [_gray(0), _gray(.125), _gray(0)]])
    ldr = mo.encode_r11g11b10f(img)
    assert mo.fxaa(ldr)[1, 1] == 0xFF787878
    assert coracle.fxaa(ldr)[1, 1] == 0xFF787878
    np.testing.assert_allclose(mo.fxaa(ldr, return_float=True)[1, 1], 0.1875, rtol=0, atol=1e-12)


def test_fxaa_flat_and_axis_aligned_edges_are_untouched(coracle):
    flat = mo.encode_r11g11b10f(np.full((9, 11, 3), 0.25))
    plain = lambda ldr: mo.encode_b8g8r8a8_srgb(mo.decode_r11g11b10f(ldr))
    for f in (mo.fxaa, coracle.fxaa):
        np.testing.assert_array_equal(f(flat), plain(flat))
    img = np.zeros((12, 16, 3))
    img[:, 8:] = 0.75                                   # vertical edge: dir = (0, -8 px): taps stay in the column
    v = mo.encode_r11g11b10f(img)
    h = mo.encode_r11g11b10f(img.transpose(1, 0, 2).copy())
    for f in (mo.fxaa, coracle.fxaa):
        np.testing.assert_array_equal(f(v), plain(v))
        np.testing.assert_array_equal(f(h), plain(h))


def fxaa_test_image(h, w, seed=0):
    """Tonemapped-looking frame with diagonal edges, thin lines, noise and dark (denormal) values."""
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.zeros((h, w, 3))
    img[..., 0] = (xx * 0.37 + yy * 0.61 > 0.45 * (w + h) / 2) * 0.8 + 0.05
    img[..., 1] = ((xx + 2 * yy) % 23 < 9) * 0.6 + 0.02 * rng.rand(h, w)
    img[..., 2] = rng.rand(h, w) ** 3
    img[: h // 8] *= 1e-4                                # B10G11R11 denormals
    img[(xx - w // 2) ** 2 + (yy - h // 2) ** 2 < (h // 4) ** 2] = (0.9, 0.85, 0.1)
    return mo.encode_r11g11b10f(img)


def test_fxaa_two_restatements_agree_and_strips(coracle):
    ldr = fxaa_test_image(70, 101, 3)
    a, b = mo.fxaa(ldr), coracle.fxaa(ldr)
    d = np.abs(_ch(a) - _ch(b)).max(-1)
    assert (d > 1).mean() <= 2e-3 and (d > 0).mean() <= 2e-2      # FP64 vs FP32: rare nearest-texel flips, else <= 1 LSB
    assert np.all(_ch(b)[..., 3] == 255)
    for f in (mo.fxaa, coracle.fxaa):                              # rows [y0, y1) of the whole frame == a strip call
        full = f(ldr)
        np.testing.assert_array_equal(f(ldr, 16, 32), full[16:48])
        np.testing.assert_array_equal(f(ldr, 69, 1), full[69:])
    # every output channel lies within the range of the 9x9 neighbourhood the taps can reach (blend of texels)
    lin = mo.fxaa(ldr, return_float=True)
    img = mo.decode_r11g11b10f(ldr)
    pad = np.pad(img, ((4, 4), (4, 4), (0, 0)), mode='edge')
    win = np.lib.stride_tricks.sliding_window_view(pad, (9, 9), axis=(0, 1))
    assert np.all(lin <= win.max((-1, -2)) + 1e-12) and np.all(lin >= win.min((-1, -2)) - 1e-12)


def test_frame_chain_golden_fixture(coracle):
    """The whole-chain fixture still comes out of the C oracle, and the float64 oracle agrees with it."""
    z, g = np.load(os.path.join(GOLDEN, 'frame_chain_96x54.npz')), np.load(os.path.join(GOLDEN, 'sweep_96x54.npz'))
    hdr = coracle.transparency_resolve(z['accum'], z['reveal'], coracle.sky_hosek(g['depth'], g['frame'], z['sky_params'], g['hdr_c32'].copy()))
    np.testing.assert_array_equal(hdr.view(np.uint32), z['hdr_c32'].view(np.uint32))
    np.testing.assert_array_equal(coracle.tonemap(hdr, 1), z['ldr_c32'])
    np.testing.assert_array_equal(coracle.fxaa(z['ldr_c32']), z['swapchain_c32'])
    h64 = mo.transparency_resolve(z['accum'], z['reveal'], mo.sky_hosek(g['depth'], g['frame'], z['sky_params'], g['hdr_f64']))
    fin = np.isfinite(h64).all(-1)
    assert (np.abs(h64 - hdr)[fin] > 1e-4 + 1e-4 * np.abs(h64)[fin]).mean() < 2e-3      # FP32 vs float64: a few grazing-angle texels
    assert (g['depth'] >= 1).mean() > 0.2 and (z['reveal'] < 1).mean() > 0.05       # the fixture does exercise sky and resolve


def test_fxaa_golden_fixture(coracle):
    z = np.load(os.path.join(GOLDEN, 'fxaa_101x70.npz'))
    np.testing.assert_array_equal(coracle.fxaa(z['ldr']), z['srgb8_c32'])
    d = np.abs(_ch(mo.fxaa(z['ldr'])) - _ch(z['srgb8_f64'])).max()
    assert d == 0


# ------------------------------------------------------------------ next rows f-4 (sky) and f-2 (transparency resolve)
sky_test_params = synth.hosek_params_synthetic


def test_kat9_sky_and_resolve_hand_derived(coracle):
    # sky: A = D = F = G = I = 0, C = (1, 2, 4), Z = (1, 1, .5): radiance = (1 + 0) * C * Z * 0.05 = (.05, .1, .1) on
    # texels at the clear depth, untouched elsewhere (sky_hosek_frag.glsl:41,51,58)
    prm = np.zeros((11, 4), np.float32)
    prm[2, :3], prm[9, :3], prm[10, :3] = (1, 2, 4), (1, 1, .5), (0, 1, 0)
    depth = np.array([[1.0, 0.5], [np.float32(0.99999994), 0.999999]], np.float32)
    frame = synth.sweep_camera(2, 2).frame_matrix_data()
    hdr = np.full((2, 2, 4), 7.0, np.float32)
    for out in (mo.sky_hosek(depth, frame, prm, hdr), coracle.sky_hosek(depth, frame, prm, hdr.copy())):
        np.testing.assert_allclose(out[0, 0], [.05, .1, .1, 1], rtol=1e-6)
        np.testing.assert_allclose(out[1, 0], [.05, .1, .1, 1], rtol=1e-6)
        assert np.all(out[0, 1] == 7) and np.all(out[1, 1] == 7)
    # resolve: accum (1, 2, 3, 2), reveal .25, hdr (4, 4, 4, 1): src = (.5, 1, 1.5); out = src * .75 + 4 * .25 =
    # (1.375, 1.75, 2.125), alpha = .25 * .75 + 1 * .25 = .4375 (fs_mcguire13_resolve.glsl:12, mcguire13/mod.rs:108-120)
    acc = np.array([[[1, 2, 3, 2]]], np.float16)
    rev = np.array([[0.25]], np.float16)
    h0 = np.array([[[4, 4, 4, 1]]], np.float32)
    for out in (mo.transparency_resolve(acc, rev, h0), coracle.transparency_resolve(acc, rev, h0.copy())):
        np.testing.assert_array_equal(out[0, 0], [1.375, 1.75, 2.125, 0.4375])
    # accum.a below 1e-4 clamps: (1e-5 stored as half) -> divide by 1e-4
    acc = np.array([[[1, 0, 0, 1e-5]]], np.float16)
    out = coracle.transparency_resolve(acc, np.array([[0.0]], np.float16), h0.copy())
    np.testing.assert_allclose(out[0, 0, 0], 1.0 / np.float32(1e-4), rtol=1e-6)


def transparency_test_inputs(h, w, seed=1):
    rng = np.random.RandomState(seed)
    acc = (rng.rand(h, w, 4) * np.array([3, 3, 3, 2])).astype(np.float16)
    empty = rng.rand(h, w) < 0.5
    acc[empty] = 0
    rev = rng.rand(h, w).astype(np.float16)
    rev[empty] = 1                                      # cleared: accum 0, reveal 1 (render/mod.rs:134-135)
    return acc, rev


def test_sky_and_resolve_two_restatements_agree(coracle):
    W, H = 160, 90
    cam = synth.sweep_camera(W, H)
    gb = synth.to_numpy(synth.scene_sweep(W, H, cam))
    frame = cam.frame_matrix_data()
    hdr0 = np.random.RandomState(0).rand(H, W, 4).astype(np.float32)
    a, b = mo.sky_hosek(gb['depth'], frame, sky_test_params(), hdr0), coracle.sky_hosek(gb['depth'], frame, sky_test_params(), hdr0.copy())
    sky = gb['depth'] >= 1
    assert 0.2 < sky.mean() < 0.8
    np.testing.assert_allclose(b, a, rtol=2e-5, atol=1e-6)
    assert np.array_equal(b[~sky], hdr0[~sky]) and np.all(b[sky][:, 3] == 1)
    y0, y1 = 16, 48                                     # strip == rows of the whole frame
    np.testing.assert_array_equal(coracle.sky_hosek(gb['depth'][y0:y1], frame, sky_test_params(), hdr0[y0:y1].copy(), y_offset=y0, full_height=H), b[y0:y1])
    acc, rev = transparency_test_inputs(H, W)
    a, b = mo.transparency_resolve(acc, rev, hdr0), coracle.transparency_resolve(acc, rev, hdr0.copy())
    np.testing.assert_allclose(b, a, rtol=1e-5, atol=1e-6)
    assert np.array_equal(b[rev == 1], hdr0[rev == 1])  # nothing transparent: hdr untouched
