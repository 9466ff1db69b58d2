"""GPU parity tests: the HIP path (through the C ABI of include/mil.h) against the CPU
oracles on the same seeded inputs.

Bars:
  * strict mode (MIL_FLAG_STRICT): BIT-EXACT against oracle/mil_oracle.c (FP32, GLSL
    operator order) for HDR, R11G11B10F and the LUT-based sRGB8 path;
  * fast mode (default): |a-b| <= 1e-3 + 1e-4*|b| in linear HDR against the float64
    oracle (BASELINE.md / north_star), bare, every pixel of every fixed test (round 3: the
    fast path forms D_GGX's 1 - (N.H)^2 without cancellation; the conditioning term
    8*2^-24*cond of rounds 1-2 remains only in the seeded fuzz, whose frames put a light ON a
    pixel), and <= 1 LSB after tonemap + sRGB8; PCF is a discrete test, so with shadows on at
    most 0.1 % of the pixels may differ by one tap;
  * integer / index work (culling, strips, fused-vs-unfused) bit-exact in both modes.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from oracle import mil_oracle as mo
from renderer_amd import lighting, synth, ubo

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
DEV = 'cuda:0'
AMB = (0.03, 0.03, 0.03)


@pytest.fixture(scope='module')
def lp():
    p = lighting.LightingPass(0, max_point_lights=4096)
    yield p
    p.close()


def planes_np(gb):
    g = synth.to_numpy(gb)
    return g['normal'], g['albedo_occ'], g['rough_metal'], g['depth']


def make_scene(kind, w, h, n_point=0, shadow=0, seed=0x5EED0000):
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV) if kind == 'sweep' else synth.scene_noise(w, h, seed, device=DEV)
    frame = cam.frame_matrix_data()
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(n_point, seed + 1) if n_point else None
    sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow else None
    return gb, frame, dl, pl, sh


def sh_np(sh):
    return None if sh is None else dict(sh, map=sh['map'].cpu().numpy())


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_bits_equal(a, b, what=''):
    a, b = bits(a), bits(b)
    if not np.array_equal(a, b):
        bad = np.argwhere(a != b)
        raise AssertionError(f'{what}: {len(bad)} words differ, first at {bad[0]}')


def fast_tolerance_ok(gpu, planes, frame, dl, pl=None, amb=None, sh=None, y_offset=0, full_height=None,
                      max_bad_frac=0.0):
    """north_star's bar, bare: |a - b| <= 1e-3 + 1e-4 |b| per channel against the float64 oracle, every pixel (max_bad_frac
    > 0 only where a PCF tap may sit on the bias).  Rounds 1-2 added a conditioning term 8 x 2^-24 x cond for D_GGX at
    mirror alignment; since the fast path forms 1 - (N.H)^2 without cancellation (one_minus_ndoth2) no fixed test needs it."""
    ref = mo.shade(*planes, frame, dl, point_lights=pl, ambient=amb, shadow=sh, y_offset=y_offset, full_height=full_height)
    tol = 1e-3 + 1e-4 * np.abs(ref[..., :3])
    err = np.abs(gpu[..., :3].astype(np.float64) - ref[..., :3])
    bad = (err > tol) | ~np.isfinite(gpu[..., :3])
    frac = bad.any(-1).mean()
    assert frac <= max_bad_frac, f'{bad.any(-1).sum()} pixels outside tolerance (max err {err.max():.3g})'
    assert np.all(gpu[..., 3] == 1.0)
    return ref


# ------------------------------------------------------------------ K1: the reference path
@pytest.mark.parametrize('kind,w,h', [('sweep', 320, 180), ('noise', 256, 144), ('sweep', 67, 35)])
def test_k1_directional_strict_bit_exact(lp, coracle, kind, w, h):
    gb, frame, dl, _, _ = make_scene(kind, w, h)
    out = lp.light_directional(gb, frame, dl, strict=True).cpu().numpy()
    assert_bits_equal(out, coracle.shade(*planes_np(gb), frame, dl), 'K1 strict vs C oracle')


@pytest.mark.parametrize('kind,w,h', [('sweep', 320, 180), ('noise', 256, 144)])
def test_k1_directional_fast_tolerance(lp, kind, w, h):
    gb, frame, dl, _, _ = make_scene(kind, w, h)
    out = lp.light_directional(gb, frame, dl).cpu().numpy()
    fast_tolerance_ok(out, planes_np(gb), frame, dl)


def test_k1_many_lights_and_limits(lp, coracle):
    gb, frame, _, _, _ = make_scene('sweep', 96, 54)
    rng = np.random.RandomState(5)
    d = rng.randn(100, 3)
    d[:, 1] = np.abs(d[:, 1])
    dl = ubo.directional_lights([(d[i], 0.05 + rng.rand(), 0.3 + 0.7 * rng.rand(3)) for i in range(100)])
    out = lp.light_directional(gb, frame, dl, strict=True).cpu().numpy()      # MAX_LIGHTS = 100 (inc_structs.glsl:1)
    assert_bits_equal(out, coracle.shade(*planes_np(gb), frame, dl), '100 lights')
    fast_tolerance_ok(lp.light_directional(gb, frame, dl).cpu().numpy(), planes_np(gb), frame, dl)
    with pytest.raises(lighting.MilError, match='-2'):                      # the reference panics at 101 (render/mod.rs:197-199)
        lp.light_directional(gb, frame, np.concatenate([dl, dl[:1]]))
    zero = lp.light_directional(gb, frame, dl[:0]).cpu().numpy()            # light_count = 0
    assert np.all(zero[..., :3] == 0) and np.all(zero[..., 3] == 1)


def test_background_and_cleared_gbuffer(lp):
    w, h = 80, 48
    z = torch.zeros((h, w), dtype=torch.int32, device=DEV)
    gb = dict(normal=z, albedo_occ=z, rough_metal=z, depth=torch.ones((h, w), device=DEV), width=w, height=h, y_offset=0)
    frame = synth.sweep_camera(w, h).frame_matrix_data()
    for strict in (False, True):
        out = lp.light_directional(gb, frame, ubo.default_directional_lights(), strict=strict).cpu().numpy()
        assert np.all(out[..., :3] == 0) and np.all(out[..., 3] == 1)       # render/mod.rs:129-133 clear values -> 0, not NaN


def test_ragged_sizes_and_pitch(lp, coracle):
    for (w, h) in [(1, 1), (17, 5), (64, 4), (65, 5), (130, 33)]:
        gb, frame, dl, pl, sh = make_scene('noise', w, h, n_point=9, shadow=32, seed=w * 131 + h)
        ref = coracle.shade(*planes_np(gb), frame, dl)
        assert_bits_equal(lp.light_directional(gb, frame, dl, strict=True).cpu().numpy(), ref, f'{w}x{h}')
        ref = coracle.shade(*planes_np(gb), frame, dl, point_lights=pl, ambient=AMB, shadow=sh_np(sh))
        assert_bits_equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_STRICT).cpu().numpy(), ref, f'ex {w}x{h}')
    # planes that are views into wider allocations (pitch_px > width), output likewise
    w, h, pitch = 50, 20, 64
    big, frame, dl, _, _ = make_scene('noise', pitch, h, seed=77)
    gb = {k: (v[:, 3:3 + w] if torch.is_tensor(v) else v) for k, v in big.items()}
    gb.update(width=w, height=h)
    outbuf = torch.full((h, pitch, 4), -7.0, device=DEV)
    out = lp.light_directional(gb, frame, dl, out=outbuf[:, 5:5 + w], strict=True)
    gbc = {k: (v.contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
    assert_bits_equal(out.cpu().numpy(), coracle.shade(*planes_np(gbc), frame, dl), 'pitched')
    assert torch.all(outbuf[:, :5] == -7.0) and torch.all(outbuf[:, 5 + w:] == -7.0)


def test_bad_arguments_return_status(lp):
    lib = lp.lib
    gb, frame, dl, _, _ = make_scene('noise', 32, 16)
    g = lp._gb(gb)
    push = ubo.push_constants(32, 16, 2)
    out = torch.empty((16, 32, 4), device=DEV)
    fp = lambda a: ctypes.c_void_p(a.ctypes.data)
    assert lib.mil_light_directional(lp.ctx, None, fp(frame), fp(dl), fp(push), out.data_ptr(), 32, 0, None) == -1
    assert lib.mil_light_directional(lp.ctx, ctypes.byref(g), fp(frame), fp(dl), fp(push), None, 32, 0, None) == -1
    assert lib.mil_light_directional(lp.ctx, ctypes.byref(g), fp(frame), fp(dl), fp(push), out.data_ptr(), 31, 0, None) == -1
    g2 = lp._gb(gb)
    g2.pitch_px = 8
    assert lib.mil_light_directional(lp.ctx, ctypes.byref(g2), fp(frame), fp(dl), fp(push), out.data_ptr(), 32, 0, None) == -1
    small = lighting.LightingPass(0, max_point_lights=4)
    with pytest.raises(lighting.MilError, match='-5'):
        small.light_ex(gb, frame, dl, synth.point_light_field(5, 1))
    small.close()
    assert lib.mil_tonemap(lp.ctx, out.data_ptr(), 32, 32, 16, out.data_ptr(), 32, 0, 0, None) == -1   # HDR is not a tonemap target


# ------------------------------------------------------------------ K4: tonemap / pack
def hdr_samples(n=64):
    rng = np.random.RandomState(3)
    a = np.concatenate([rng.rand(n, n, 4) * 8, rng.rand(n, n, 4) * 0.02, rng.rand(n, n, 4) * 3e-5,
                        10.0 ** rng.uniform(-8, 6, (n, n, 4))], 0).astype(np.float32)
    a[0, 0] = [0, 1, 10, 1]
    a[0, 1] = [1e30, 1e19, 65504, 1]        # x*x overflows FP32: see test_k4_tonemap_fast_tolerance
    return a


def channels(u):
    return np.stack([(u >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(int)


def test_k4_tonemap_strict_bit_exact(lp, coracle):
    hdr = hdr_samples()
    t = torch.from_numpy(hdr).to(DEV)
    for fmt in (lighting.OUT_LDR, lighting.OUT_SRGB8_VIA_LDR):
        out = lp.tonemap(t, fmt, strict=True).cpu().numpy().view(np.uint32)
        np.testing.assert_array_equal(out, coracle.tonemap(hdr, fmt))
    out = lp.tonemap(t, lighting.OUT_SRGB8, strict=True).cpu().numpy().view(np.uint32)
    d = np.abs(channels(out) - channels(coracle.tonemap(hdr, 2)))                   # OETF in double on both sides;
    assert d.max() <= 1 and (d > 0).mean() < 1e-5                                   # only pow()'s last ulp could differ


def test_k4_tonemap_fast_tolerance(lp, coracle):
    hdr = hdr_samples()
    t = torch.from_numpy(hdr).to(DEV)
    ref = mo.tonemap(hdr, mo.OUT_LDR)
    out = lp.tonemap(t, lighting.OUT_LDR).cpu().numpy().view(np.uint32)
    # pixel (0,1): x*x overflows FP32 above ~1.8e19.  Strict mode reproduces the literal FP32 result
    # (inf/inf = NaN -> clamp -> 0, test above); fast mode saturates x first and agrees with float64 (1.0).
    d = out != ref
    assert d.mean() < 2e-3                                                   # rounding-boundary straddles only ...
    step = np.abs(np.stack([(out & 0x7FF).astype(int) - (ref & 0x7FF), ((out >> 11) & 0x7FF).astype(int) - ((ref >> 11) & 0x7FF),
                            (out >> 22).astype(int) - (ref >> 22)], -1))
    assert step.max() <= 1                                                   # ... and never more than one code
    out8 = lp.tonemap(t, lighting.OUT_SRGB8).cpu().numpy().view(np.uint32)
    ref8 = mo.tonemap(hdr, mo.OUT_SRGB8)
    assert np.abs(channels(out8) - channels(ref8)).max() <= 1               # <= 1 LSB after 8-bit tonemap
    assert np.all(channels(out8)[..., 3] == 255)


@pytest.mark.parametrize('strict', [False, True])
def test_fused_equals_unfused(lp, strict):
    gb, frame, dl, pl, sh = make_scene('sweep', 200, 120, n_point=40, shadow=128)
    fl = lighting.FLAG_STRICT if strict else 0
    hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=fl)
    for fmt in (lighting.OUT_LDR, lighting.OUT_SRGB8, lighting.OUT_SRGB8_VIA_LDR):
        fused = lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=fmt, flags=fl)
        assert torch.equal(fused, lp.tonemap(hdr, fmt, strict=strict)), fmt
        fused = lp.light_ex(gb, frame, dl, fmt=fmt, flags=fl)                # streaming kernel, reference path only
        assert torch.equal(fused, lp.tonemap(lp.light_directional(gb, frame, dl, strict=strict), fmt, strict=strict)), fmt


# ------------------------------------------------------------------ extensions E1-E4
@pytest.mark.parametrize('kind,w,h,n_point', [('sweep', 320, 180, 64), ('noise', 128, 72, 24), ('sweep', 96, 54, 1500)])
def test_extensions_strict_bit_exact(lp, coracle, kind, w, h, n_point):
    gb, frame, dl, pl, sh = make_scene(kind, w, h, n_point=n_point, shadow=256)
    ref = coracle.shade(*planes_np(gb), frame, dl, point_lights=pl, ambient=AMB, shadow=sh_np(sh))
    S = lighting.FLAG_STRICT
    for fl in (S, S | lighting.FLAG_NO_CULL, S | lighting.FLAG_SHADOW_LDS, S | lighting.FLAG_NO_CULL | lighting.FLAG_SHADOW_LDS):
        assert_bits_equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=fl).cpu().numpy(), ref, f'flags {fl}')


def test_golden_fixtures_on_gpu(lp):
    for name in ('sweep_96x54', 'noise_64x36'):
        z = np.load(os.path.join(GOLDEN, name + '.npz'))
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a).view(dt)).to(DEV)
        h, w = z['depth'].shape
        gb = dict(normal=t(z['normal'], np.int32), albedo_occ=t(z['albedo_occ'], np.int32),
                  rough_metal=t(z['rough_metal'], np.int32), depth=t(z['depth'], np.float32), width=w, height=h, y_offset=0)
        sh = dict(light_view_proj=z['shadow_lvp'], map=t(z['shadow_map'], np.float32), bias=float(z['shadow_bias']))
        S = lighting.FLAG_STRICT
        assert_bits_equal(lp.light_directional(gb, z['frame'], z['dir_lights'], strict=True).cpu().numpy(), z['hdr_dir_only_c32'], name)
        hdr = lp.light_ex(gb, z['frame'], z['dir_lights'], z['point_lights'], z['ambient'], sh, flags=S)
        assert_bits_equal(hdr.cpu().numpy(), z['hdr_c32'], name)
        np.testing.assert_array_equal(lp.tonemap(hdr, lighting.OUT_LDR, strict=True).cpu().numpy().view(np.uint32), z['ldr_c32'])
        np.testing.assert_array_equal(lp.tonemap(hdr, lighting.OUT_SRGB8_VIA_LDR, strict=True).cpu().numpy().view(np.uint32), z['srgb8_via_ldr_c32'])
        # fast mode: the bar of every other fast-mode test (fast_tolerance_ok), PCF taps on the bias aside
        planes = tuple(np.ascontiguousarray(z[k]) for k in ('normal', 'albedo_occ', 'rough_metal', 'depth'))
        shn = dict(light_view_proj=z['shadow_lvp'], map=z['shadow_map'], bias=float(z['shadow_bias']))
        fast = lp.light_ex(gb, z['frame'], z['dir_lights'], z['point_lights'], z['ambient'], None).cpu().numpy()
        fast_tolerance_ok(fast, planes, z['frame'], z['dir_lights'], z['point_lights'], z['ambient'], None)
        fast = lp.light_ex(gb, z['frame'], z['dir_lights'], z['point_lights'], z['ambient'], sh).cpu().numpy()
        ref = fast_tolerance_ok(fast, planes, z['frame'], z['dir_lights'], z['point_lights'], z['ambient'], shn, max_bad_frac=1e-3)
        np.testing.assert_allclose(ref, z['hdr_f64'], rtol=1e-9, atol=1e-12)      # the committed float64 image is what the oracle makes today


@pytest.mark.parametrize('kind,w,h', [('sweep', 320, 180), ('noise', 128, 72)])
def test_extensions_fast_tolerance(lp, kind, w, h):
    gb, frame, dl, pl, sh = make_scene(kind, w, h, n_point=64, shadow=256)
    out = lp.light_ex(gb, frame, dl, pl, AMB, None).cpu().numpy()
    fast_tolerance_ok(out, planes_np(gb), frame, dl, pl, AMB, None)
    out = lp.light_ex(gb, frame, dl, pl, AMB, sh).cpu().numpy()            # PCF is discrete: allow rare one-tap flips
    ref = fast_tolerance_ok(out, planes_np(gb), frame, dl, pl, AMB, sh_np(sh), max_bad_frac=1e-3)
    srgb = lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8).cpu().numpy().view(np.uint32)
    d = np.abs(channels(srgb) - channels(mo.tonemap(ref, mo.OUT_SRGB8))).max(-1)
    assert (d > 1).mean() <= 1e-3                                            # <= 1 LSB after 8-bit tonemap (tap flips aside)


@pytest.mark.parametrize('strict', [False, True])
def test_culled_equals_brute_force(lp, strict):
    """Tile lists keep ascending light order, so culling never changes a bit."""
    fl = lighting.FLAG_STRICT if strict else 0
    for kind, w, h, n in [('sweep', 320, 180, 256), ('sweep', 100, 60, 2500), ('noise', 64, 48, 300)]:
        gb, frame, dl, pl, _ = make_scene(kind, w, h, n_point=n)
        counts = lp.tile_counts(gb)
        a = lp.light_ex(gb, frame, dl, pl, flags=fl)
        lp.tile_counts(gb, enable=False)
        b = lp.light_ex(gb, frame, dl, pl, flags=fl | lighting.FLAG_NO_CULL)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (kind, n)
        if kind == 'sweep':
            assert 0 < counts.float().mean().item() < 0.6 * n                # the culling does cull


def test_tile_list_overflow_rounds(lp, coracle):
    """More lights reach a tile than its LDS list holds (1024): the tile is shaded in rounds, partial sums parked in LDS.
    Every light reaches every pixel here (radius 60), 1500 and 2300 of them: still == brute force bit for bit in both
    modes and == the C oracle in strict mode."""
    for n in (1500, 2300):
        gb, frame, dl, pl, sh = make_scene('sweep', 80, 48, n_point=n, shadow=64)
        pl = pl.copy()
        pl[:, 3] = 60.0
        pl[:, 7] = 0.01
        counts = lp.tile_counts(gb)
        a = lp.light_ex(gb, frame, dl, pl, AMB, sh)
        lp.tile_counts(gb, enable=False)
        lit_tiles = (((gb['albedo_occ'] >> 24) & 0xFF) != 0).view(3, 16, 5, 16).any(3).any(1)
        assert torch.all(counts[lit_tiles] == n) and counts.max().item() == n
        b = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
        S = lighting.FLAG_STRICT
        ref = coracle.shade(*planes_np(gb), frame, dl, point_lights=pl, ambient=AMB, shadow=sh_np(sh))
        assert_bits_equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=S).cpu().numpy(), ref, f'overflow rounds strict {n}')
        assert_bits_equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=S | lighting.FLAG_NO_CULL).cpu().numpy(), ref, f'brute strict {n}')
        one = lp.shade_frame(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_LDR)
        assert torch.equal(one, lp.tonemap(a, lighting.OUT_LDR))


@pytest.mark.parametrize('strict', [False, True])
def test_extensions_vanish_when_disabled(lp, strict):
    gb, frame, dl, _, _ = make_scene('sweep', 160, 90)
    fl = lighting.FLAG_STRICT if strict else 0
    a = lp.light_directional(gb, frame, dl, strict=strict)
    b = lp.light_ex(gb, frame, dl, None, (0, 0, 0), None, flags=fl)
    c = lp.light_ex(gb, frame, dl, synth.point_light_field(16, 3) * np.array([1, 1, 1, 0, 1, 1, 1, 1], np.float32), flags=fl)  # radius 0
    assert torch.equal(a.view(torch.int32), b.view(torch.int32)) and torch.equal(a.view(torch.int32), c.view(torch.int32))


@pytest.mark.parametrize('strict', [False, True])
def test_strips_equal_monolithic(lp, strict):
    """Strip-sharded == monolithic, bit for bit: global y via y_offset."""
    w, h = 256, 144
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=48, shadow=128)
    fl = lighting.FLAG_STRICT if strict else 0
    full = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=fl)
    for bounds in ([0, 48, 96, 144], [0, 37, 101, 144]):
        for y0, y1 in zip(bounds[:-1], bounds[1:]):
            strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
            strip['y_offset'] = y0
            part = lp.light_ex(strip, frame, dl, pl, AMB, sh, flags=fl)
            assert torch.equal(part.view(torch.int32), full[y0:y1].view(torch.int32)), (y0, y1)
            gen = synth.scene_sweep(w, h, synth.sweep_camera(w, h), device=DEV, rows=(y0, y1))   # generator agrees too
            assert torch.equal(gen['depth'], gb['depth'][y0:y1]) and torch.equal(gen['normal'], gb['normal'][y0:y1])


def test_light_staging_ring_and_cache(lp, coracle):
    """Light arrays are staged through a 4-slot ring and re-used when unchanged: alternate more
    distinct light sets than slots, with repeats, and check every frame against the oracle."""
    gb, frame, dl, _, _ = make_scene('sweep', 96, 54)
    sets = [synth.point_light_field(40 + 7 * i, 100 + i) for i in range(6)]
    order = [0, 0, 1, 2, 3, 4, 5, 0, 5, 5, 1]
    outs = [lp.light_ex(gb, frame, dl, sets[i], flags=lighting.FLAG_STRICT) for i in order]   # enqueue all, then check
    for i, o in zip(order, outs):
        assert_bits_equal(o.cpu().numpy(), coracle.shade(*planes_np(gb), frame, dl, point_lights=sets[i]), f'set {i}')
    many = ubo.directional_lights([((1, 2 + i, 0.5 * i), 0.1, (1, 1, 1)) for i in range(9)])   # > 4 inline records
    a = lp.light_directional(gb, frame, many, strict=True).cpu().numpy()
    assert_bits_equal(a, coracle.shade(*planes_np(gb), frame, many), '9 directional lights')


def test_host_caches_follow_camera_strip_and_stream_changes(lp):
    """The host caches (staged lights recognised from the input arrays, K2 row order per lights / camera / strip) must
    never change a result: same lights under a moving camera, different strips, alternating streams and a light edited
    in place, each culled launch == its brute-force twin; contexts come and go."""
    from renderer_amd import sharding
    w, h = 192, 112
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=90, shadow=64)
    pl = pl.copy()
    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    torch.cuda.synchronize()
    for i in range(8):
        cam = sharding.orbit_camera(i % 5, 5, w, h)
        g2 = synth.scene_sweep(w, h, cam, device=DEV)
        fr = cam.frame_matrix_data()
        if i == 5:
            pl[3, 0:3] += 0.5                                                 # edited in place: same pointer, new content
        st = streams[i % 2]
        y0, y1 = (0, h) if i % 3 else (32, 80)
        strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in g2.items()}
        strip['y_offset'] = y0
        a = lp.light_ex(strip, fr, dl, pl, AMB, sh, stream=st)
        b = lp.light_ex(strip, fr, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL, stream=st)
        st.synchronize()
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), i
    for _ in range(6):
        lp2 = lighting.LightingPass(0, 128)
        a = lp2.light_ex(gb, frame, dl, pl, AMB, sh)
        lp2.sync()
        assert torch.equal(a.view(torch.int32), lp.light_ex(gb, frame, dl, pl, AMB, sh).view(torch.int32))
        lp2.close()


def test_count_pairs(lp, coracle):
    gb, frame, dl, pl, _ = make_scene('sweep', 160, 90, n_point=100)
    _, pairs = coracle.shade(*planes_np(gb), frame, dl[:0], point_lights=pl, return_pairs=True)
    got = lp.count_pairs(gb, frame, pl)
    assert abs(got - pairs) <= max(2, 1e-4 * pairs)


def test_fuzz_random_frames_and_degenerate_lights(lp, coracle):
    """Seeded fuzz: random sizes (tile-edge cases), noise G-buffers, lights with radius 0, huge radius, a light
    sitting exactly on a pixel's world position (d = 0), zero intensity, more lights than one 64-entry block."""
    rng = np.random.RandomState(1234)
    for it in range(6):
        w, h = int(rng.randint(1, 150)), int(rng.randint(1, 70))
        n = int(rng.choice([1, 7, 64, 65, 130, 300]))
        gb, frame, dl, pl, sh = make_scene('noise' if it % 2 else 'sweep', w, h, n_point=n, shadow=64, seed=1000 + it)
        pl = pl.copy()
        pl[0, 3] = 0.0                                   # radius 0: never in range
        if n > 2:
            pl[1, 3] = 1e4                               # reaches every pixel
            pl[2, 7] = 0.0                               # zero intensity
        planes = planes_np(gb)
        pos = mo.position_from_depth(planes[3].astype(np.float64), w, h, 0, *[mo.unpack_frame_ubo(frame)[k] for k in ('inv_projection', 'inv_view')])
        if n > 3:
            pl[3, 0:3] = pos[h // 2, w // 2].astype(np.float32)   # on top of a pixel (up to FP32 rounding of P)
            pl[3, 3] = 2.0
        ref = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=AMB, shadow=sh_np(sh))
        S = lighting.FLAG_STRICT
        assert_bits_equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=S).cpu().numpy(), ref, f'fuzz {it} strict')
        assert_bits_equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=S | lighting.FLAG_NO_CULL).cpu().numpy(), ref, f'fuzz {it} strict brute')
        a = lp.light_ex(gb, frame, dl, pl, AMB, sh)
        b = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL)
        assert torch.equal(a.view(torch.int32), b.view(torch.int32)), f'fuzz {it}: culled != brute force'
        # fast mode against the float64 oracle: the bare bar + the conditioning term (the light sitting on a pixel is a hot
        # spot 1 / (d^2 + 1e-4) that amplifies the FP32 rounding of the reconstructed position; noise frames hold
        # roughness byte 0, alpha = 1e-8); a PCF tap on the bias may flip a pixel or two
        r64, cond = coracle.shade_f64(*planes, frame, dl, point_lights=pl, ambient=AMB, shadow=sh_np(sh), want_cond=True)
        fin = np.isfinite(r64[..., :3]).all(-1)
        fa = a.cpu().numpy()
        err = np.abs(fa[..., :3] - r64[..., :3])
        bare = ((err > 1e-3 + 1e-4 * np.abs(r64[..., :3])).any(-1) & fin).sum()
        bad = ((err > 1e-3 + 1e-4 * np.abs(r64[..., :3]) + 8 * 2.0 ** -24 * cond).any(-1) & fin).sum()
        print(f'fuzz {it} ({w}x{h}, {n} lights): {int(bare)} pixels outside the bare bar, {int(bad)} outside bar + conditioning term')
        assert bad <= max(2, 2e-3 * w * h), (it, int(bad))


# ------------------------------------------------------------------ next row f-1: FXAA
def test_fxaa_strict_bit_exact_and_fast_tolerance(lp, coracle):
    from test_oracle_kat import fxaa_test_image
    z = np.load(os.path.join(GOLDEN, 'fxaa_101x70.npz'))
    for ldr in (z['ldr'], fxaa_test_image(144, 256, 9), fxaa_test_image(33, 65, 1), fxaa_test_image(1, 1), fxaa_test_image(5, 3)):
        t = torch.from_numpy(ldr.view(np.int32)).to(DEV)
        ref = coracle.fxaa(ldr)
        np.testing.assert_array_equal(lp.fxaa(t, strict=True).cpu().numpy().view(np.uint32), ref)   # FP32 in GLSL order: every texel pick identical
        fast = lp.fxaa(t).cpu().numpy().view(np.uint32)
        d = np.abs(channels(fast) - channels(mo.fxaa(ldr))).max(-1)
        assert (d > 1).mean() <= 2e-3                                       # <= 1 LSB; a nearest-texel pick may flip on a boundary
        assert np.all(channels(fast)[..., 3] == 255)
    np.testing.assert_array_equal(lp.fxaa(torch.from_numpy(z['ldr'].view(np.int32)).to(DEV), strict=True).cpu().numpy().view(np.uint32),
                                  z['srgb8_c32'])


def test_fxaa_strips_pitch_and_bad_args(lp):
    from test_oracle_kat import fxaa_test_image
    ldr = fxaa_test_image(96, 160, 5)
    t = torch.from_numpy(ldr.view(np.int32)).to(DEV)
    for strict in (False, True):
        full = lp.fxaa(t, strict=strict)
        for y0, n in [(0, 16), (16, 48), (64, 32), (95, 1)]:               # a strip reads its halo from the whole frame
            assert torch.equal(lp.fxaa(t, y0, n, strict=strict), full[y0:y0 + n])
        big = torch.zeros((96, 200), dtype=torch.int32, device=DEV)
        big[:, 7:167] = t
        outbuf = torch.full((96, 180), -7, dtype=torch.int32, device=DEV)
        o = lp.fxaa(big[:, 7:167], out=outbuf[:, 3:163], strict=strict)     # pitched input and output
        assert torch.equal(o, full) and torch.all(outbuf[:, :3] == -7) and torch.all(outbuf[:, 163:] == -7)
    lib = lp.lib
    out = torch.empty((96, 160), dtype=torch.int32, device=DEV)
    assert lib.mil_fxaa(lp.ctx, None, 160, 160, 96, 0, 96, out.data_ptr(), 160, 0, None) == -1
    assert lib.mil_fxaa(lp.ctx, t.data_ptr(), 100, 160, 96, 0, 96, out.data_ptr(), 160, 0, None) == -1     # pitch < width
    assert lib.mil_fxaa(lp.ctx, t.data_ptr(), 160, 160, 96, 90, 7, out.data_ptr(), 160, 0, None) == -1      # rows beyond the frame
    assert lib.mil_fxaa(lp.ctx, t.data_ptr(), 160, 160, 96, 0, 0, out.data_ptr(), 160, 0, None) == -1


def test_full_chain_4k_gbuffer_to_swapchain(lp):
    """G-buffer -> lighting -> tonemap (R11G11B10F) -> FXAA -> B8G8R8A8 sRGB at 3840x2160, against the oracle
    chain on a 48-row band (the band plus its 4-row FXAA halo is shaded by the oracle)."""
    w, h = 3840, 2160
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=256, shadow=2048)
    hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh)
    ldr = lp.tonemap(hdr, lighting.OUT_LDR)
    img = lp.fxaa(ldr)
    assert torch.all((img >> 24) & 0xFF == 255)
    y0, y1 = 992, 1040
    # the oracle treats the band (+- 8 halo rows) as its whole frame: the taps move by the same pixel offsets
    # (dir * inverseVP * size), so the texel picks agree up to FP rounding on a boundary
    ldr_band = ldr[y0 - 8:y1 + 8].cpu().numpy().view(np.uint32)
    ref = mo.fxaa(ldr_band, 8, y1 - y0)
    d = np.abs(channels(img[y0:y1].cpu().numpy().view(np.uint32)) - channels(ref)).max(-1)
    assert (d > 1).mean() <= 5e-3


# ------------------------------------------------------------------ next rows f-4 (sky) and f-2 (transparency resolve)
def test_sky_hosek(lp, coracle):
    from test_oracle_kat import sky_test_params
    prm = sky_test_params()
    for (w, h) in [(320, 180), (67, 35)]:
        gb, frame, dl, _, _ = make_scene('sweep', w, h)
        depth = gb['depth'].cpu().numpy()
        hdr0 = lp.light_directional(gb, frame, dl)
        ref32 = coracle.sky_hosek(depth, frame, prm, hdr0.cpu().numpy())
        ref64 = mo.sky_hosek(depth, frame, prm, hdr0.cpu().numpy())
        strict = lp.sky_hosek(gb, frame, prm, hdr0.clone(), strict=True).cpu().numpy()
        # FP32 in GLSL order with exp / pow / acos taken in double on both sides: equal up to the last ulp of those
        np.testing.assert_allclose(strict, ref32, rtol=3e-7, atol=0)
        assert (strict != ref32).mean() < 1e-3
        fast = lp.sky_hosek(gb, frame, prm, hdr0.clone()).cpu().numpy()
        assert np.all(np.abs(fast - ref64) <= 1e-3 + 1e-4 * np.abs(ref64))
        sky = depth >= 1
        assert np.array_equal(fast[~sky], hdr0.cpu().numpy()[~sky]) and np.all(fast[sky][:, 3] == 1)
        y0, y1 = 16, 32                                                     # strip == rows of the whole frame, bit for bit
        strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
        strip['y_offset'] = y0
        for st in (False, True):
            full = lp.sky_hosek(gb, frame, prm, hdr0.clone(), strict=st)
            assert torch.equal(lp.sky_hosek(strip, frame, prm, hdr0[y0:y1].clone(), strict=st), full[y0:y1])


@pytest.mark.parametrize('name', ['default', 'basic'])
def test_sky_hosek_on_reference_coefficient_blocks(lp, coracle, name):
    """k_sky had only seen synthetic coefficients.  tests/golden/hosek_params.npz holds the blocks the reference
    computes for HosekSky::new and for scenes/basic.rs (render/hosek/mod.rs:123-224 on its dataset.rs) and what its own
    sky_hosek_frag.glsl, executed, makes of them: the strict kernel reproduces that image (to the last ulp of exp / pow /
    acos, as for the synthetic block), the fast kernel is within 1e-3 + 1e-4 |x| of float64, and mil_shade_frame with that
    sky equals the chain of separate calls bit for bit."""
    z = np.load(os.path.join(GOLDEN, 'hosek_params.npz'))
    g = np.load(os.path.join(GOLDEN, 'sweep_96x54.npz'))
    prm, img = z['params_' + name], z['sky_hdr_' + name]
    gb = {k: torch.from_numpy(g[k].view(np.int32) if g[k].dtype == np.uint32 else g[k]).to(DEV) for k in ('normal', 'albedo_occ', 'rough_metal', 'depth')}
    gb.update(width=96, height=54, y_offset=0)
    frame, dl = g['frame'], g['dir_lights']
    hdr0 = lp.light_directional(gb, frame, dl, strict=True)
    assert_bits_equal(hdr0.cpu().numpy(), g['hdr_dir_only_c32'], 'lighting under the sky')
    strict = lp.sky_hosek(gb, frame, prm, hdr0.clone(), strict=True).cpu().numpy()
    np.testing.assert_allclose(strict, img, rtol=3e-7, atol=0)
    assert (strict != img).mean() < 1e-3
    f0 = lp.light_directional(gb, frame, dl)
    fast = lp.sky_hosek(gb, frame, prm, f0.clone()).cpu().numpy()
    r64 = mo.sky_hosek(g['depth'], frame, prm, f0.cpu().numpy())
    assert np.all(np.abs(fast - r64) <= 1e-3 + 1e-4 * np.abs(r64))
    sky = g['depth'] >= 1
    assert sky.mean() > 0.2 and np.all(fast[sky][:, :3] > 0)
    for st in (False, True):
        for fmt in (lighting.OUT_HDR, lighting.OUT_SRGB8):
            chain = lp.sky_hosek(gb, frame, prm, lp.light_directional(gb, frame, dl, strict=st), strict=st)
            if fmt != lighting.OUT_HDR:
                chain = lp.tonemap(chain, fmt, strict=st)
            one = lp.shade_frame(gb, frame, dl, sky=prm, fmt=fmt, flags=lighting.FLAG_STRICT if st else 0)
            assert torch.equal(one.view(torch.int32), chain.view(torch.int32)), (name, st, fmt)


def test_transparency_resolve(lp, coracle):
    from test_oracle_kat import transparency_test_inputs
    for (w, h) in [(256, 144), (65, 5)]:
        acc, rev = transparency_test_inputs(h, w, seed=w)
        hdr0 = np.random.RandomState(2).rand(h, w, 4).astype(np.float32) * 4
        ta, tr, th = torch.from_numpy(acc).to(DEV), torch.from_numpy(rev).to(DEV), torch.from_numpy(hdr0).to(DEV)
        strict = lp.transparency_resolve(ta, tr, th.clone(), strict=True).cpu().numpy()
        assert_bits_equal(strict, coracle.transparency_resolve(acc, rev, hdr0.copy()), 'resolve strict')
        fast = lp.transparency_resolve(ta, tr, th.clone()).cpu().numpy()
        ref = mo.transparency_resolve(acc, rev, hdr0)
        assert np.all(np.abs(fast - ref) <= 1e-3 + 1e-4 * np.abs(ref))
        assert np.array_equal(fast[rev == 1], hdr0[rev == 1])               # nothing transparent there: hdr untouched


def frags_to_device(fr):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in fr.items()}


def test_transparency_accumulate(lp, coracle):
    """mil_transparency_accumulate (fs_mcguire13_accumulation.glsl over a fragment list): strict bit for bit against the C
    oracle, fast within 1e-3 + 1e-4 |x| (+ the D_GGX conditioning term) of the float64 restatement; ragged sizes, 0 / 1 / 7 / 100 directional lights
    (beyond the 4 inline records the raw array comes from a staged slot), the empty list, the light-count limit."""
    cam = synth.sweep_camera(320, 180)
    frame = cam.frame_matrix_data()
    rng = np.random.RandomState(3)
    for n, n_dir in [(1, 2), (63, 1), (257, 0), (5000, 2), (40001, 7), (3000, 100)]:
        fr = synth.transparent_fragments(n, seed=n)
        dl = np.concatenate([ubo.default_directional_lights(),
                             ubo.directional_lights([(d, 0.5 + i, rng.rand(3)) for i, d in enumerate(rng.randn(max(n_dir - 2, 0), 3))])])[:n_dir]
        dev = frags_to_device(fr)
        args = [fr[k] for k in ('position_depth', 'normal_opacity', 'albedo_roughness', 'metallic')] + [frame[64:67], dl]
        ca, cr = coracle.transparency_accumulate(*args)
        sa, sr = lp.transparency_accumulate(dev, frame, dl, strict=True)
        assert_bits_equal(sa.cpu().numpy(), ca, f'accumulate strict n={n} lights={n_dir}')
        assert np.array_equal(sr.cpu().numpy(), cr)
        fa, frv = lp.transparency_accumulate(dev, frame, dl)
        ra, rr, cond = mo.transparency_accumulate(*args, want_cond=True)
        assert np.isfinite(ra).all()
        bare = 1e-3 + 1e-4 * np.abs(ra)                                  # north_star's bar, bare, for the fast kernel
        assert np.all(np.abs(fa.cpu().numpy() - ra) <= bare), (n, n_dir, float((np.abs(fa.cpu().numpy() - ra) / bare).max()))
        assert np.all(np.abs(ca - ra) <= bare + 8 * 2.0 ** -24 * cond)  # the reference-order FP32 evaluation needs the D_GGX conditioning term
        assert np.array_equal(frv.cpu().numpy(), fr['normal_opacity'][:, 3])
    # the lighting call of the frame and the accumulate call share the staged light slot (same lights UBO in the reference)
    gb, frame2, dl2, pl, _ = make_scene('sweep', 96, 64, n_point=16)
    hdr_a = lp.light_ex(gb, frame2, dl, pl, AMB, None)
    fa2, _ = lp.transparency_accumulate(dev, frame, dl)
    hdr_b = lp.light_ex(gb, frame2, dl, pl, AMB, None)
    assert torch.equal(fa2, fa) and torch.equal(hdr_a, hdr_b)
    # empty list: nothing to do; too many lights: the reference panics (render/mod.rs:197-199), this returns the status
    empty = {k: v[:0].contiguous() for k, v in dev.items()}
    a0, r0 = lp.transparency_accumulate(empty, frame, dl)
    assert a0.shape == (0, 4) and r0.shape == (0,)
    with pytest.raises(lighting.MilError, match='-2'):
        lp.transparency_accumulate(dev, frame, np.zeros((101, 8), np.float32))


def test_external_memory_hand_over(lp):
    """SURVEY.md 8 f-3, HIP half: every image of a frame lives in ONE allocation that is exported as a POSIX fd (what
    VK_KHR_external_memory_fd hands out on amdgpu is such a dma-buf) and imported with mil_import_external_memory; the
    passes read the G-buffer and write hdr / ldr / swapchain through the imported mapping.  Results equal the ones on
    ordinary allocations bit for bit, and are visible through the exporting side's mapping (same memory)."""
    w, h = 200, 120
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=24, shadow=128)
    px = w * h
    need = px * (4 * 4 + 16 + 4 + 4)
    try:
        mem_a, ptr_a, fd, size = lp.alloc_exportable_memory(need)
    except lighting.MilError as e:
        if '(hip 801)' in str(e):                               # hipErrorNotSupported: this driver cannot export dma-bufs
            pytest.skip('the HIP driver of this box does not export device memory as a POSIX fd')
        raise
    assert size >= need and fd >= 0
    mem_b, ptr_b = lp.import_external_memory(fd, size)
    os.close(fd)                                                # both sides hold their own reference now
    assert ptr_b != ptr_a
    try:
        off = 0
        def carve(base, n_bytes):
            nonlocal off
            p = base + off
            off += n_bytes
            return p
        gbx = dict(gb)
        for k in ('normal', 'albedo_occ', 'rough_metal', 'depth'):
            t = lp.tensor_at(carve(ptr_b, px * 4), (h, w), gb[k].dtype)
            t.copy_(gb[k])
            gbx[k] = t
        hdr_off = off
        hdr_x = lp.tensor_at(carve(ptr_b, px * 16), (h, w, 4), torch.float32)
        ldr_x = lp.tensor_at(carve(ptr_b, px * 4), (h, w), torch.int32)
        out_x = lp.tensor_at(carve(ptr_b, px * 4), (h, w), torch.int32)
        for flags in (0, lighting.FLAG_STRICT):
            st = bool(flags)
            ref_hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=flags)
            ref_out = lp.fxaa(lp.tonemap(ref_hdr, lighting.OUT_LDR, strict=st), strict=st)
            lp.light_ex(gbx, frame, dl, pl, AMB, sh, flags=flags, out=hdr_x)
            lp.tonemap(hdr_x, lighting.OUT_LDR, out=ldr_x, strict=st)
            lp.fxaa(ldr_x, out=out_x, strict=st)
            assert torch.equal(hdr_x.view(torch.int32), ref_hdr.view(torch.int32)) and torch.equal(out_x, ref_out)
            seen_by_exporter = lp.tensor_at(ptr_a + hdr_off, (h, w, 4), torch.float32)
            assert torch.equal(seen_by_exporter.view(torch.int32), ref_hdr.view(torch.int32))
        torch.cuda.synchronize()
    finally:
        lp.release_external_memory(mem_b)
        lp.release_external_memory(mem_a)
    with pytest.raises(lighting.MilError):
        lp.import_external_memory(-1, 4096)


def test_shade_frame_equals_subpass_chain(lp):
    """mil_shade_frame (one launch) == mil_light_ex(HDR) -> mil_sky_hosek -> mil_transparency_resolve -> mil_tonemap,
    bit for bit in both numeric modes, on the streaming (no point lights) and the tiled kernel, whole frame and strip."""
    from test_oracle_kat import sky_test_params, transparency_test_inputs
    prm = sky_test_params()
    for scene, (w, h), n_point, shadow in [('sweep', (320, 180), 40, 256), ('sweep', (67, 35), 0, 0), ('noise', (96, 64), 12, 0),
                                           ('sweep', (200, 120), 0, 128)]:
        gb, frame, dl, pl, sh = make_scene(scene, w, h, n_point=n_point, shadow=shadow)
        gb['depth'][h // 3, : w // 2] = 1.0                                  # texels at the clear depth whatever their occlusion
        acc, rev = transparency_test_inputs(h, w, seed=w)
        ta, tr = torch.from_numpy(acc).to(DEV), torch.from_numpy(rev).to(DEV)
        for flags in (0, lighting.FLAG_STRICT):
            st = bool(flags & lighting.FLAG_STRICT)
            for use_sky, use_acc in [(True, True), (True, False), (False, True), (False, False)]:
                hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=flags)
                if use_sky: lp.sky_hosek(gb, frame, prm, hdr, strict=st)
                if use_acc: lp.transparency_resolve(ta, tr, hdr, strict=st)
                kw = dict(sky=prm if use_sky else None, accum=ta if use_acc else None, reveal=tr if use_acc else None, flags=flags)
                one = lp.shade_frame(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_HDR, **kw)
                assert torch.equal(one.view(torch.int32), hdr.view(torch.int32)), (scene, w, h, flags, use_sky, use_acc, 'hdr')
                for fmt in (lighting.OUT_LDR, lighting.OUT_SRGB8, lighting.OUT_SRGB8_VIA_LDR):
                    one = lp.shade_frame(gb, frame, dl, pl, AMB, sh, fmt=fmt, **kw)
                    assert torch.equal(one, lp.tonemap(hdr, fmt, strict=st)), (scene, w, h, flags, use_sky, use_acc, fmt)
            # a strip of the frame == those rows of the whole frame
            y0, y1 = 16, 32
            strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
            strip['y_offset'] = y0
            full = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, flags=flags)
            part = lp.shade_frame(strip, frame, dl, pl, AMB, sh, sky=prm, accum=ta[y0:y1], reveal=tr[y0:y1], flags=flags)
            assert torch.equal(part, full[y0:y1])
    # no passes at all == mil_light_ex; bad arguments
    gb, frame, dl, pl, sh = make_scene('sweep', 128, 72, n_point=8)
    assert torch.equal(lp.shade_frame(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_LDR), lp.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_LDR))
    fp = lighting.FramePasses()
    fp.accum = 1
    import ctypes
    out = torch.empty((72, 128), dtype=torch.int32, device=DEV)
    g = lp._gb(gb)
    push = lighting.ubo.push_constants(128, 72, 0)
    fr = np.ascontiguousarray(frame, np.float32).reshape(68)
    assert lp.lib.mil_shade_frame(lp.ctx, ctypes.byref(g), fr.ctypes.data, None, push.ctypes.data, None, ctypes.byref(fp),
                                  out.data_ptr(), 128, 1, 0, None) == -1          # accum without reveal


def test_shade_frame_largest_lds_carve(lp):
    """More than one chunk of point lights (stash in LDS) + LDS-staged shadow rectangle + LDS-staged accum / revealage:
    every per-wave LDS region in use at once (56 KiB per workgroup); still == the chain of separate calls."""
    from test_oracle_kat import sky_test_params, transparency_test_inputs
    prm = sky_test_params()
    w, h = 128, 80
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=1100, shadow=128)
    acc, rev = transparency_test_inputs(h, w, seed=11)
    ta, tr = torch.from_numpy(acc).to(DEV), torch.from_numpy(rev).to(DEV)
    for flags in (lighting.FLAG_SHADOW_LDS, lighting.FLAG_SHADOW_LDS | lighting.FLAG_STRICT):
        st = bool(flags & lighting.FLAG_STRICT)
        hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=flags)
        lp.sky_hosek(gb, frame, prm, hdr, strict=st); lp.transparency_resolve(ta, tr, hdr, strict=st)
        one = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_HDR, flags=flags)
        assert torch.equal(one.view(torch.int32), hdr.view(torch.int32)), flags


def test_shade_frame_pitched_and_misaligned_transparency_inputs(lp):
    """accum / revealage as views into larger images: pitch > width, column offsets that leave the revealage pointer only
    2-byte aligned (the LDS-DMA staging needs 4: the kernel must fall back to direct loads), odd widths, tiny frames."""
    from test_oracle_kat import sky_test_params, transparency_test_inputs
    prm = sky_test_params()
    for (w, h), off in [((160, 96), 0), ((160, 96), 2), ((160, 96), 3), ((67, 35), 1), ((2, 3), 0), ((1, 1), 5), ((18, 17), 4)]:
        gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=20, shadow=64)
        acc, rev = transparency_test_inputs(h, w, seed=w + off)
        pad = 12 if off == 2 else 11                     # off 2 / pad 12: even pitch, 4-byte aligned views -> the LDS-DMA path with pitch > width
        big_a = torch.zeros((h, w + pad, 4), dtype=torch.float16, device=DEV)
        big_r = torch.ones((h, w + pad), dtype=torch.float16, device=DEV)
        big_a[:, off:off + w] = torch.from_numpy(acc).to(DEV)
        big_r[:, off:off + w] = torch.from_numpy(rev).to(DEV)
        va, vr = big_a[:, off:off + w], big_r[:, off:off + w]
        ta, tr = va.contiguous(), vr.contiguous()
        for flags in (0, lighting.FLAG_STRICT):
            ref = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_HDR, flags=flags)
            got = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=va, reveal=vr, fmt=lighting.OUT_HDR, flags=flags)
            assert torch.equal(got.view(torch.int32), ref.view(torch.int32)), (w, h, off, flags)
            hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=flags)
            st = bool(flags & lighting.FLAG_STRICT)
            lp.sky_hosek(gb, frame, prm, hdr, strict=st); lp.transparency_resolve(va, vr, hdr, strict=st)
            assert torch.equal(got.view(torch.int32), hdr.view(torch.int32)), (w, h, off, flags, 'chain')


def test_shade_frame_strict_vs_oracle_chain(lp, coracle):
    """Strict whole-frame launch against the C oracle's subpass chain (shade -> sky -> resolve -> tonemap), and the fast
    launch against the float64 chain."""
    from test_oracle_kat import sky_test_params, transparency_test_inputs
    prm = sky_test_params()
    w, h = 160, 96
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=24, shadow=128)
    acc, rev = transparency_test_inputs(h, w, seed=3)
    ta, tr = torch.from_numpy(acc).to(DEV), torch.from_numpy(rev).to(DEV)
    planes = planes_np(gb)
    lit = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=AMB, shadow=sh_np(sh))
    ref = coracle.sky_hosek(planes[3], frame, prm, lit.copy())
    ref = coracle.transparency_resolve(acc, rev, ref)
    S = lighting.FLAG_STRICT
    got = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_HDR, flags=S).cpu().numpy()
    np.testing.assert_allclose(got, ref, rtol=3e-7, atol=0)                  # sky: exp / pow / acos to their last ulp
    sky = planes[3] >= 1
    assert_bits_equal(got[~sky], ref[~sky], 'frame strict, non-sky texels')
    ldr = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_LDR, flags=S).cpu().numpy().view(np.uint32)
    ref_ldr = coracle.tonemap(ref, 1)
    assert (ldr != ref_ldr).mean() < 2e-3                                    # a last-ulp sky difference may flip a rounding
    assert np.array_equal(ldr[~sky], ref_ldr[~sky])
    r64 = mo.transparency_resolve(acc, rev, mo.sky_hosek(planes[3], frame, prm, lit.astype(np.float64)))
    fast = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_HDR).cpu().numpy()
    bad = np.abs(fast - r64) > 2e-3 + 1e-3 * np.abs(r64)
    assert bad.mean() < 1e-3


def test_shade_frame_golden_fixture(lp):
    """The committed whole-chain fixture: strict mil_shade_frame + mil_fxaa reproduce it (sky texels to the last ulp of
    exp / pow / acos, everything else bit for bit); fast mode within 1 LSB of the swapchain image."""
    z, g = np.load(os.path.join(GOLDEN, 'frame_chain_96x54.npz')), np.load(os.path.join(GOLDEN, 'sweep_96x54.npz'))
    gb = {k: torch.from_numpy(g[k].view(np.int32) if g[k].dtype == np.uint32 else g[k]).to(DEV) for k in ('normal', 'albedo_occ', 'rough_metal', 'depth')}
    sh = dict(light_view_proj=g['shadow_lvp'], map=torch.from_numpy(g['shadow_map']).to(DEV), bias=float(g['shadow_bias']))
    ta, tr = torch.from_numpy(z['accum']).to(DEV), torch.from_numpy(z['reveal']).to(DEV)
    args = (gb, g['frame'], g['dir_lights'], g['point_lights'], g['ambient'], sh)
    kw = dict(sky=z['sky_params'], accum=ta, reveal=tr)
    S = lighting.FLAG_STRICT
    hdr = lp.shade_frame(*args, fmt=lighting.OUT_HDR, flags=S, **kw).cpu().numpy()
    sky = g['depth'] >= 1
    np.testing.assert_allclose(hdr, z['hdr_c32'], rtol=3e-7, atol=0)
    assert_bits_equal(hdr[~sky], z['hdr_c32'][~sky], 'frame fixture, non-sky texels')
    ldr = lp.shade_frame(*args, fmt=lighting.OUT_LDR, flags=S, **kw)
    assert (ldr.cpu().numpy().view(np.uint32) != z['ldr_c32']).mean() < 2e-3
    np.testing.assert_array_equal(lp.fxaa(torch.from_numpy(z['ldr_c32'].view(np.int32)).to(DEV), strict=True).cpu().numpy().view(np.uint32), z['swapchain_c32'])
    fast = lp.fxaa(lp.shade_frame(*args, fmt=lighting.OUT_LDR, **kw)).cpu().numpy().view(np.uint32)
    d = np.abs(channels(fast) - channels(z['swapchain_c32'])).max(-1)
    assert (d > 1).mean() <= 1e-2                        # R11G11B10F rounding of a fast-mode value may flip a code; FXAA may flip a texel pick


def test_shade_frame_4k_chain(lp):
    """The whole 4K frame: one launch + FXAA == the five-call chain, and the ldr image against the oracle on a band."""
    from test_oracle_kat import sky_test_params
    w, h = 3840, 2160
    prm = sky_test_params()
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=256, shadow=2048)
    acc = torch.zeros((h, w, 4), dtype=torch.float16, device=DEV); rev = torch.ones((h, w), dtype=torch.float16, device=DEV)
    acc[h // 3: 2 * h // 3, w // 4: 3 * w // 4] = torch.tensor([0.5, 0.4, 0.3, 0.8], dtype=torch.float16, device=DEV)
    rev[h // 3: 2 * h // 3, w // 4: 3 * w // 4] = 0.4
    hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh)
    lp.sky_hosek(gb, frame, prm, hdr); lp.transparency_resolve(acc, rev, hdr)
    ldr = lp.tonemap(hdr, lighting.OUT_LDR)
    one = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=acc, reveal=rev, fmt=lighting.OUT_LDR)
    assert torch.equal(one, ldr)
    assert torch.equal(lp.fxaa(one), lp.fxaa(ldr))
    sky = gb['depth'] >= 1
    assert sky.float().mean() > 0.3 and torch.all(hdr[sky][:, :3].sum(-1) > 0)   # the sky half of Scene A is no longer black


# ------------------------------------------------------------------ full-size properties (BASELINE.json sizes)
def test_4k_properties(lp):
    w, h = 3840, 2160
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=256, shadow=2048)
    a = lp.light_ex(gb, frame, dl, pl, AMB, sh)
    b = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL | lighting.FLAG_SHADOW_LDS)
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))             # culled + LDS-staged == brute force + direct
    assert torch.isfinite(a).all() and torch.all(a[..., 3] == 1)
    # strips at 16-row-aligned cuts (2160 = 7*272 + 256)
    cuts = [0] + [272 * i for i in range(1, 8)] + [h]
    for y0, y1 in zip(cuts[:-1], cuts[1:]):
        strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
        strip['y_offset'] = y0
        part = lp.light_ex(strip, frame, dl, pl, AMB, sh)
        assert torch.equal(part.view(torch.int32), a[y0:y1].view(torch.int32))
    # linearity in the light set (HDR, no shadow): shade(l0) + shade(l1) == shade(l0, l1) to rounding
    s0, s1 = lp.light_directional(gb, frame, dl[:1]), lp.light_directional(gb, frame, dl[1:])
    s01 = lp.light_directional(gb, frame, dl)
    err = (s0[..., :3] + s1[..., :3] - s01[..., :3]).abs()
    assert torch.all(err <= 1e-5 * s01[..., :3].abs() + 1e-6)
    # fused packed output == unfused
    ldr = lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_LDR)
    assert torch.equal(ldr, lp.tonemap(a, lighting.OUT_LDR))
    # oracle spot-check on a 16-row band through the sphere wall
    y0, y1 = 1000, 1016
    strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
    strip['y_offset'] = y0
    fast_tolerance_ok(a[y0:y1].cpu().numpy(), planes_np(strip), frame, dl, pl, AMB, sh_np(sh), y_offset=y0,
                      full_height=h, max_bad_frac=1e-3)


def test_8k_4096_lights_properties(lp, coracle):
    """BASELINE.json configs[3] at full size (7680x4320, 4096 point lights, more than one culling chunk per tile):
    culled == brute force bit for bit, strip-sharded (8 strips on 16-row cuts) == monolithic, fused sRGB8 == unfused; and
    against the ORACLES at that size: a 16-row band through the sphere wall of the two-kernel launch within
    north_star's bar of the float64 oracle, the strict launch on the same band == the C oracle bit for bit."""
    from renderer_amd import sharding
    w, h = 7680, 4320
    gb, frame, dl, pl, sh = make_scene('sweep', w, h, n_point=4096, shadow=2048)
    a = lp.light_ex(gb, frame, dl, pl, AMB, sh)                                # k_bin -> k_order -> k_shade (4096 >= the threshold)
    assert torch.isfinite(a).all() and torch.all(a[..., 3] == 1)
    assert lp.workspace_info()['last_items'] > 100_000 and lp.workspace_info()['last_brute_macro_tiles'] == 0
    per_band = (((gb['albedo_occ'] >> 24) & 0xFF) != 0).sum(1).view(-1, 16).sum(1)
    full_band = int(torch.argmax(per_band).item()) * 16                        # a band where (nearly) every pixel is lit: floor + spheres
    for yb0 in (2112, full_band):                                              # 2112: across the sphere wall at the horizon
        yb1 = yb0 + 16
        band = {k: (v[yb0:yb1] if torch.is_tensor(v) else v) for k, v in gb.items()}
        band['y_offset'] = yb0
        planes = planes_np(band)
        kw = dict(point_lights=pl, ambient=AMB, shadow=sh_np(sh), y_offset=yb0, full_height=h)
        r64 = coracle.shade_f64(*planes, frame, dl, **kw)
        got = a[yb0:yb1].cpu().numpy()
        err = np.abs(got[..., :3].astype(np.float64) - r64[..., :3])
        bad = (err > 1e-3 + 1e-4 * np.abs(r64[..., :3])).any(-1)
        lit = ((planes[1] >> 24) != 0).mean()
        print(f'8K / 4096 lights, rows {yb0}..{yb1} of the two-kernel frame vs float64: {int(bad.sum())} of {bad.size} pixels outside the bare bar '
              f'(PCF taps on the bias), max err {err.max():.3g}; {lit:.0%} of the band lit')
        assert lit > (0.9 if yb0 == full_band else 0.15) and bad.sum() <= 16
        strict = lp.light_ex(band, frame, dl, pl, AMB, sh, flags=lighting.FLAG_STRICT).cpu().numpy()
        assert_bits_equal(strict, coracle.shade(*planes, frame, dl, **kw), f'8K band {yb0} strict')
    y0, y1 = 2048, 2048 + 544                                                # brute force on one strip through the sphere wall
    strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
    strip['y_offset'] = y0
    b = lp.light_ex(strip, frame, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL)
    assert torch.equal(b.view(torch.int32), a[y0:y1].view(torch.int32))
    out = torch.empty_like(a)
    for y0, y1 in sharding.strip_bounds(h, 8):
        strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
        strip['y_offset'] = y0
        lp.light_ex(strip, frame, dl, pl, AMB, sh, out=out[y0:y1])
    assert torch.equal(out.view(torch.int32), a.view(torch.int32))
    del out, b
    assert torch.equal(lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8), lp.tonemap(a, lighting.OUT_SRGB8))


def test_strict_kernels_vs_executed_reference_shaders(lp):
    """tests/golden/glsl_reference.npz holds what the reference's own GLSL files produce (executed by oracle/glsl_exec.py
    in the development container).  The strict kernels reproduce it: lighting, tonemap, resolve and FXAA bit for bit
    (NaN == NaN), the sky to the last ulp of exp / pow / acos."""
    ref = np.load(os.path.join(GOLDEN, 'glsl_reference.npz'))

    def same(a, b):
        a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
        return bool(np.all((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))))
    gbs = {}
    for name in ('sweep_96x54', 'noise_64x36'):
        g = np.load(os.path.join(GOLDEN, name + '.npz'))
        gb = {k: torch.from_numpy(g[k].view(np.int32) if g[k].dtype == np.uint32 else g[k]).to(DEV) for k in ('normal', 'albedo_occ', 'rough_metal', 'depth')}
        gbs[name] = (gb, g)
        got = lp.light_directional(gb, g['frame'], g['dir_lights'], strict=True).cpu().numpy()
        assert same(got, ref['hdr_' + name.split('_')[0]]), name
    gb, g = gbs['sweep_96x54']
    assert same(lp.light_directional(gb, g['frame'], g['dir_lights'][:1], strict=True).cpu().numpy(), ref['hdr_sweep_1light'])
    # tonemap: ACESFilm of the shader, packed by the oracle's R11G11B10F encoder (the attachment encode is the driver's)
    from oracle import coracle
    coracle.build()
    ti = torch.from_numpy(ref['tonemap_in'].reshape(-1, 1, 4).copy()).to(DEV)
    mine = np.array([coracle.pack_r11g11b10f(*t[:3]) for t in ref['tonemap_out']], np.uint32)
    np.testing.assert_array_equal(lp.tonemap(ti, lighting.OUT_LDR, strict=True).cpu().numpy().view(np.uint32).reshape(-1), mine)
    # resolve (shader + blend) and sky on the executed lighting output
    hdr = torch.from_numpy(ref['hdr_sweep'].copy()).to(DEV)
    ta, tr = torch.from_numpy(ref['resolve_accum']).to(DEV), torch.from_numpy(ref['resolve_reveal']).to(DEV)
    assert same(lp.transparency_resolve(ta, tr, hdr.clone(), strict=True).cpu().numpy(), ref['resolve_hdr'])
    sky = lp.sky_hosek(gb, g['frame'], ref['sky_params'], hdr.clone(), strict=True).cpu().numpy()
    np.testing.assert_allclose(sky, ref['sky_hdr'], rtol=3e-7, atol=0)
    # transparency accumulate: the fragment shader's two outputs
    fr = frags_to_device({k: ref['accumulate_' + k] for k in ('position_depth', 'normal_opacity', 'albedo_roughness', 'metallic')})
    frame_a = np.zeros(68, np.float32)
    frame_a[64:67] = ref['accumulate_camera']
    acc, rev = lp.transparency_accumulate(fr, frame_a, ref['accumulate_dir_lights'], strict=True)
    assert same(acc.cpu().numpy(), ref['accumulate_accum']) and same(rev.cpu().numpy(), ref['accumulate_reveal'])
    # FXAA: the shader's float colour through the swapchain's sRGB8 encode
    f = np.load(os.path.join(GOLDEN, 'fxaa_101x70.npz'))
    enc = mo.encode_b8g8r8a8_srgb(ref['fxaa_color'][..., :3].astype(np.float64))
    got = lp.fxaa(torch.from_numpy(f['ldr'].view(np.int32)).to(DEV), strict=True).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, enc)


def test_randomised_sweep(lp):
    """Sixty configurations of tools/fuzz_parity.py (random frame size, strip, pitch, light counts, shadow size, flags, output
    format): strict == C oracle, culled == brute force, mil_shade_frame == the chain of calls, FXAA strip == whole-frame rows,
    accumulate == C oracle, all bit for bit; fast within the documented tolerance.  The tool runs thousands of these."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fuzz_parity
    fuzz_parity.coracle.build()
    fails = []
    for seed in range(1000, 1060):
        fails += fuzz_parity.one(lp, seed)
    assert not fails, fails
