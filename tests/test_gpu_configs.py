"""BASELINE.json's configurations at their own sizes and light combinations (cfg1, cfg2 and cfg5 were only
reached at reduced size or through random seeds), and how much of the fast-path tolerance the conditioning term carries
on the headline frame.  Checker: oracle/mil_oracle.c (FP32, bit-exact bar for strict mode) and oracle/mil_oracle_f64.c
(float64, the bar for fast mode: |a-b| <= 1e-3 + 1e-4 |b| in linear HDR, <= 1 LSB after tonemap + sRGB8).

What the shaders are: fs_deferred_lighting.glsl:40-67, inc_brdf.glsl:2-63, fs_tonemap.glsl:14-27 (paths relative to
/root/reference/renderer/shaders/); ambient, PCF and point lights are the build-defined extensions of SURVEY.md 7.3.
"""
import numpy as np
import pytest
import torch

from oracle import mil_oracle as mo
from renderer_amd import lighting, sharding, synth, ubo

pytestmark = pytest.mark.gpu
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


def sh_np(sh):
    return None if sh is None else dict(sh, map=sh['map'].cpu().numpy())


def channels(u):
    return np.stack([(u >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(int)


def bits_equal(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))


def hdr_census(gpu, r64, cond, ref32):
    """Pixels of a fast-mode HDR frame outside (a) the BARE bar 1e-3 + 1e-4 |b| of north_star, (b) the bar plus the
    conditioning term the parity tests allow, (c) the bare bar while the reference-order FP32 evaluation (the C oracle:
    what the reference's own shader computes in FP32) is inside it - i.e. failures that are the fast path's own and not
    the FP32 conditioning of the reference's formula (D_GGX at mirror alignment, alpha down to 1e-4)."""
    b = np.abs(r64[..., :3])
    err = np.abs(gpu[..., :3].astype(np.float64) - r64[..., :3])
    bare = (err > 1e-3 + 1e-4 * b).any(-1)
    with_cond = (err > 1e-3 + 1e-4 * b + 8 * 2.0 ** -24 * cond).any(-1)
    ref_bare = (np.abs(ref32[..., :3].astype(np.float64) - r64[..., :3]) > 1e-3 + 1e-4 * b).any(-1)
    return bare, with_cond, bare & ~ref_bare, float(err.max())


def rows_of(gb, y0, y1):
    strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
    strip['y_offset'] = y0
    return strip


def test_cfg1_720p_one_directional_plus_ambient(lp, coracle):
    """configs[0]: 1280x720, ONE directional light (light 0 of main.rs:81-85) + ambient, nothing else: strict == the C
    oracle on the whole frame bit for bit, fast within the bare HDR bar of the float64 oracle on the whole frame."""
    w, h = 1280, 720
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()[:1]
    planes = planes_np(gb)
    ref = coracle.shade(*planes, frame, dl, ambient=AMB)
    strict = lp.light_ex(gb, frame, dl, None, AMB, None, flags=lighting.FLAG_STRICT).cpu().numpy()
    assert bits_equal(strict, ref)
    r64, cond = coracle.shade_f64(*planes, frame, dl, ambient=AMB, want_cond=True)
    fast = lp.light_ex(gb, frame, dl, None, AMB, None).cpu().numpy()
    bare, with_cond, own, emax = hdr_census(fast, r64, cond, ref)
    print(f'cfg1: of {w * h} pixels {int(bare.sum())} are outside the bare 1e-3 + 1e-4|b| (mirror highlights of the roughness-0.01 '
          f'spheres, HDR values up to {r64[..., :3].max():.3g}; max abs err {emax:.3g}); {int(own.sum())} of them where the reference-order '
          f'FP32 evaluation is inside it; {int(with_cond.sum())} outside bar + conditioning term')
    assert with_cond.sum() == 0 and bare.sum() <= 8 and own.sum() <= 2               # measured: 0 / 0
    assert np.all(fast[..., 3] == 1.0)
    # the ambient term is there: the same frame without it is darker wherever albedo * occlusion > 0
    no_amb = lp.light_ex(gb, frame, dl, None, None, None).cpu().numpy()
    lit = ((planes[1] >> 24) != 0) & ((planes[1] & 0xFFFFFF) != 0)
    assert np.all(fast[lit][:, :3].sum(-1) > no_amb[lit][:, :3].sum(-1))
    # and through the tonemap to the 8-bit image: <= 1 LSB
    srgb = lp.light_ex(gb, frame, dl, None, AMB, None, fmt=lighting.OUT_SRGB8).cpu().numpy().view(np.uint32)
    d = np.abs(channels(srgb) - channels(mo.tonemap(r64, mo.OUT_SRGB8))).max(-1)
    assert d.max() <= 1


def test_cfg2_1080p_directional_pcf_2048_fused_tonemap(lp, coracle):
    """configs[1]: 1920x1080, one directional light with the 2048^2 3x3-PCF shadow, no point light (the streaming
    kernel's direct-tap path), fused with the tonemap: strict == C oracle on the whole frame (HDR and packed), fast
    <= 1 LSB after tonemap + sRGB8 except where a PCF tap sits on the bias."""
    w, h = 1920, 1080
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()[:1]
    sh = synth.shadow_setup(dl[0, 0:3], 2048, device=DEV)
    planes = planes_np(gb)
    ref = coracle.shade(*planes, frame, dl, shadow=sh_np(sh))
    S = lighting.FLAG_STRICT
    assert bits_equal(lp.light_ex(gb, frame, dl, None, None, sh, flags=S).cpu().numpy(), ref)
    for fmt in (lighting.OUT_LDR, lighting.OUT_SRGB8_VIA_LDR):
        fused = lp.light_ex(gb, frame, dl, None, None, sh, fmt=fmt, flags=S).cpu().numpy().view(np.uint32)
        assert np.array_equal(fused, coracle.tonemap(ref, fmt)), fmt
    # the frame does contain shadow edges: every PCF level 0/9 .. 9/9 occurs among the lit pixels
    unshadowed = coracle.shade(*planes, frame, dl)
    lit = unshadowed[..., 0] > 1e-4
    ratio = ref[..., 0][lit] / unshadowed[..., 0][lit]
    levels = np.unique(np.round(ratio * 9).astype(int))
    assert set(levels.tolist()) == set(range(10)), levels
    r64 = coracle.shade_f64(*planes, frame, dl, shadow=sh_np(sh))
    fast = lp.light_ex(gb, frame, dl, None, None, sh).cpu().numpy()
    err = np.abs(fast[..., :3] - r64[..., :3])
    bare = (err > 1e-3 + 1e-4 * np.abs(r64[..., :3])).any(-1)
    print(f'cfg2: {int(bare.sum())} of {w * h} pixels outside the bare HDR bar (PCF taps on the bias flip a ninth of light 0)')
    assert bare.mean() <= 1e-3
    srgb = lp.light_ex(gb, frame, dl, None, None, sh, fmt=lighting.OUT_SRGB8).cpu().numpy().view(np.uint32)
    d = np.abs(channels(srgb) - channels(mo.tonemap(r64, mo.OUT_SRGB8))).max(-1)
    assert (d > 1).mean() <= 1e-3 and np.all(d[~bare] <= 1)         # <= 1 LSB wherever no tap flipped
    assert np.array_equal(srgb, lp.tonemap(torch.from_numpy(fast).to(DEV), lighting.OUT_SRGB8).cpu().numpy().view(np.uint32))


def test_cfg5_animated_4k_sequence(lp, coracle):
    """configs[4]: the 64-frame 4K sequence (orbiting camera, lights moving in closed form), shadows + tonemap, through
    mil_shade_frame: ALL 64 frames, each checked on a band through the sphere wall - strict == the C oracle chain bit for
    bit (every frame, 16 rows); fast <= 1 LSB in the packed sRGB8 image (every eighth frame, 32 rows, float64 oracle)."""
    w, h, n_frames = 3840, 2160, 64
    dl = ubo.default_directional_lights()
    sh = synth.shadow_setup(dl[0, 0:3], 2048, device=DEV)
    base = synth.point_light_field(256, 0x5EED0003)
    y0, y1 = 1056, 1088
    worst = 0
    shn = sh_np(sh)
    for i in range(n_frames):
        cam = sharding.orbit_camera(i, n_frames, w, h)
        gb = synth.scene_sweep(w, h, cam, device=DEV)
        frame = cam.frame_matrix_data()
        pl = sharding.animated_point_lights(base, i, n_frames)
        ys0, ys1 = (y0, y1) if i % 8 == 0 else (y0 + 8, y0 + 24)        # strict: every frame; the full check every eighth
        band = rows_of(gb, ys0, ys1)
        ref = coracle.shade(*planes_np(band), frame, dl, point_lights=pl, ambient=AMB, shadow=shn, y_offset=ys0, full_height=h)
        # the strict launch on the band alone (global rows through y_offset): the same pixels as the whole-frame launch
        strict = lp.shade_frame(band, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8_VIA_LDR, flags=lighting.FLAG_STRICT)
        assert np.array_equal(strict.cpu().numpy().view(np.uint32), coracle.tonemap(ref, 3)), f'frame {i} strict'
        if i % 8:
            continue
        whole = lp.shade_frame(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8_VIA_LDR, flags=lighting.FLAG_STRICT)
        assert torch.equal(whole[ys0:ys1], strict), f'frame {i}: strict band launch != rows of the whole-frame launch'
        planes = planes_np(band)
        kw = dict(point_lights=pl, ambient=AMB, shadow=shn, y_offset=y0, full_height=h)
        r64 = coracle.shade_f64(*planes, frame, dl, **kw)
        fast = lp.shade_frame(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8)
        d = np.abs(channels(fast[y0:y1].cpu().numpy().view(np.uint32)) - channels(mo.tonemap(r64, mo.OUT_SRGB8))).max(-1)
        worst = max(worst, int((d > 1).sum()))
        assert (d > 1).mean() <= 1e-3, f'frame {i}: {(d > 1).sum()} pixels off by more than 1 LSB'
        assert torch.all((fast >> 24) & 0xFF == 255)
        lit = ((gb['albedo_occ'] >> 24) & 0xFF) != 0
        assert 0.05 < lit.float().mean().item() < 0.95             # every frame shows the scene and some background
    print(f'cfg5: worst frame has {worst} of {(y1 - y0) * w} band pixels off by more than 1 LSB (PCF tap flips)')


def test_headline_4k_bare_tolerance_census(lp, coracle):
    """How much the conditioning term carries: the 4K headline frame (2 directional + PCF 2048^2 + 256
    point lights + ambient) against the float64 oracle over ALL 8.3 M pixels, counted against the BARE bar
    1e-3 + 1e-4 |b| of north_star, with and without the shadow (whose discrete taps are a separate error class)."""
    w, h = 3840, 2160
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
    pl = synth.point_light_field(256, 0x5EED0003)
    sh = synth.shadow_setup(dl[0, 0:3], 2048, device=DEV)
    planes = planes_np(gb)

    def census(shadow, gpu):
        kw = dict(point_lights=pl, ambient=AMB, shadow=sh_np(shadow))
        r64, cond = coracle.shade_f64(*planes, frame, dl, want_cond=True, **kw)
        ref32 = coracle.shade(*planes, frame, dl, **kw)
        return hdr_census(gpu, r64, cond, ref32) + (channels(mo.tonemap(r64, mo.OUT_SRGB8)),)

    fast = lp.light_ex(gb, frame, dl, pl, AMB, None).cpu().numpy()
    bare, with_cond, own, emax, ref8 = census(None, fast)
    print(f'headline 4K, no shadow: of {w * h} pixels {int(bare.sum())} are outside the bare 1e-3 + 1e-4|b|, {int(own.sum())} of them where the '
          f'reference-order FP32 evaluation is inside it, {int(with_cond.sum())} outside bar + conditioning term (max abs err {emax:.3g})')
    # round 2 measured 532 / 129 / 0 of 8 294 400 here, all of it GGX highlights of the mirror-like spheres: 1 - (N.H)^2 formed
    # from a rounded N.H.  Since round 3 the fast path forms it as |N - H|^2 (one_minus_ndoth2, csrc/mil_device.hpp) and is
    # closer to float64 there than the reference-order FP32 evaluation is: north_star's bare bar is the ceiling again
    assert with_cond.sum() == 0 and bare.sum() <= 32 and own.sum() <= 4            # measured: 3 / 0 / 0 (VERDICT r2 asked for 256 / 16)
    got8 = channels(lp.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8).cpu().numpy().view(np.uint32))
    d8 = np.abs(got8 - ref8).max(-1)
    r64 = coracle.shade_f64(*planes, frame, dl, point_lights=pl, ambient=AMB)
    ref32 = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=AMB)
    for (y, x) in np.argwhere(d8 > 1)[:8]:
        print(f'  > 1 LSB at ({x}, {y}): sRGB8 {got8[y, x]} vs {ref8[y, x]}; hdr fast {fast[y, x, :3]}, FP32 reference order {ref32[y, x, :3]}, f64 {r64[y, x, :3]}')
    print(f'headline 4K, no shadow: {int((d8 > 1).sum())} pixels differ by more than 1 LSB after tonemap + sRGB8, {int((d8 > 0).sum())} by 1 LSB')
    assert (d8 > 1).sum() == 0                                             # <= 1 LSB after 8-bit tonemap, every pixel
    fast = lp.light_ex(gb, frame, dl, pl, AMB, sh).cpu().numpy()
    bare_s, with_cond_s, own_s, emax, ref8 = census(sh, fast)
    print(f'headline 4K, PCF 2048^2: {int(bare_s.sum())} pixels outside the bare bar (PCF taps sitting on the bias flip a ninth of light 0), '
          f'{int(with_cond_s.sum())} with the conditioning term (max abs err {emax:.3g})')
    assert bare_s.mean() <= 2e-4
    got8 = channels(lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8).cpu().numpy().view(np.uint32))
    assert (np.abs(got8 - ref8).max(-1) > 1).mean() <= 2e-4


def test_noise_1080p_fast_census(lp, coracle):
    """Scene B at 1920x1080 (every plane i.i.d. noise - every roughness byte incl. 0, i.e. alpha = 1e-8, normals and
    depths unrelated: the worst case for D_GGX's conditioning): the fast path against the float64 oracle on all 2 M pixels.
    Measured in round 3: 0 pixels outside the bare 1e-3 + 1e-4 |b| with the two directional lights alone (the reference-order
    FP32 evaluation: 100), 5 with 64 point lights added (190), 1 of them the fast path's own; none outside bar + conditioning
    term, nothing non-finite."""
    w, h = 1920, 1080
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_noise(w, h, 0x5EED0001, device=DEV)
    planes = planes_np(gb)
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
    for n_point, ceiling in ((0, 4), (64, 24)):
        pl = synth.point_light_field(n_point, 5) if n_point else None
        r64, cond = coracle.shade_f64(*planes, frame, dl, point_lights=pl, ambient=AMB, want_cond=True)
        ref32 = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=AMB)
        fast = lp.light_ex(gb, frame, dl, pl, AMB, None).cpu().numpy()
        assert np.isfinite(fast).all() and np.all(fast[..., 3] == 1.0)
        fin = np.isfinite(r64[..., :3]).all(-1) & np.isfinite(ref32[..., :3]).all(-1)      # the literal shader is NaN at L = -V exactly
        bare, with_cond, own, emax = hdr_census(fast, r64, cond, ref32)
        ref_bare = (np.abs(ref32[..., :3].astype(np.float64) - r64[..., :3]) > 1e-3 + 1e-4 * np.abs(r64[..., :3])).any(-1)
        print(f'noise 1080p, {n_point} point lights: fast outside the bare bar {int((bare & fin).sum())} (reference-order FP32: {int((ref_bare & fin).sum())}), '
              f'own {int((own & fin).sum())}, with the conditioning term {int((with_cond & fin).sum())}')
        assert (bare & fin).sum() <= ceiling and (with_cond & fin).sum() == 0 and (own & fin).sum() <= 4
        assert (bare & fin).sum() <= (ref_bare & fin).sum()


@pytest.mark.parametrize('strict', [False, True])
def test_dealt_bands_equal_monolithic(lp, strict):
    """mil_shade_bands: the frame's bands dealt cyclically over n ranks (each rank emulated on this one GPU), every
    rank's compacted output put back == the monolithic mil_shade_frame bit for bit - tiled and streaming kernel, with
    the sky / resolve subpasses, compacted output and rank 0 writing straight into the assembled frame."""
    from test_oracle_kat import sky_test_params, transparency_test_inputs
    prm = sky_test_params()
    fl = lighting.FLAG_STRICT if strict else 0
    for (w, h), n_point, shadow, band, n in [((320, 200), 48, 128, 16, 3), ((320, 200), 0, 128, 32, 4), ((130, 77), 20, 0, 16, 8),
                                             ((256, 144), 30, 64, 64, 2), ((64, 40), 5, 0, 16, 1)]:
        cam = synth.sweep_camera(w, h)
        gb = synth.scene_sweep(w, h, cam, device=DEV)
        frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
        pl = synth.point_light_field(n_point, 9) if n_point else None
        sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow else None
        acc, rev = transparency_test_inputs(h, w, seed=w)
        ta, tr = torch.from_numpy(acc).to(DEV), torch.from_numpy(rev).to(DEV)
        for fmt in (lighting.OUT_HDR, lighting.OUT_SRGB8):
            mono = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=fmt, flags=fl)
            full = torch.zeros_like(mono)
            for r in range(n):
                rows = torch.from_numpy(lighting.band_rows_of_rank(h, band, n, r)).to(DEV)
                if rows.numel() == 0:
                    continue
                part = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
                part['y_offset'] = 0
                kw = dict(point_lights=pl, ambient=AMB, shadow=sh, sky=prm, accum=ta[rows].contiguous(), reveal=tr[rows].contiguous(),
                          fmt=fmt, flags=fl)
                if r == 0:          # rank 0 of a gather writes its bands straight into the assembled frame
                    lp.shade_bands(part, frame, dl, (band, n, r), out=full, out_is_frame=True, **kw)
                else:
                    full[rows] = lp.shade_bands(part, frame, dl, (band, n, r), **kw)
            assert torch.equal(full.view(torch.int32), mono.view(torch.int32)), (w, h, n_point, band, n, fmt)
    # bad layouts are refused before anything is enqueued
    part = {k: (v[:16].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
    with pytest.raises(lighting.MilError, match='-1'):
        lp.shade_bands(part, frame, dl, (24, 2, 0))
    with pytest.raises(lighting.MilError, match='-1'):
        lp.shade_bands(part, frame, dl, (16, 2, 2))
