"""The two-kernel form of the tiled lighting launch (k_bin -> k_order -> k_shade, csrc/mil_device.hpp), which fast-mode
mil_light_ex takes from MIL_TWO_KERNEL_DEFAULT point lights on, forced here (threshold 1) onto frames the oracle finishes
in seconds.  Bars as everywhere: index / integer work (lists, buckets, order, strips, bands) bit-exact - the result must
equal the brute-force launch and the one-kernel form bit for bit - and the fast arithmetic within 1e-3 + 1e-4 |b| (+ the
conditioning term) of the float64 oracle.  What the shaders are: fs_deferred_lighting.glsl:40-67, inc_brdf.glsl:2-63."""
import os
import sys

import numpy as np
import pytest
import torch

from renderer_amd import lighting, synth, ubo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
AMB = (0.03, 0.03, 0.03)
NEVER = 0xFFFFFFFF


@pytest.fixture(scope='module')
def two():
    p = lighting.LightingPass(0, max_point_lights=4096)
    p.set_two_kernel_threshold(1)
    yield p
    p.close()


@pytest.fixture(scope='module')
def one():
    p = lighting.LightingPass(0, max_point_lights=4096)
    p.set_two_kernel_threshold(NEVER)
    yield p
    p.close()


def scene(kind, w, h, n_point, shadow, seed=0x5EED0000):
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV) if kind == 'sweep' else synth.scene_noise(w, h, seed, device=DEV)
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(n_point, seed + 1)
    sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow else None
    return gb, cam.frame_matrix_data(), dl, pl, sh


def same(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize('kind,w,h,n_point,shadow', [('sweep', 320, 180, 256, 128), ('sweep', 100, 60, 2500, 0), ('noise', 64, 48, 300, 64),
                                                     ('sweep', 130, 77, 9, 32), ('noise', 17, 5, 40, 0), ('sweep', 1, 1, 3, 0),
                                                     ('sweep', 640, 360, 600, 256), ('sweep', 200, 120, 5000, 0)])
def test_two_kernel_equals_one_kernel_and_brute_force(two, one, kind, w, h, n_point, shadow):
    """Every output format; 600 lights take the two-level culling (more than COARSE_MIN), 5000 the coarse list in LDS near
    its capacity; 2500 lights on 100 x 60 put several hundred on every half tile."""
    if n_point > 4096:
        two2, one2 = lighting.LightingPass(0, n_point), lighting.LightingPass(0, n_point)
        two2.set_two_kernel_threshold(1); one2.set_two_kernel_threshold(NEVER)
    else:
        two2, one2 = two, one
    gb, frame, dl, pl, sh = scene(kind, w, h, n_point, shadow)
    for fmt in (lighting.OUT_HDR, lighting.OUT_LDR, lighting.OUT_SRGB8, lighting.OUT_SRGB8_VIA_LDR):
        a = two2.light_ex(gb, frame, dl, pl, AMB, sh, fmt=fmt)
        assert same(a, one2.light_ex(gb, frame, dl, pl, AMB, sh, fmt=fmt)), fmt
        assert same(a, one2.light_ex(gb, frame, dl, pl, AMB, sh, fmt=fmt, flags=lighting.FLAG_NO_CULL)), fmt
    counts = two2.tile_counts(gb)
    two2.light_ex(gb, frame, dl, pl, AMB, sh)
    two2.tile_counts(gb, enable=False)
    c1 = one2.tile_counts(gb)
    one2.light_ex(gb, frame, dl, pl, AMB, sh)
    one2.tile_counts(gb, enable=False)
    # the union of a tile's two half-tile lists never holds more than the one-kernel form's tile list, and is not empty where that is not
    assert torch.all(counts <= c1) and torch.equal(counts > 0, c1 > 0) or n_point > 1024      # (one-kernel lists are capped per round)
    if n_point > 4096:
        two2.close(); one2.close()


@pytest.mark.parametrize('n_point', [1, 2, 3, 62, 63, 64, 65, 66, 127, 128, 129, 130, 191, 193])
def test_list_lengths_at_the_record_loop_boundaries(two, one, n_point):
    """The fast kernels walk a tile's list in blocks of 64 indices, two lights per turn of the loop, the next light's record
    requested (s_load_dwordx8 issued by hand, mil_device.hpp srec_load) while the current one is evaluated.  Every light here
    reaches every pixel, so every tile's and half tile's list holds exactly n_point entries: odd and even lengths, one below /
    at / above a block and two blocks.  Both forms == the brute-force launch bit for bit, on a frame with ragged edges."""
    gb, frame, dl, pl, sh = scene('sweep', 88, 52, n_point, 32, seed=0x5EED0100 + n_point)
    pl = pl.copy()
    pl[:, 3] = 80.0
    pl[:, 7] = 0.02
    brute = one.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL)
    assert same(brute, one.light_ex(gb, frame, dl, pl, AMB, sh))
    assert same(brute, two.light_ex(gb, frame, dl, pl, AMB, sh))
    counts = one.tile_counts(gb)
    one.light_ex(gb, frame, dl, pl, AMB, sh)
    one.tile_counts(gb, enable=False)
    lit_tiles = counts[counts > 0]
    assert lit_tiles.numel() > 0 and int(lit_tiles.min()) == n_point and int(lit_tiles.max()) == n_point


def test_two_kernel_lists_beyond_capacity_fall_back_to_brute_force(two, one, coracle):
    """More lights reach a half tile than a workspace list holds (1024): that half tile is shaded by the brute-force loop,
    still in ascending light order: == the one-kernel form (which takes rounds) bit for bit; strict stays == the C oracle."""
    gb, frame, dl, pl, sh = scene('sweep', 80, 48, 1500, 64)
    pl = pl.copy()
    pl[:, 3] = 60.0
    pl[:, 7] = 0.01
    a = two.light_ex(gb, frame, dl, pl, AMB, sh)
    assert same(a, one.light_ex(gb, frame, dl, pl, AMB, sh))
    assert same(a, one.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_NO_CULL))
    g = synth.to_numpy(gb)
    ref = coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'], frame, dl, point_lights=pl, ambient=AMB,
                        shadow=dict(sh, map=sh['map'].cpu().numpy()))
    strict = two.light_ex(gb, frame, dl, pl, AMB, sh, flags=lighting.FLAG_STRICT).cpu().numpy()
    assert np.array_equal(strict.view(np.uint32), ref.view(np.uint32))


def test_two_kernel_strips_bands_pitch_and_streams(two, one):
    """Strips (y_offset), dealt bands, planes and output as views into wider images, alternating streams (one workspace per
    context: launches on different streams are serialised), frames of changing size (the workspace grows), all == monolithic."""
    w, h = 256, 144
    gb, frame, dl, pl, sh = scene('sweep', w, h, 90, 128)
    full = two.light_ex(gb, frame, dl, pl, AMB, sh)
    for y0, y1 in ((0, 48), (37, 101), (96, 144)):
        strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
        strip['y_offset'] = y0
        assert same(two.light_ex(strip, frame, dl, pl, AMB, sh), full[y0:y1])
    for band, n in ((16, 3), (32, 2)):
        out = torch.zeros_like(full)
        for r in range(n):
            rows = torch.from_numpy(lighting.band_rows_of_rank(h, band, n, r)).to(DEV)
            part = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
            part['y_offset'] = 0
            if r == 0:
                two.shade_bands(part, frame, dl, (band, n, r), pl, AMB, sh, fmt=lighting.OUT_HDR, out=out, out_is_frame=True)
            else:
                out[rows] = two.shade_bands(part, frame, dl, (band, n, r), pl, AMB, sh, fmt=lighting.OUT_HDR)
        assert same(out, full), (band, n)
    big = {k: (torch.nn.functional.pad(v, (3, 9)) if torch.is_tensor(v) else v) for k, v in gb.items()}
    view = {k: (v[:, 3:3 + w] if torch.is_tensor(v) else v) for k, v in big.items()}
    outbuf = torch.full((h, w + 11, 4), -7.0, device=DEV)
    got = two.light_ex(view, frame, dl, pl, AMB, sh, out=outbuf[:, 5:5 + w])
    assert same(got, full) and torch.all(outbuf[:, :5] == -7.0) and torch.all(outbuf[:, 5 + w:] == -7.0)
    streams = [torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)]
    torch.cuda.synchronize()
    outs = [two.light_ex(gb, frame, dl, pl, AMB, sh, stream=streams[i % 2]) for i in range(6)]
    torch.cuda.synchronize()
    assert all(same(o, full) for o in outs)
    # a frame larger than 4K: inside the 7680 x 4320 reservation of mil_create, nothing is allocated; then smaller frames again
    gb2, frame2, dl2, pl2, _ = scene('sweep', 4096, 2304, 64, 0)
    a = two.light_ex(gb2, frame2, dl2, pl2, AMB, None, fmt=lighting.OUT_LDR)
    assert same(a, one.light_ex(gb2, frame2, dl2, pl2, AMB, None, fmt=lighting.OUT_LDR))
    assert same(two.light_ex(gb, frame, dl, pl, AMB, sh), full)
    assert two.workspace_info()['frame_call_allocations'] == 0


def test_two_kernel_workspace_is_compact_and_per_frame_calls_allocate_nothing(one):
    """the workspace was 2 KiB per half tile (133 MB at 4K, 531 MB at 8K, grown by hipFree + hipMalloc inside a
    frame call).  Now: 12 bytes per half tile + a pool of compact lists, reserved at mil_create for 7680 x 4320; no per-frame
    entry point allocates up to that size (the context's own counter and hipMemGetInfo agree); a pool too small for the
    frame's lists sends the macro tiles that found no room to the brute-force loop - same bits; a frame beyond the
    reservation grows the workspace once, inside the call, and is counted."""
    lp = lighting.LightingPass(0, max_point_lights=4096)
    lp.set_two_kernel_threshold(1)
    info = lp.workspace_info()
    assert info['half_tiles'] == 2 * 480 * 270 and info['pool_entries'] == 12 << 20 and info['list_entries_max'] == 960
    assert info['bytes'] <= 30e6, info                                     # 29.3 MB: counters + 16 B x 259 200 + 24 MiB
    gb, frame, dl, pl, sh = scene('sweep', 3840, 2160, 4096, 0)
    out = torch.empty((2160, 3840), dtype=torch.int32, device=DEV)
    lp.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8, out=out)          # stages the lights (ring slot, allocated at mil_create)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        lp.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8, out=out)
    strip = {k: (v[160:1200] if torch.is_tensor(v) else v) for k, v in gb.items()}
    strip['y_offset'] = 160
    lp.light_ex(strip, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8, out=out[160:1200])
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    info = lp.workspace_info()
    assert info['frame_call_allocations'] == 0
    assert 500_000 < info['last_pool_entries'] < 2_400_000, info          # the strip's lists (last launch): compact, a fraction of the pool
    ref = one.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8)
    assert same(out, ref) and info['last_brute_macro_tiles'] == 0
    # pool too small: most macro tiles find no room and take the brute-force loop; the bits do not change
    lp.set_list_pool(200_000)
    got = lp.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8)
    info = lp.workspace_info()
    assert info['pool_entries'] == 200_000 and info['last_brute_macro_tiles'] > 100 and info['frame_call_allocations'] == 0
    assert same(got, ref)
    lp.set_list_pool(12 << 20)
    assert same(lp.light_ex(gb, frame, dl, pl, AMB, None, fmt=lighting.OUT_SRGB8), ref)
    # beyond the reservation (a 16-pixel-wide frame of 2.08 M rows has more half tiles than 7680 x 4320): grows once, counted
    h = 16 * 130_001
    g = torch.Generator(device=DEV).manual_seed(1)
    tall = dict(normal=torch.randint(0, 1 << 30, (h, 16), dtype=torch.int32, device=DEV, generator=g),
                albedo_occ=torch.randint(-(1 << 31), (1 << 31) - 1, (h, 16), dtype=torch.int32, device=DEV, generator=g),
                rough_metal=torch.randint(0, 1 << 16, (h, 16), dtype=torch.int32, device=DEV, generator=g),
                depth=torch.rand((h, 16), dtype=torch.float32, device=DEV, generator=g), width=16, height=h, y_offset=0)
    frame_t = synth.sweep_camera(16, h).frame_matrix_data()
    a = lp.light_ex(tall, frame_t, dl, pl[:64], AMB, None, fmt=lighting.OUT_LDR)
    assert lp.workspace_info()['frame_call_allocations'] == 1 and lp.workspace_info()['half_tiles'] == 2 * 130_001
    assert same(a, one.light_ex(tall, frame_t, dl, pl[:64], AMB, None, fmt=lighting.OUT_LDR))
    a = lp.light_ex(tall, frame_t, dl, pl[:64], AMB, None, fmt=lighting.OUT_LDR)
    assert lp.workspace_info()['frame_call_allocations'] == 1
    lp.reserve(7680, 4320)                                                 # already held: a no-op
    lp.close()


def test_two_kernel_fast_tolerance_and_noalloc(two, coracle):
    gb, frame, dl, pl, sh = scene('sweep', 320, 180, 200, 0)
    g = synth.to_numpy(gb)
    planes = (g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'])
    r64 = coracle.shade_f64(*planes, frame, dl, point_lights=pl, ambient=AMB)
    fast = two.light_ex(gb, frame, dl, pl, AMB, None).cpu().numpy()
    err = np.abs(fast[..., :3] - r64[..., :3])
    assert np.all(err <= 1e-3 + 1e-4 * np.abs(r64[..., :3])) and np.all(fast[..., 3] == 1.0)      # north_star's bar, bare
    # the entry point that allocates nothing takes the two kernels too (its workspace was reserved at mil_create)
    a = two.light_ex_noalloc(gb, frame, dl, torch.from_numpy(pl).to(DEV), AMB, None)
    assert same(a, two.light_ex(gb, frame, dl, pl, AMB, None))


def test_two_kernel_randomised_sweep(two):
    """Forty configurations of tools/fuzz_parity.py with every culled fast-mode launch going through k_bin -> k_shade."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import fuzz_parity
    fuzz_parity.coracle.build()
    fails = []
    for seed in range(3000, 3040):
        fails += fuzz_parity.one(two, seed)
    assert not fails, fails


@pytest.mark.parametrize('w,h,n_point,use_sky,use_acc', [(256, 144, 300, True, True), (250, 141, 90, True, False), (256, 144, 90, False, True),
                                                         (64, 40, 1500, True, True), (256, 144, 60, False, False)])
def test_two_kernel_whole_frame(two, one, w, h, n_point, use_sky, use_acc):
    """mil_shade_frame through k_bin -> k_shade<FMT, FRAME>: with a sky or a resolve subpass every half tile of the frame is an
    item (the unlit ones with an empty list) and k_shade's store does sky / resolve / tonemap; == the one-kernel form and == the
    chain of separate calls it folds (tested against that form in test_gpu_parity.py) bit for bit, every format, strips and
    dealt bands with the passes."""
    gb, frame, dl, pl, sh = scene('sweep', w, h, n_point, 64)
    prm = synth.hosek_params_synthetic() if use_sky else None
    ta, tr = synth.transparency_layer(w, h, DEV) if use_acc else (None, None)
    for fmt in (lighting.OUT_HDR, lighting.OUT_LDR, lighting.OUT_SRGB8, lighting.OUT_SRGB8_VIA_LDR):
        a = two.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=fmt)
        assert same(a, one.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=fmt)), fmt
    full = two.shade_frame(gb, frame, dl, pl, AMB, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_LDR)
    y0, y1 = 32, 96
    strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
    strip['y_offset'] = y0
    part = two.shade_frame(strip, frame, dl, pl, AMB, sh, sky=prm, accum=None if ta is None else ta[y0:y1],
                           reveal=None if tr is None else tr[y0:y1], fmt=lighting.OUT_LDR)
    assert same(part, full[y0:y1])
    if h % 16 == 0:
        out = torch.zeros_like(full)
        for r in range(3):
            rows = torch.from_numpy(lighting.band_rows_of_rank(h, 16, 3, r)).to(DEV)
            sub = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
            sub['y_offset'] = 0
            out[rows] = two.shade_bands(sub, frame, dl, (16, 3, r), pl, AMB, sh, sky=prm, accum=None if ta is None else ta[rows].contiguous(),
                                        reveal=None if tr is None else tr[rows].contiguous(), fmt=lighting.OUT_LDR)
        assert same(out, full)
