"""mil_extras.ldr_out: the tonemap subpass of every pixel in the lighting launch (render/mod.rs:201-216 + 297-309,
fs_tonemap.glsl:14-27).  Integer / index bar: the sRGB8 image must equal mil_tonemap of the hdr image the same launch
stored, bit for bit, on every kernel that can serve the call (streaming, tiled, two-kernel, whole-frame, strips, dealt bands),
and the hdr image must not change."""
import numpy as np
import pytest
import torch

from renderer_amd import lighting, synth, ubo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
AMB = (0.03, 0.03, 0.03)


@pytest.fixture(scope='module')
def lp():
    p = lighting.LightingPass(0, max_point_lights=4096)
    yield p
    p.close()


def scene(w, h, n_point, shadow, seed=0x5EED0000):
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(n_point, seed + 1) if n_point else None
    sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow else None
    return gb, cam.frame_matrix_data(), dl, pl, sh


def same(a, b):
    return torch.equal(a.view(torch.int32), b.view(torch.int32))


@pytest.mark.parametrize('w,h,n_point,shadow,flags,two', [(320, 180, 0, 0, 0, False), (320, 180, 0, 128, 0, False), (333, 181, 60, 128, 0, False),
                                                          (320, 180, 60, 64, lighting.FLAG_NO_CULL, False), (320, 180, 60, 64, lighting.FLAG_SHADOW_LDS, False),
                                                          (200, 100, 60, 64, lighting.FLAG_STRICT, False), (320, 180, 300, 128, 0, True), (17, 5, 3, 0, 0, True)])
def test_ldr_out_equals_tonemap_of_the_stored_hdr(lp, w, h, n_point, shadow, flags, two):
    gb, frame, dl, pl, sh = scene(w, h, n_point, shadow)
    lp.set_two_kernel_threshold(1 if two else 0xFFFFFFFF)
    try:
        ref = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=flags)
        ldr = torch.full((h, w + 7), 0x55555555, dtype=torch.int32, device=DEV)
        hdr = lp.light_ex(gb, frame, dl, pl, AMB, sh, flags=flags, ldr_out=ldr[:, 3:3 + w])
        assert same(hdr, ref)
        assert torch.equal(ldr[:, 3:3 + w], lp.tonemap(ref, lighting.OUT_SRGB8, strict=bool(flags & lighting.FLAG_STRICT)))
        assert torch.all(ldr[:, :3] == 0x55555555) and torch.all(ldr[:, 3 + w:] == 0x55555555)
    finally:
        lp.set_two_kernel_threshold(1024)


def test_ldr_out_whole_frame_strips_and_bands(lp):
    w, h = 256, 144
    gb, frame, dl, pl, sh = scene(w, h, 90, 128)
    sky = synth.hosek_params_synthetic()
    ref = lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=sky, fmt=lighting.OUT_HDR)
    want = lp.tonemap(ref, lighting.OUT_SRGB8)
    ldr = torch.zeros((h, w), dtype=torch.int32, device=DEV)
    assert same(lp.shade_frame(gb, frame, dl, pl, AMB, sh, sky=sky, fmt=lighting.OUT_HDR, ldr_out=ldr), ref) and torch.equal(ldr, want)
    ref = lp.light_ex(gb, frame, dl, pl, AMB, sh)
    want = lp.tonemap(ref, lighting.OUT_SRGB8)
    for y0, y1 in ((0, 48), (37, 101)):
        strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
        strip['y_offset'] = y0
        ldr = torch.zeros((y1 - y0, w), dtype=torch.int32, device=DEV)
        assert same(lp.light_ex(strip, frame, dl, pl, AMB, sh, ldr_out=ldr), ref[y0:y1]) and torch.equal(ldr, want[y0:y1])
    # dealt bands, output as the assembled frame: both images take the frame's rows
    full_h, full_l = torch.zeros_like(ref), torch.zeros_like(want)
    for r in range(3):
        rows = torch.from_numpy(lighting.band_rows_of_rank(h, 16, 3, r)).to(DEV)
        part = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
        part['y_offset'] = 0
        lp.shade_bands(part, frame, dl, (16, 3, r), pl, AMB, sh, fmt=lighting.OUT_HDR, out=full_h, out_is_frame=True, ldr_out=full_l)
    assert same(full_h, ref) and torch.equal(full_l, want)


def test_ldr_out_needs_the_hdr_format(lp):
    gb, frame, dl, pl, sh = scene(64, 32, 0, 0)
    ldr = torch.zeros((32, 64), dtype=torch.int32, device=DEV)
    with pytest.raises(lighting.MilError):
        lp.light_ex(gb, frame, dl, None, AMB, None, fmt=lighting.OUT_SRGB8, ldr_out=ldr)
    with pytest.raises(lighting.MilError):
        lp.light_ex(gb, frame, dl, None, AMB, None, ldr_out=torch.zeros((64, 32), dtype=torch.int32, device=DEV))      # pitch < width


def test_ldr_out_through_the_entry_point_that_allocates_nothing(lp):
    gb, frame, dl, pl, sh = scene(200, 120, 150, 64)
    ref = lp.light_ex(gb, frame, dl, pl, AMB, sh)
    want = lp.tonemap(ref, lighting.OUT_SRGB8)
    pld = torch.from_numpy(pl).to(DEV)
    hdr = torch.zeros_like(ref)
    ldr = torch.zeros_like(want)
    ex = lighting.Extras()
    # the mirror's light_ex_noalloc has no ldr_out keyword: the raw call, as a C caller would make it
    import ctypes
    import numpy as np
    from renderer_amd.lighting import _np_ptr
    lights = np.ascontiguousarray(dl, np.float32).reshape(-1, 8)
    fr = np.ascontiguousarray(frame, np.float32).reshape(68)
    push = lp._push(gb, lights.shape[0])
    ex.point_lights, ex.n_point = pld.data_ptr(), pld.shape[0]
    ex.ambient = (ctypes.c_float * 3)(*AMB)
    shs = lighting.Shadow((ctypes.c_float * 16)(*[float(v) for v in sh['light_view_proj']]), sh['map'].data_ptr(), sh['map'].shape[0], float(sh['bias']))
    ex.shadow = ctypes.pointer(shs)
    ex.ldr_out, ex.ldr_out_pitch_px = ldr.data_ptr(), ldr.stride(0)
    g = lp._gb(gb)
    lp._check(lp.lib.mil_light_ex_noalloc(lp.ctx, ctypes.byref(g), _np_ptr(fr), _np_ptr(lights), _np_ptr(push), ctypes.byref(ex),
                                          hdr.data_ptr(), hdr.stride(0) // 4, lighting.OUT_HDR, 0, None))
    torch.cuda.synchronize()
    assert same(hdr, ref) and torch.equal(ldr, want)
