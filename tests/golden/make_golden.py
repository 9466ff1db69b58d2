"""Generates the committed fixtures from the CPU oracles (the reference has no golden
vectors for this path and cannot be run here).  Run from the repo root:
    python tests/golden/make_golden.py
Each .npz holds the packed G-buffer planes, uniforms, lights, shadow map and the expected
outputs: hdr_f64 (float64 NumPy oracle), hdr_c32 / ldr_c32 / srgb8_via_ldr_c32 (FP32 C
oracle, the bit-exact target of the strict HIP kernels), hdr_dir_only_c32 (reference path
only: directional lights, no extensions).  fxaa_101x70.npz: an R11G11B10F frame and its
FXAA'd B8G8R8A8 sRGB image from both oracles.  frame_chain_96x54.npz: the whole subpass chain of Frame::build on the
sweep scene - lighting, sky, transparency resolve (hdr), tonemap (ldr R11G11B10F), FXAA (B8G8R8A8 sRGB) - with its
sky parameters and accum / revealage inputs, from the FP32 C oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import coracle as co, mil_oracle as mo   # noqa: E402
from renderer_amd import synth, ubo                  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def make(name, gb, cam):
    frame = cam.frame_matrix_data()
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(24, 0x5EED0000 & 0xFFFF)
    sh = synth.shadow_setup(dl[0, 0:3], 128)
    smap = sh['map'].numpy()
    amb = np.array([0.03, 0.03, 0.03], np.float32)
    planes = (gb['normal'], gb['albedo_occ'], gb['rough_metal'], gb['depth'])
    shd = dict(sh, map=smap)
    kw = dict(point_lights=pl, ambient=amb, shadow=shd)
    hdr64 = mo.shade(*planes, frame, dl, **kw)
    hdr32 = co.shade(*planes, frame, dl, **kw)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), normal=planes[0], albedo_occ=planes[1],
                        rough_metal=planes[2], depth=planes[3], frame=frame, dir_lights=dl, point_lights=pl,
                        ambient=amb, shadow_lvp=sh['light_view_proj'], shadow_map=smap,
                        shadow_bias=np.float32(sh['bias']), hdr_f64=hdr64, hdr_c32=hdr32,
                        ldr_c32=co.tonemap(hdr32, 1), srgb8_via_ldr_c32=co.tonemap(hdr32, 3),
                        hdr_dir_only_c32=co.shade(*planes, frame, dl))


if __name__ == '__main__':
    co.build()
    cam = synth.sweep_camera(96, 54)
    make('sweep_96x54', synth.to_numpy(synth.scene_sweep(96, 54, cam)), cam)
    cam = synth.sweep_camera(64, 36)
    make('noise_64x36', synth.to_numpy(synth.scene_noise(64, 36, 0x5EED0001)), cam)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_oracle_kat import fxaa_test_image
    ldr = fxaa_test_image(70, 101, 3)
    np.savez_compressed(os.path.join(OUT, 'fxaa_101x70.npz'), ldr=ldr, srgb8_c32=co.fxaa(ldr), srgb8_f64=mo.fxaa(ldr))
    # the whole chain of hdr-writing subpasses + tonemap + FXAA (mil_shade_frame + mil_fxaa) on the sweep fixture
    from test_oracle_kat import transparency_test_inputs
    z = np.load(os.path.join(OUT, 'sweep_96x54.npz'))
    prm = synth.hosek_params_synthetic()
    acc, rev = transparency_test_inputs(54, 96, seed=96)
    hdr = co.transparency_resolve(acc, rev, co.sky_hosek(z['depth'], z['frame'], prm, z['hdr_c32'].copy()))
    ldr = co.tonemap(hdr, 1)
    np.savez_compressed(os.path.join(OUT, 'frame_chain_96x54.npz'), sky_params=prm, accum=acc, reveal=rev, hdr_c32=hdr,
                        ldr_c32=ldr, swapchain_c32=co.fxaa(ldr))
    print('wrote fixtures to', OUT)
