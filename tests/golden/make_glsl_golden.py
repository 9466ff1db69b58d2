"""Fixtures produced by EXECUTING THE REFERENCE'S OWN SHADER TEXT (oracle/glsl_exec.py, a GLSL-subset interpreter; the
.glsl files are read from /root/reference/renderer/shaders at run time and never copied).  Run in the development
container from the repo root:
    python tests/golden/make_glsl_golden.py
writes tests/golden/glsl_reference.npz: outputs of
    fs_deferred_lighting.glsl (+ inc_brdf.glsl, inc_structs.glsl)   on the inputs of sweep_96x54.npz / noise_64x36.npz
    fs_tonemap.glsl                                                on a set of hdr samples
    fs_mcguire13_resolve.glsl (+ the blend of mcguire13/mod.rs:108-120, applied here in float32)
    sky_hosek_frag.glsl                                            on the sky texels of the sweep fixture
    fs_fxaa.glsl                                                   on the ldr frame of fxaa_101x70.npz
    fs_mcguire13_accumulation.glsl (+ inc_brdf.glsl)               on 600 synthetic fragments (material factors x texels)
What stays outside the reference's source - texel format decode/encode, built-in precision, blending - is done by this
script the way the oracles do it (see oracle/glsl_exec.py) and remains unpinned."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import glsl_exec, mil_oracle as mo          # noqa: E402
from renderer_amd import synth                          # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SHADERS = '/root/reference/renderer/shaders/'
F32 = np.float32


def unorm(u, bits):
    return (u.astype(F32) / F32((1 << bits) - 1)).astype(F32)


def frame_uniforms(frame):
    f = np.asarray(frame, F32).reshape(-1)
    return dict(view=f[0:16], projection=f[16:32], invProjection=f[32:48], invView=f[48:64], cameraPosition=f[64:67])


def frag_coord(w, h, y_offset=0):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([xx.reshape(-1) + 0.5, yy.reshape(-1) + y_offset + 0.5, np.zeros(w * h), np.ones(w * h)], 1).astype(F32)


def lighting(z, n_lights=None, rows=None):
    """fs_deferred_lighting.glsl on the planes of a committed fixture (rows = (y0, y1): only that strip's fragments, with
    their global gl_FragCoord and the full-frame resolution); returns (hdr, shader instance)."""
    full_h, w = z['depth'].shape
    y0, y1 = rows or (0, full_h)
    h = y1 - y0
    z = {k: (z[k][y0:y1] if k in ('normal', 'albedo_occ', 'rough_metal', 'depth') else z[k]) for k in z.files}
    n, a, r = (z[k].reshape(-1) for k in ('normal', 'albedo_occ', 'rough_metal'))
    inputs = {
        'normal_l_model': np.stack([unorm(n & 0x3FF, 10), unorm((n >> 10) & 0x3FF, 10), unorm((n >> 20) & 0x3FF, 10), unorm(n >> 30, 2)], 1),
        'albedo_occlusion': np.stack([unorm((a >> s) & 0xFF, 8) for s in (0, 8, 16, 24)], 1),
        'roughness_metallic': np.stack([unorm((r >> s) & 0xFF, 8) for s in (0, 8, 16, 24)], 1),
        'depth': np.stack([z['depth'].reshape(-1)] + [np.zeros(w * h, F32)] * 3, 1),
    }
    dl = z['dir_lights'] if n_lights is None else z['dir_lights'][:n_lights]
    uni = {'frame_matrix_data': frame_uniforms(z['frame']),
           'lights_ubo': dict(lights=[dict(direction=l[0:3], intensity=l[3], color=l[4:7]) for l in dl]),
           'push_constants': dict(resolution=np.array([w, full_h], F32), light_count=len(dl))}
    sh = glsl_exec.Shader(SHADERS + 'fs_deferred_lighting.glsl')
    hdr = sh.run(w * h, uniforms=uni, frag_coord=frag_coord(w, h, y0), inputs=inputs)['hdr'].reshape(h, w, 4)
    return hdr, sh


def main():
    from test_oracle_kat import transparency_test_inputs
    out = {}
    sweep, noise = np.load(os.path.join(OUT, 'sweep_96x54.npz')), np.load(os.path.join(OUT, 'noise_64x36.npz'))
    # ---- lighting
    out['hdr_sweep'], sh_sweep = lighting(sweep)
    out['hdr_noise'], _ = lighting(noise)
    out['hdr_sweep_1light'], _ = lighting(sweep, 1)
    out['hdr_sweep_0lights'], _ = lighting(sweep, 0)
    out['hdr_sweep_rows16_32'], _ = lighting(sweep, rows=(16, 32))      # a strip: global gl_FragCoord.y, full-frame resolution
    # ---- tonemap: ACESFilm on assorted hdr values (the attachment's R11G11B10F encode is the driver's)
    rng = np.random.RandomState(5)
    hdr_s = np.concatenate([np.array([0, 1e-8, 1e-4, 0.01, 0.18, 0.5, 1, 2, 10, 1e3, 6.5e4, 1e9], F32), (rng.rand(500) ** 4 * 20).astype(F32)])
    hdr_s = np.stack([hdr_s, hdr_s[::-1], np.roll(hdr_s, 7), np.ones_like(hdr_s)], 1)
    sh = glsl_exec.Shader(SHADERS + 'fs_tonemap.glsl')
    out['tonemap_in'] = hdr_s
    out['tonemap_out'] = sh.run(len(hdr_s), inputs={'hdr_buffer': hdr_s})['f_color']
    # ---- transparency resolve: shader output + blend (src * (1 - src.a) + dst * src.a, mcguire13/mod.rs:108-120)
    acc, rev = transparency_test_inputs(54, 96, seed=96)
    P = acc.shape[0] * acc.shape[1]
    sh = glsl_exec.Shader(SHADERS + 'fs_mcguire13_resolve.glsl')
    src = sh.run(P, inputs={'accum_buff': acc.reshape(P, 4).astype(F32),
                            'reveal_buff': np.stack([rev.reshape(P).astype(F32)] + [np.zeros(P, F32)] * 3, 1)})['f_color']
    dst = out['hdr_sweep'].reshape(P, 4)
    a = src[:, 3:4]
    out['resolve_accum'], out['resolve_reveal'] = acc, rev
    out['resolve_src'] = src.reshape(54, 96, 4)
    out['resolve_hdr'] = ((src * (F32(1.0) - a)).astype(F32) + (dst * a).astype(F32)).astype(F32).reshape(54, 96, 4)
    # ---- sky: fragments are the texels at the clear depth; `position` (the interpolated mesh position) lies on the pixel's
    # view ray: the world position PositionFromDepth gives (evaluated by the lighting shader's own function)
    prm = synth.hosek_params_synthetic()
    depth = sweep['depth'].reshape(-1)
    sky = depth >= 1.0
    sh_sweep.g['gl_FragCoord'] = frag_coord(96, 54)[sky]
    sh_sweep.P = int(sky.sum())
    pos = sh_sweep.call_function('PositionFromDepth', depth[sky].reshape(-1, 1))
    sh = glsl_exec.Shader(SHADERS + 'sky_hosek_frag.glsl')
    names = ['A', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'Z', 'sun_direction']
    col = sh.run(int(sky.sum()), uniforms={'frame_matrix_data': frame_uniforms(sweep['frame']),
                                           'params': {n: prm[i, :3] for i, n in enumerate(names)}},
                 varyings={'position': pos})['f_color']
    hdr_sky = out['hdr_sweep'].reshape(-1, 4).copy()
    hdr_sky[sky] = col
    out['sky_params'] = prm
    out['sky_hdr'] = hdr_sky.reshape(54, 96, 4)
    # ---- FXAA: sampler = NEAREST / CLAMP_TO_EDGE over the decoded R11G11B10F frame (render/fxaa/mod.rs:76-88)
    ldr = np.load(os.path.join(OUT, 'fxaa_101x70.npz'))['ldr']
    h, w = ldr.shape
    img = np.concatenate([mo.decode_r11g11b10f(ldr).astype(F32), np.ones((h, w, 1), F32)], -1)

    def tex(uv):
        ix = np.clip(np.floor(uv[:, 0] * F32(w)).astype(np.int64), 0, w - 1)
        iy = np.clip(np.floor(uv[:, 1] * F32(h)).astype(np.int64), 0, h - 1)
        return img[iy, ix]
    sh = glsl_exec.Shader(SHADERS + 'fs_fxaa.glsl')
    col = sh.run(w * h, uniforms={'push_constants': dict(resolution=np.array([w, h], F32))}, frag_coord=frag_coord(w, h),
                 samplers={'tex': tex})['f_color']
    out['fxaa_color'] = col.reshape(h, w, 4)
    # ---- transparency accumulate: per-fragment varyings, texels and material factors in; the planes of
    # mil_transparent_fragments are the FP32 products the shader forms on its first lines (:91-96)
    P = 600
    fr = synth.transparent_fragments(P, seed=13)
    rng = np.random.RandomState(14)
    mat = dict(albedo_color=np.array([0.9, 0.8, 0.7], F32), alpha_cutoff=0.5, roughness=F32(0.85), metallic=F32(0.6), opacity=F32(0.75), ior=1.0)
    texel = {k: rng.rand(P, 4).astype(F32) for k in ('albedo_map', 'roughness_map', 'metallic_map', 'opacity_map')}
    texel['roughness_map'][:, 0] = F32(0.03) + F32(0.97) * texel['roughness_map'][:, 0]
    other = rng.rand(P, 4).astype(F32)
    samplers = {k: (lambda uv, v=v: v) for k, v in texel.items()}
    samplers.update({k: (lambda uv: other) for k in ('normal_map', 'displacement_map', 'occlusion_map')})
    dl = np.concatenate([sweep['dir_lights'], np.array([[-0.3, 0.5, 0.81, 3.0, 0.2, 0.9, 0.5, 0.0]], F32)])
    cam = np.asarray(sweep['frame'], F32).reshape(-1)[64:67]
    sh = glsl_exec.Shader(SHADERS + 'fs_mcguire13_accumulation.glsl')
    fc = np.concatenate([rng.rand(P, 2) * 90, fr['position_depth'][:, 3:4], np.ones((P, 1))], 1).astype(F32)
    res = sh.run(P, uniforms={'frame_matrix_data': frame_uniforms(sweep['frame']),
                              'lights_ubo': dict(lights=[dict(direction=l[0:3], intensity=l[3], color=l[4:7]) for l in dl]),
                              'push_constants': dict(resolution=np.array([96, 54], F32), light_count=len(dl)),
                              'TheBlock': dict(material_data=mat)},
                 frag_coord=fc, samplers=samplers,
                 varyings={'in_uv': rng.rand(P, 2), 'in_tbn': np.zeros((P, 9)), 'in_wsPosition': fr['position_depth'][:, :3],
                           'in_normal': fr['normal_opacity'][:, :3]})
    out['accumulate_position_depth'] = fr['position_depth']
    out['accumulate_normal_opacity'] = np.concatenate([fr['normal_opacity'][:, :3], (mat['opacity'] * texel['opacity_map'][:, :1]).astype(F32)], 1)
    out['accumulate_albedo_roughness'] = np.concatenate([(mat['albedo_color'] * texel['albedo_map'][:, :3]).astype(F32),
                                                         (mat['roughness'] * texel['roughness_map'][:, :1]).astype(F32)], 1)
    out['accumulate_metallic'] = (mat['metallic'] * texel['metallic_map'][:, 0]).astype(F32)
    out['accumulate_camera'], out['accumulate_dir_lights'] = cam, dl
    out['accumulate_accum'], out['accumulate_reveal'] = res['accum'], res['reveal'][:, 0]
    np.savez_compressed(os.path.join(OUT, 'glsl_reference.npz'), **out)
    print('wrote', os.path.join(OUT, 'glsl_reference.npz'), {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
