"""Randomised parity sweep on the GPU (not part of the pytest suite: `python tools/fuzz_parity.py [iterations] [seed]`).
Each iteration draws a frame size, strip, pitch, light counts, shadow size, flags and output format and checks
  * strict mil_light_ex                      == oracle/mil_oracle.c, bit for bit
  * fast culled                              == fast MIL_FLAG_NO_CULL, bit for bit (culling must not change a bit)
  * fast                                     within the documented tolerance of the float64 restatement (small frames only;
                                               with a shadow map a PCF tap on the bias may flip: <= 1e-3 of the pixels or one pixel)
  * mil_shade_frame(fmt)                     == mil_light_ex + mil_sky_hosek + mil_transparency_resolve + mil_tonemap, bit for bit
  * strict sky / resolve / tonemap chain      == the C oracle's chain (sky: to the last ulp of exp / pow / acos)
  * mil_fxaa on a strip                      == the rows of the whole-frame call
  * mil_transparency_accumulate strict       == the C oracle, bit for bit
FUZZ_BIG=1: frames up to 1000 x 500 with 600 - 4000 point lights and shadow maps up to 2048^2 (seconds per iteration: the C
oracle is brute force).  FUZZ_TWO_KERNEL=1: the two-kernel form of the tiled launch at every light count.  Prints one line per failure with the seed that reproduces it; exit status 1 if any."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import coracle, mil_oracle as mo          # noqa: E402
from renderer_amd import lighting, synth, ubo        # noqa: E402
from test_oracle_kat import sky_test_params, transparency_test_inputs   # noqa: E402

DEV = 'cuda:0'


def bits_eq(a, b):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def one(lp, seed):
    rng = np.random.RandomState(seed)
    fails = []
    big = bool(os.environ.get('FUZZ_BIG'))      # larger frames and light counts: tile lists beyond one LDS round, big shadow maps
    w, full_h = (int(rng.randint(300, 1000)), int(rng.randint(100, 500))) if big else (int(rng.randint(1, 300)), int(rng.randint(1, 200)))
    y0 = int(rng.randint(0, full_h)) if rng.rand() < 0.4 else 0
    h = int(rng.randint(1, full_h - y0 + 1)) if y0 or rng.rand() < 0.3 else full_h
    kind = 'noise' if rng.rand() < 0.4 else 'sweep'
    n_dir = int(rng.choice([0, 1, 2, 2, 2, 5, 9]))
    n_point = int(rng.choice([600, 1500, 4000])) if big else int(rng.choice([0, 0, 1, 7, 40, 130, 300]))
    shadow = int(rng.choice([0, 512, 2048])) if big else int(rng.choice([0, 0, 33, 64, 256]))
    amb = (0.03, 0.02, 0.04) if rng.rand() < 0.6 else None
    pad_l, pad_r = (int(rng.randint(0, 9)), int(rng.randint(0, 9))) if rng.rand() < 0.4 else (0, 0)
    cam = synth.sweep_camera(w, full_h)
    frame = cam.frame_matrix_data()
    pitch = w + pad_l + pad_r
    full = synth.scene_sweep(pitch, full_h, synth.sweep_camera(pitch, full_h), device=DEV) if kind == 'sweep' else synth.scene_noise(pitch, full_h, seed, device=DEV)
    gb = {k: (v[y0:y0 + h, pad_l:pad_l + w] if torch.is_tensor(v) else v) for k, v in full.items()}
    gb.update(width=w, height=full_h, y_offset=y0)
    gnp = {k: v.cpu().numpy() for k, v in gb.items() if torch.is_tensor(v)}
    planes = (gnp['normal'].view(np.uint32), gnp['albedo_occ'].view(np.uint32), gnp['rough_metal'].view(np.uint32), gnp['depth'])
    planes = tuple(np.ascontiguousarray(p) for p in planes)
    d = rng.randn(max(n_dir, 1), 3)
    d[:, 1] = np.abs(d[:, 1]) + 0.1
    dl = ubo.directional_lights([(d[i], 0.2 + 2 * rng.rand(), 0.3 + 0.7 * rng.rand(3)) for i in range(n_dir)]) if n_dir else np.zeros((0, 8), np.float32)
    pl = synth.point_light_field(n_point, seed + 1) if n_point else None
    sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow and n_dir else None
    shn = None if sh is None else dict(sh, map=sh['map'].cpu().numpy())
    cfg = f'seed={seed} {kind} {w}x{h} of {full_h} y0={y0} pad={pad_l},{pad_r} dir={n_dir} point={n_point} shadow={shadow if sh else 0} amb={amb is not None}'
    extra = lighting.FLAG_SHADOW_LDS if rng.rand() < 0.3 else 0

    ref = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=amb, shadow=shn, y_offset=y0, full_height=full_h)
    strict = lp.light_ex(gb, frame, dl, pl, amb, sh, flags=lighting.FLAG_STRICT | extra).cpu().numpy()
    if not bits_eq(strict, ref):
        fails.append('strict != C oracle: ' + cfg)
    strict_nc = lp.light_ex(gb, frame, dl, pl, amb, sh, flags=lighting.FLAG_STRICT | lighting.FLAG_NO_CULL).cpu().numpy()
    if not bits_eq(strict_nc, ref):
        fails.append('strict NO_CULL != C oracle: ' + cfg)
    fast = lp.light_ex(gb, frame, dl, pl, amb, sh, flags=extra)
    fast_nc = lp.light_ex(gb, frame, dl, pl, amb, sh, flags=lighting.FLAG_NO_CULL)
    if not torch.equal(fast.view(torch.int32), fast_nc.view(torch.int32)):
        fails.append('fast culled != fast NO_CULL: ' + cfg)
    if w * h * (n_dir + n_point + 1) < 3e6:
        r64, cond = mo.shade(*planes, frame, dl, point_lights=pl, ambient=amb, shadow=shn, want_cond=True, y_offset=y0, full_height=full_h)
        # The pytest scenes hold the conditioning term at 8 ulp x cond; the sweep, whose random frames are rich in hot spots and
        # grazing highlights, flags from 32 (it is what found the cancellation of the unit-length N.H shortcut).
        tol = (1e-3 + 1e-4 * np.abs(r64[..., :3]) + 32 * 2.0 ** -24 * cond) * (2.0 if kind == 'noise' else 1.0)
        f = fast.cpu().numpy()
        fin = np.isfinite(r64[..., :3]).all(-1)
        # outside the tolerance AND further from float64 than the reference-order FP32 result is (next to a point light both
        # FP32 paths inherit the error of the reconstructed position; the fast path's is the smaller one)
        e_fast, e_strict = np.abs(f[..., :3] - r64[..., :3]), np.abs(strict[..., :3] - r64[..., :3])
        bad = (((e_fast > tol) & (e_fast > e_strict)) | ~np.isfinite(f[..., :3])).any(-1) & fin
        # a PCF tap whose depth compare sits on the bias flips between any two evaluation orders (strict vs float64 too): 1/9 steps
        if bad.mean() > (2e-3 if kind == 'noise' else (1e-3 if sh is not None else 0.0)) and not (sh is not None and bad.sum() <= 1):
            fails.append(f'fast outside tolerance on {bad.sum()} px: ' + cfg)
            if os.environ.get('FUZZ_DIAG'):
                for (y, x) in np.argwhere(bad)[:3]:
                    print(f'  px ({x},{y}) fast {f[y, x, :3]} strict {strict[y, x, :3]} f64 {r64[y, x, :3]} tol {tol[y, x]} cond {cond[y, x]}')
                    print(f'     normal {planes[0][y, x]:#x} albedo_occ {planes[1][y, x]:#x} rough_metal {planes[2][y, x]:#x} depth {planes[3][y, x]!r}')

    # whole-frame launch against the chain of separate calls
    prm = sky_test_params()
    acc, rev = transparency_test_inputs(h, w, seed=seed & 0xFFFF)
    ta, tr = torch.from_numpy(acc).to(DEV), torch.from_numpy(rev).to(DEV)
    fmt = int(rng.choice([lighting.OUT_HDR, lighting.OUT_LDR, lighting.OUT_SRGB8, lighting.OUT_SRGB8_VIA_LDR]))
    use_sky, use_acc = bool(rng.rand() < 0.7), bool(rng.rand() < 0.7)
    for flags in (0, lighting.FLAG_STRICT):
        st = bool(flags)
        hdr = lp.light_ex(gb, frame, dl, pl, amb, sh, flags=flags)
        if use_sky: lp.sky_hosek(gb, frame, prm, hdr, strict=st)
        if use_acc: lp.transparency_resolve(ta, tr, hdr, strict=st)
        chain = hdr if fmt == lighting.OUT_HDR else lp.tonemap(hdr, fmt, strict=st)
        onec = lp.shade_frame(gb, frame, dl, pl, amb, sh, sky=prm if use_sky else None, accum=ta if use_acc else None,
                              reveal=tr if use_acc else None, fmt=fmt, flags=flags)
        if not torch.equal(onec.view(torch.int32), chain.view(torch.int32)):
            fails.append(f'shade_frame != chain (fmt {fmt}, strict {st}, sky {use_sky}, resolve {use_acc}): ' + cfg)
        if st:      # the strict chain against the C oracle's chain: sky to the last ulp of exp / pow / acos, the rest bit for bit
            o = ref.copy()
            if use_sky: o = coracle.sky_hosek(planes[3], frame, prm, o, y_offset=y0, full_height=full_h)
            if use_acc: o = coracle.transparency_resolve(acc, rev, o)
            g = hdr.cpu().numpy()
            if not (bits_eq(g, o) if not use_sky else (np.allclose(g, o, rtol=1e-6, atol=0, equal_nan=True))):
                fails.append(f'strict chain != C oracle chain (sky {use_sky}, resolve {use_acc}): ' + cfg)
            if fmt != lighting.OUT_HDR and not np.array_equal(chain.cpu().numpy().view(np.uint32), coracle.tonemap(g, fmt)):
                fails.append(f'strict tonemap fmt {fmt} != C oracle: ' + cfg)

    # FXAA strip == rows of the whole frame
    if h == full_h:
        ldr = lp.tonemap(lp.light_ex(gb, frame, dl, pl, amb, sh), lighting.OUT_LDR)
        for st in (False, True):
            whole = lp.fxaa(ldr, strict=st)
            r0 = int(rng.randint(0, h))
            rc = int(rng.randint(1, h - r0 + 1))
            if not torch.equal(lp.fxaa(ldr, row_begin=r0, row_count=rc, strict=st), whole[r0:r0 + rc]):
                fails.append(f'fxaa strip rows {r0}+{rc} (strict {st}): ' + cfg)
        if w * h <= 20000:
            if not np.array_equal(lp.fxaa(ldr, strict=True).cpu().numpy().view(np.uint32), coracle.fxaa(ldr.cpu().numpy().view(np.uint32))):
                fails.append('fxaa strict != C oracle: ' + cfg)

    # accumulate
    n = int(rng.randint(1, 5000))
    fr = synth.transparent_fragments(n, seed=seed)
    dev = {k: torch.from_numpy(v).to(DEV) for k, v in fr.items()}
    ca, cr = coracle.transparency_accumulate(fr['position_depth'], fr['normal_opacity'], fr['albedo_roughness'], fr['metallic'], frame[64:67], dl)
    sa, sr = lp.transparency_accumulate(dev, frame, dl, strict=True)
    if not (bits_eq(sa.cpu().numpy(), ca) and bits_eq(sr.cpu().numpy(), cr)):
        fails.append(f'accumulate strict != C oracle (n {n}): ' + cfg)
    return fails


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    coracle.build()
    lp = lighting.LightingPass(0, max_point_lights=4096)
    if os.environ.get('FUZZ_TWO_KERNEL'):       # every culled fast-mode launch through k_bin -> k_shade, whatever the light count
        lp.set_two_kernel_threshold(1)
    total = []
    for i in range(iters):
        new = one(lp, seed0 + i)
        for f in new:
            print('FAIL', f, flush=True)
        total += new
        if (i + 1) % 10 == 0:
            print(f'{i + 1}/{iters} iterations, {len(total)} failures', flush=True)
        if len(total) >= 10:
            break
    print('fuzz:', 'OK' if not total else f'{len(total)} failures')
    sys.exit(1 if total else 0)


if __name__ == '__main__':
    main()
