"""HosekWilkieParams blocks the REFERENCE would produce (k_sky had only ever seen synthetic coefficients).

Run in the development container from the repo root:
    python tests/golden/make_hosek_golden.py
writes tests/golden/hosek_params.npz.  The coefficient tables are parsed at run time, as numbers, from
/root/reference/renderer/src/render/hosek/dataset.rs (never copied: only the 2 x 11 x vec3 results travel) and evaluated as
make_hosek_wilkie_params does (render/hosek/mod.rs:123-224): quintic Bezier splines in elevation^(1/3), bilinear in
turbidity and ground albedo - in float32, one rounding per Rust operator, left to right; `powi` as compiler-rt's
square-and-multiply (__powisf2: what llvm.powi.f32 lowers to), `powf(1/3)` / `acos` as the correctly rounded float32 of
the double result (a libm may differ by an ulp: unpinned to that).  Two parameter sets:
    default   HosekSky::new (render/hosek/mod.rs:81-83): sun (0, 1, 0), turbidity 1, ground albedo 0
    basic     scenes/basic.rs:373-375: sun = directional light 0 of main.rs:81-85 = normalize(5, 5, 1), turbidity 8,
              ground albedo (1, 0, 0) - the sky the reference actually draws
plus, for each, what the reference's own sky_hosek_frag.glsl (executed by oracle/glsl_exec.py) makes of it on the sky
texels of the sweep fixture, and a float64 evaluation of the block (how much float32 matters: printed)."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
DATASET = '/root/reference/renderer/src/render/hosek/dataset.rs'
SHADERS = '/root/reference/renderer/shaders/'
F = np.float32


def load_tables():
    text = open(DATASET).read()
    tabs = {}
    for m in re.finditer(r'pub static (DATASET_RGB(?:_RAD)?[123]): \[f32; (\d+)\] = \[(.*?)\];', text, re.S):
        body = re.sub(r'//[^\n]*', '', m.group(3))
        vals = np.array([float(v) for v in re.findall(r'[-+]?\d+\.\d+e[-+]\d+', body)], np.float64)
        assert len(vals) == int(m.group(2)), (m.group(1), len(vals))
        tabs[m.group(1)] = vals
    return [tabs[f'DATASET_RGB{i}'] for i in (1, 2, 3)], [tabs[f'DATASET_RGB_RAD{i}'] for i in (1, 2, 3)]


def powi(a, n, T):                     # compiler-rt __powisf2 / __powidf2
    r = T(1.0)
    while True:
        if n & 1:
            r = T(r * a)
        n //= 2
        if n == 0:
            return r
        a = T(a * a)


def evaluate_spline(d, start, stride, v, T):           # mod.rs:124-131: c * (1 - v)^(5-k) * v^k * d[k], left to right
    u = T(T(1.0) - v)
    acc = None
    for k, c in enumerate((1.0, 5.0, 10.0, 10.0, 5.0, 1.0)):
        t = T(c)
        if k < 5:
            t = T(t * powi(u, 5 - k, T))
        if k > 0:
            t = T(t * powi(v, k, T))
        t = T(t * T(d[start + k * stride]))
        acc = t if acc is None else T(acc + t)
    return acc


def evaluate(d, stride, turbidity, albedo, sun_theta, T):      # mod.rs:133-176
    half_pi = T(np.pi / 2)                                         # std::f32::consts::FRAC_PI_2
    e = max(T(T(1.0) - T(sun_theta / half_pi)), T(0.0))
    # .powf(1.0 / 3.0): the exponent is the float32 constant 0.33333334 in the reference
    elevation_k = F(np.float64(e) ** np.float64(F(1.0) / F(3.0))) if T is F else e ** (1.0 / 3.0)
    t0 = min(max(int(turbidity), 1), 10)
    t1 = min(10, t0 + 1)
    tk = min(max(T(T(turbidity) - T(t0)), T(0.0)), T(1.0))
    a0, a1 = 0, stride * 6 * 10
    a0t0 = evaluate_spline(d, a0 + stride * 6 * (t0 - 1), stride, elevation_k, T)
    a1t0 = evaluate_spline(d, a1 + stride * 6 * (t0 - 1), stride, elevation_k, T)
    a0t1 = evaluate_spline(d, a0 + stride * 6 * (t1 - 1), stride, elevation_k, T)
    a1t1 = evaluate_spline(d, a1 + stride * 6 * (t1 - 1), stride, elevation_k, T)
    al, one = T(albedo), T(1.0)
    r = T(T(a0t0 * T(one - al)) * T(one - tk))
    r = T(r + T(T(a1t0 * al) * T(one - tk)))
    r = T(r + T(T(a0t1 * T(one - al)) * tk))
    return T(r + T(T(a1t1 * al) * tk))


def make_params(sun_dir, turbidity, albedo, T=F):              # mod.rs:181-224
    rgb, rad = load_tables()
    sun_dir = np.asarray(sun_dir, F)
    c = min(max(T(sun_dir[1]), T(0.0)), T(1.0))
    sun_theta = T(np.arccos(np.float64(c)))
    prm = np.zeros((11, 4), np.float64)
    for row, start in enumerate((0, 1, 2, 3, 4, 5, 6, 8, 7)):          # a b c d e f g h i: h and i are stored swapped
        for ch in range(3):
            prm[row, ch] = evaluate(rgb[ch][start:], 9, turbidity, albedo[ch], sun_theta, T)
    for ch in range(3):
        prm[9, ch] = evaluate(rad[ch], 1, turbidity, albedo[ch], sun_theta, T)
    prm[10, :3] = sun_dir
    return prm.astype(F) if T is F else prm


def parameter_sets():
    from renderer_amd import ubo
    return {'default': ((0.0, 1.0, 0.0), 1.0, (0.0, 0.0, 0.0)),
            'basic': (tuple(ubo.default_directional_lights()[0, 0:3]), 8.0, (1.0, 0.0, 0.0))}


def main():
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from oracle import glsl_exec
    import make_glsl_golden as mg
    sweep = np.load(os.path.join(OUT, 'sweep_96x54.npz'))
    hdr0, sh_sweep = mg.lighting(sweep)
    out = {}
    depth = sweep['depth'].reshape(-1)
    sky = depth >= 1.0
    sh_sweep.g['gl_FragCoord'] = mg.frag_coord(96, 54)[sky]
    sh_sweep.P = int(sky.sum())
    pos = sh_sweep.call_function('PositionFromDepth', depth[sky].reshape(-1, 1))
    names = ['A', 'B', 'C', 'D', 'E', 'F', 'G', 'H', 'I', 'Z', 'sun_direction']
    for name, (sun, turb, alb) in parameter_sets().items():
        prm = make_params(sun, turb, alb)
        p64 = make_params(sun, turb, alb, np.float64)
        rel = np.abs(prm[:10, :3] - p64[:10, :3]) / np.maximum(np.abs(p64[:10, :3]), 1e-30)
        print(f'{name}: sun {np.asarray(sun, F)}, turbidity {turb}, albedo {alb}; float32 vs float64 evaluation: max relative difference {rel.max():.2e}')
        print(np.array2string(prm[:, :3], precision=6))
        out[f'params_{name}'] = prm
        out[f'params_{name}_f64'] = p64
        sh = glsl_exec.Shader(SHADERS + 'sky_hosek_frag.glsl')
        col = sh.run(int(sky.sum()), uniforms={'frame_matrix_data': mg.frame_uniforms(sweep['frame']),
                                               'params': {n: prm[i, :3] for i, n in enumerate(names)}},
                     varyings={'position': pos})['f_color']
        img = hdr0.reshape(-1, 4).copy()
        img[sky] = col
        out[f'sky_hdr_{name}'] = img.reshape(54, 96, 4)
    np.savez_compressed(os.path.join(OUT, 'hosek_params.npz'), **out)
    print('wrote hosek_params.npz')


if __name__ == '__main__':
    main()
