"""CPU census of a candidate fast-path formula (tools/emul/fast_emul.c) on the headline frame or a smaller view of it,
against the float64 oracle and the reference-order FP32 oracle: the numbers test_headline_4k_bare_tolerance_census asserts,
without a GPU.  Usage: python tools/fast_emul.py [W H] [variant ...]   (variant 0 = round-2 formula, 1 = |N - H|^2 form)"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import coracle  # noqa: E402
from oracle import mil_oracle as mo  # noqa: E402
from renderer_amd import synth, ubo  # noqa: E402

SRC = os.path.join(ROOT, 'tools', 'emul', 'fast_emul.c')
SO = os.path.join(ROOT, 'tools', 'emul', 'fast_emul.so')


def lib():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
        subprocess.check_call(['gcc', '-O2', '-fopenmp', '-ffp-contract=off', '-shared', '-fPIC', SRC, '-o', SO, '-lm'])
    return ctypes.CDLL(SO)


def fold(frame, w, h):
    ip = frame[32:48].astype(np.float64).reshape(4, 4).T      # column-major -> M[r, c]
    iv = frame[48:64].astype(np.float64).reshape(4, 4).T
    cam = frame[64:67].astype(np.float64)
    C = np.array([[2.0 / w, 0, 0, 1.0 / w - 1.0], [0, 2.0 / h, 0, 1.0 / h - 1.0], [0, 0, 1, 0], [0, 0, 0, 1]])
    PC = ip @ C
    ivc = iv.copy()
    ivc[:3, 3] -= cam
    A = ivc @ PC
    A[3] = PC[3]
    return A.astype(np.float32).reshape(16)


def emul(planes, frame, dl, pl, amb, variant, perturb):
    h, w = planes[3].shape
    m = fold(frame, w, h)
    dprep = np.zeros((len(dl), 8), np.float32)
    dprep[:, 0:3] = dl[:, 0:3]
    dprep[:, 4:7] = dl[:, 4:7] * dl[:, 3:4]
    pprep = np.zeros((len(pl), 8), np.float32)
    pprep[:, 0:3] = pl[:, 0:3]
    pprep[:, 3] = pl[:, 3] * pl[:, 3]
    pprep[:, 4:7] = pl[:, 4:7] * pl[:, 7:8]
    pprep[:, 7] = np.float32(1.0) / pprep[:, 3]
    out = np.empty((h, w, 4), np.float32)
    P = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
    cam = np.ascontiguousarray(frame[64:67], np.float32)
    amb = np.asarray(amb, np.float32)
    lib().emul_shade(P(planes[0], ctypes.c_uint32), P(planes[1], ctypes.c_uint32), P(planes[2], ctypes.c_uint32), P(planes[3], ctypes.c_float),
                     w, h, P(m, ctypes.c_float), P(cam, ctypes.c_float), P(dprep, ctypes.c_float), len(dl), P(pprep, ctypes.c_float), len(pl),
                     P(amb, ctypes.c_float), P(out, ctypes.c_float), variant, perturb)
    return out


def channels(u):
    return np.stack([(u >> s) & 0xFF for s in (0, 8, 16, 24)], -1).astype(int)


def main():
    args = [int(a) for a in sys.argv[1:]]
    w, h = (args[0], args[1]) if len(args) >= 2 else (3840, 2160)
    variants = args[2:] or [0, 1]
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device='cpu')
    g = synth.to_numpy(gb)
    planes = tuple(np.ascontiguousarray(g[k]) for k in ('normal', 'albedo_occ', 'rough_metal', 'depth'))
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
    pl = synth.point_light_field(256, 0x5EED0003)
    amb = (0.03, 0.03, 0.03)
    r64, cond = coracle.shade_f64(*planes, frame, dl, want_cond=True, point_lights=pl, ambient=amb)
    ref32 = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=amb)
    b = np.abs(r64[..., :3])
    ref_bare = (np.abs(ref32[..., :3].astype(np.float64) - r64[..., :3]) > 1e-3 + 1e-4 * b).any(-1)
    ref8 = channels(mo.tonemap(r64, mo.OUT_SRGB8))
    print(f'{w}x{h}: reference-order FP32 outside the bare bar: {int(ref_bare.sum())}')
    for v in variants:
        for perturb in (0, 1):
            fast = emul(planes, frame, dl, pl, amb, v, perturb)
            err = np.abs(fast[..., :3].astype(np.float64) - r64[..., :3])
            bare = (err > 1e-3 + 1e-4 * b).any(-1)
            with_cond = (err > 1e-3 + 1e-4 * b + 8 * 2.0 ** -24 * cond).any(-1)
            own = bare & ~ref_bare
            d8 = np.abs(channels(mo.tonemap(fast.astype(np.float64), mo.OUT_SRGB8)) - ref8).max(-1)
            print(f'variant {v} perturb {perturb}: bare {int(bare.sum())}  own {int(own.sum())}  with_cond {int(with_cond.sum())}  '
                  f'>1LSB {int((d8 > 1).sum())}  1LSB {int((d8 == 1).sum())}  max err {err.max():.3g}')
            for (y, x) in np.argwhere(own)[:4]:
                print(f'   own ({x},{y}) rough byte {planes[2][y, x] & 0xFF}: fast {fast[y, x, :3]} ref32 {ref32[y, x, :3]} f64 {r64[y, x, :3]}')


if __name__ == '__main__':
    main()
