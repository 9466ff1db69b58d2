"""ctypes front-end of oracle/libmil_oracle.so (the FP32 C restatement).

TEST INFRASTRUCTURE ONLY — see the header of oracle/mil_oracle.c.  Loaded by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by
renderer_amd/.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libmil_oracle.so')
_lib = None


def build(force=False):
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ('mil_oracle.c', 'mil_oracle_f64.c', 'Makefile'))
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < newest:
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libmil_oracle.so'], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.milo_pack_r11g11b10f.restype = ctypes.c_uint32
        _lib.milo_pack_r11g11b10f.argtypes = [ctypes.c_float] * 3
        _lib.milo_max_threads.restype = ctypes.c_int
    return _lib


def _p(a, ct):
    return None if a is None else a.ctypes.data_as(ctypes.POINTER(ct))


def max_threads():
    return int(lib().milo_max_threads())


def shade(normal, albedo_occ, rough_metal, depth, frame_ubo, dir_lights, width=None, full_height=None,
          y_offset=0, point_lights=None, ambient=None, shadow=None, nthreads=0, return_pairs=False):
    """Same contract as oracle.mil_oracle.shade; returns (h, w, 4) float32."""
    h, w = depth.shape
    c = np.ascontiguousarray
    normal, albedo_occ, rough_metal = (c(normal, np.uint32), c(albedo_occ, np.uint32), c(rough_metal, np.uint32))
    depth = c(depth, np.float32)
    ubo = c(frame_ubo, np.float32).reshape(68)
    dl = c(dir_lights, np.float32).reshape(-1, 8) if dir_lights is not None else np.zeros((0, 8), np.float32)
    pl = c(point_lights, np.float32).reshape(-1, 8) if point_lights is not None else np.zeros((0, 8), np.float32)
    amb = c(ambient, np.float32) if ambient is not None else None
    lvp = smap = None
    ssize, sbias = 0, 0.0
    if shadow is not None:
        lvp = c(shadow['light_view_proj'], np.float32)
        smap = c(np.asarray(shadow['map']), np.float32)
        ssize, sbias = smap.shape[0], float(shadow['bias'])
    hdr = np.empty((h, w, 4), np.float32)
    pairs = ctypes.c_longlong(0)
    f, u = ctypes.c_float, ctypes.c_uint32
    lib().milo_shade(_p(normal, u), _p(albedo_occ, u), _p(rough_metal, u), _p(depth, f),
                     ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(y_offset),
                     ctypes.c_int(h if full_height is None else full_height),
                     _p(ubo, f), _p(dl, f), ctypes.c_int(dl.shape[0]), _p(pl, f), ctypes.c_int(pl.shape[0]),
                     _p(amb, f), _p(lvp, f), _p(smap, f), ctypes.c_int(ssize), ctypes.c_float(sbias),
                     _p(hdr, f), ctypes.c_int(nthreads), ctypes.byref(pairs))
    if width is not None and width != w:
        raise ValueError('strips are full-width')
    return (hdr, pairs.value) if return_pairs else hdr


def shade_f64(normal, albedo_occ, rough_metal, depth, frame_ubo, dir_lights, full_height=None, y_offset=0,
              point_lights=None, ambient=None, shadow=None, want_cond=False, nthreads=0):
    """oracle.mil_oracle.shade (float64) in C + OpenMP (oracle/mil_oracle_f64.c): whole 4K frames in seconds.
    Returns hdr (h, w, 4) float64 [, cond (h, w, 3) float64]."""
    h, w = depth.shape
    c = np.ascontiguousarray
    normal, albedo_occ, rough_metal = (c(normal, np.uint32), c(albedo_occ, np.uint32), c(rough_metal, np.uint32))
    depth = c(depth, np.float32)
    ubo = c(frame_ubo, np.float32).reshape(68)
    dl = c(dir_lights, np.float32).reshape(-1, 8) if dir_lights is not None else np.zeros((0, 8), np.float32)
    pl = c(point_lights, np.float32).reshape(-1, 8) if point_lights is not None else np.zeros((0, 8), np.float32)
    amb = c(ambient, np.float32) if ambient is not None else None
    lvp = smap = None
    ssize, sbias = 0, 0.0
    if shadow is not None:
        lvp = c(shadow['light_view_proj'], np.float32)
        smap = c(np.asarray(shadow['map']), np.float32)
        ssize, sbias = smap.shape[0], float(shadow['bias'])
    hdr = np.empty((h, w, 4), np.float64)
    cond = np.empty((h, w, 3), np.float64) if want_cond else None
    f, u, d = ctypes.c_float, ctypes.c_uint32, ctypes.c_double
    lib().milo_shade_f64(_p(normal, u), _p(albedo_occ, u), _p(rough_metal, u), _p(depth, f),
                         ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(y_offset),
                         ctypes.c_int(h if full_height is None else full_height),
                         _p(ubo, f), _p(dl, f), ctypes.c_int(dl.shape[0]), _p(pl, f), ctypes.c_int(pl.shape[0]),
                         _p(amb, f), _p(lvp, f), _p(smap, f), ctypes.c_int(ssize), ctypes.c_float(sbias),
                         _p(hdr, d), _p(cond, d), ctypes.c_int(nthreads))
    return (hdr, cond) if want_cond else hdr


def tonemap(hdr, fmt, nthreads=0):
    hdr = np.ascontiguousarray(hdr, np.float32)
    h, w = hdr.shape[:2]
    out = np.empty((h, w), np.uint32)
    lib().milo_tonemap(_p(hdr, ctypes.c_float), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(w),
                       ctypes.c_int(fmt), _p(out, ctypes.c_uint32), ctypes.c_int(nthreads))
    return out


def pack_r11g11b10f(r, g, b):
    return int(lib().milo_pack_r11g11b10f(float(r), float(g), float(b)))


def fxaa(ldr, row_begin=0, row_count=None, nthreads=0):
    """oracle.mil_oracle.fxaa in FP32: (h, w) uint32 R11G11B10F -> (row_count, w) uint32 B8G8R8A8 sRGB."""
    ldr = np.ascontiguousarray(ldr, np.uint32)
    h, w = ldr.shape
    row_count = h - row_begin if row_count is None else row_count
    out = np.empty((row_count, w), np.uint32)
    lib().milo_fxaa(_p(ldr, ctypes.c_uint32), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(row_begin),
                    ctypes.c_int(row_count), _p(out, ctypes.c_uint32), ctypes.c_int(nthreads))
    return out


def sky_hosek(depth, frame_ubo, params, hdr, y_offset=0, full_height=None, nthreads=0):
    """oracle.mil_oracle.sky_hosek in FP32; hdr (h, w, 4) float32 is updated in place and returned."""
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = depth.shape
    ubo = np.ascontiguousarray(frame_ubo, np.float32).reshape(68)
    prm = np.ascontiguousarray(params, np.float32).reshape(44)
    hdr = np.ascontiguousarray(hdr, np.float32)
    f = ctypes.c_float
    lib().milo_sky_hosek(_p(depth, f), ctypes.c_int(w), ctypes.c_int(h), ctypes.c_int(w), ctypes.c_int(y_offset),
                         ctypes.c_int(h if full_height is None else full_height), _p(ubo, f), _p(prm, f), _p(hdr, f),
                         ctypes.c_int(w), ctypes.c_int(nthreads))
    return hdr


def transparency_resolve(accum_f16, reveal_f16, hdr, nthreads=0):
    """oracle.mil_oracle.transparency_resolve in FP32; accum (h, w, 4) float16, reveal (h, w) float16."""
    accum = np.ascontiguousarray(accum_f16, np.float16)
    reveal = np.ascontiguousarray(reveal_f16, np.float16)
    h, w = reveal.shape
    hdr = np.ascontiguousarray(hdr, np.float32)
    u16 = ctypes.c_uint16
    lib().milo_transparency_resolve(accum.view(np.uint16).ctypes.data_as(ctypes.POINTER(u16)), ctypes.c_int(w),
                                    reveal.view(np.uint16).ctypes.data_as(ctypes.POINTER(u16)), ctypes.c_int(w),
                                    ctypes.c_int(w), ctypes.c_int(h), _p(hdr, ctypes.c_float), ctypes.c_int(w),
                                    ctypes.c_int(nthreads))
    return hdr


def transparency_accumulate(pos_z, nrm_op, alb_rough, metallic, camera_position, dir_lights, nthreads=0):
    """oracle.mil_oracle.transparency_accumulate in FP32: four fragment planes ((n, 4) x 3, (n,)) -> accum (n, 4), reveal (n,)."""
    f = ctypes.c_float
    pos_z, nrm_op, alb_rough = (np.ascontiguousarray(a, np.float32).reshape(-1, 4) for a in (pos_z, nrm_op, alb_rough))
    metallic = np.ascontiguousarray(metallic, np.float32).reshape(-1)
    n = metallic.shape[0]
    assert pos_z.shape[0] == nrm_op.shape[0] == alb_rough.shape[0] == n
    cam = np.ascontiguousarray(camera_position, np.float32).reshape(3)
    dl = np.ascontiguousarray(dir_lights, np.float32).reshape(-1, 8)
    accum, reveal = np.empty((n, 4), np.float32), np.empty(n, np.float32)
    lib().milo_transparency_accumulate(_p(pos_z, f), _p(nrm_op, f), _p(alb_rough, f), _p(metallic, f), ctypes.c_int(n), _p(cam, f),
                                       _p(dl, f), ctypes.c_int(dl.shape[0]), _p(accum, f), _p(reveal, f), ctypes.c_int(nthreads))
    return accum, reveal
