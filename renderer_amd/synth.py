"""Seeded synthetic inputs for the lighting pass.

Scene A "sweep" restates /root/reference/renderer/src/scenes/roughness_test.rs:62-116
analytically (floor plane y=0 scaled 50, 10x10 spheres of radius 0.5 at
(0, 3+m, r) with roughness r/10+0.01, metallic m/10+0.01, albedo (0.8,0.4,0.3))
and ray-casts it through a ``PerspectiveCamera`` (camera.rs:75-88) into the
reference's G-buffer formats (render/pbr/mod.rs:150-160, written by
shaders/fs_deferred_geometry.glsl:47-49).  Background pixels keep the render
pass's clear values (render/mod.rs:129-132).  Scene B "noise" draws every plane
i.i.d.  Everything is torch so the same code fills HBM-resident planes on the
GPU box and small CPU tensors in the tests; packed planes are int32 tensors
holding the u32 bit patterns.
"""
import math
import numpy as np
import torch

from . import ubo

SPHERE_RADIUS = 0.5
FLOOR_HALF = 50.0
STEPS = 10


def sweep_camera(width, height):
    """main.rs:70-78 with the eye moved so the whole sphere wall is in view and lit
    by light 0 of main.rs:81-85 (the reference's default eye (0,3,0) sits inside
    sphere (m=0, r=0))."""
    return ubo.PerspectiveCamera(position=(9.0, 7.5, 4.5), forward=(-1.0, 0.0, 0.0), up=(0.0, -1.0, 0.0),
                                 fov_deg=90.0, aspect_ratio=width / height, near=0.05, far=100.0)


def all_lit_camera(width, height):
    """The same scene seen from above the sphere wall, pitched 63 degrees down: every pixel hits the floor or a sphere
    (12 % spheres), so no texel keeps the cleared G-buffer and nothing is skipped by an occlusion-0 early-out."""
    f = np.array([-0.5, -1.0, 0.0])
    return ubo.PerspectiveCamera(position=(7.0, 9.0, 4.5), forward=tuple(f / np.linalg.norm(f)), up=(0.0, -1.0, 0.0),
                                 fov_deg=90.0, aspect_ratio=width / height, near=0.05, far=100.0)


def _as_i32(v):
    """int64 holding a u32 bit pattern -> int32 with the same bits."""
    v = v & 0xFFFFFFFF
    return torch.where(v >= 2 ** 31, v - 2 ** 32, v).to(torch.int32)


def _sphere_list():
    out = []
    for r in range(STEPS):
        for m in range(STEPS):
            out.append(((0.0, 3.0 + m, 0.0 + r), r / STEPS + 0.01, m / STEPS + 0.01))
    return out


def _trace(o, d, t_best, hit_id, nrm, rect, spheres_rects):
    """Nearest hit of rays o + t d against floor and spheres. o, d: (h, w, 3) float64
    (either may broadcast).  spheres_rects: per sphere (y0, y1, x0, x1) to restrict work."""
    h, w = t_best.shape
    oy, dy = o[..., 1].expand(h, w), d[..., 1].expand(h, w)
    t = torch.where(dy.abs() > 1e-12, -oy / dy, torch.full_like(dy, -1.0))
    px = o[..., 0] + t * d[..., 0]
    pz = o[..., 2] + t * d[..., 2]
    ok = (t > 1e-6) & (px.abs() <= FLOOR_HALF) & (pz.abs() <= FLOOR_HALF)
    t_best[:] = torch.where(ok, t, t_best)
    hit_id[:] = torch.where(ok, torch.zeros_like(hit_id), hit_id)
    nrm[..., 1] = torch.where(ok, torch.ones_like(t), nrm[..., 1])
    for idx, ((c, _, _), (y0, y1, x0, x1)) in enumerate(zip(_sphere_list(), spheres_rects)):
        if y1 <= y0 or x1 <= x0:
            continue
        cc = torch.tensor(c, dtype=torch.float64, device=t_best.device)
        oo = (o.expand(h, w, 3)[y0:y1, x0:x1] - cc)
        dd = d.expand(h, w, 3)[y0:y1, x0:x1]
        b = (oo * dd).sum(-1)
        disc = b * b - ((oo * oo).sum(-1) - SPHERE_RADIUS ** 2)
        ts = -b - torch.sqrt(disc.clamp_min(0.0))
        tb = t_best[y0:y1, x0:x1]
        ok = (disc > 0) & (ts > 1e-6) & (ts < tb)
        tb[:] = torch.where(ok, ts, tb)
        hid = hit_id[y0:y1, x0:x1]
        hid[:] = torch.where(ok, torch.full_like(hid, idx + 1), hid)
        n = (oo + ts[..., None] * dd) / SPHERE_RADIUS
        nv = nrm[y0:y1, x0:x1]
        nv[:] = torch.where(ok[..., None], n, nv)


def scene_sweep(width, height, camera=None, device='cpu', rows=None):
    """Scene A.  rows=(y0, y1) generates only that strip (global pixel rows)."""
    cam = camera or sweep_camera(width, height)
    y0, y1 = rows if rows is not None else (0, height)
    h = y1 - y0
    f64 = dict(dtype=torch.float64, device=device)
    view = torch.tensor(cam.view_matrix().astype(np.float64), **f64)
    proj = torch.tensor(cam.projection_matrix().astype(np.float64), **f64)
    inv_pv = torch.linalg.inv(proj @ view)
    xs = (torch.arange(width, **f64) + 0.5) / width * 2.0 - 1.0
    ys = (torch.arange(y0, y1, **f64) + 0.5) / height * 2.0 - 1.0
    clip = torch.stack([xs[None, :].expand(h, width), ys[:, None].expand(h, width),
                        torch.ones(h, width, **f64), torch.ones(h, width, **f64)], -1)
    far = clip @ inv_pv.T
    far = far[..., :3] / far[..., 3:4]
    eye = torch.tensor(np.asarray(cam.position, np.float64), **f64)
    d = far - eye
    d = d / d.norm(dim=-1, keepdim=True)
    o = eye.view(1, 1, 3)

    # screen rectangles of the spheres (conservative)
    fwd = np.asarray(cam.forward, np.float64) / np.linalg.norm(cam.forward)
    fcot = 1.0 / math.tan(cam.fov / 2.0)
    pv = (cam.projection_matrix().astype(np.float64) @ cam.view_matrix().astype(np.float64))
    rects = []
    for (c, _, _) in _sphere_list():
        dist = float(np.dot(np.asarray(c) - np.asarray(cam.position, np.float64), fwd))
        if dist <= 2.0 * SPHERE_RADIUS:
            rects.append((0, h, 0, width))
            continue
        q = pv @ np.array([c[0], c[1], c[2], 1.0])
        sx = (q[0] / q[3] * 0.5 + 0.5) * width
        sy = (q[1] / q[3] * 0.5 + 0.5) * height
        rad = SPHERE_RADIUS / (dist - SPHERE_RADIUS) * fcot * (height / 2.0) * 1.6 + 3.0
        rects.append((max(0, int(sy - rad) - y0), min(h, int(sy + rad) + 1 - y0),
                      max(0, int(sx - rad)), min(width, int(sx + rad) + 1)))

    t_best = torch.full((h, width), float('inf'), **f64)
    hit_id = torch.full((h, width), -1, dtype=torch.int64, device=device)
    nrm = torch.zeros(h, width, 3, **f64)
    _trace(o, d, t_best, hit_id, nrm, None, rects)

    hit = hit_id >= 0
    tb = torch.where(hit, t_best, torch.ones_like(t_best))
    P = eye + tb[..., None] * d
    q = torch.cat([P, torch.ones(h, width, 1, **f64)], -1) @ (proj @ view).T
    depth = q[..., 2] / q[..., 3]
    hit = hit & (depth >= 0.0) & (depth <= 1.0)

    mats = torch.tensor([[1.0, 1.0, 1.0, 0.5, 0.0]] +
                        [[0.8, 0.4, 0.3, r, m] for (_, r, m) in _sphere_list()], **f64)
    mat = mats[hit_id.clamp_min(0)]
    q8 = lambda v: torch.round(v.clamp(0, 1) * 255.0).to(torch.int64)
    q10 = lambda v: torch.round(v.clamp(0, 1) * 1023.0).to(torch.int64)
    enc_n = nrm * 0.5 + 0.5                                   # fs_deferred_geometry.glsl:47
    normal = q10(enc_n[..., 0]) | (q10(enc_n[..., 1]) << 10) | (q10(enc_n[..., 2]) << 20)
    albedo_occ = q8(mat[..., 0]) | (q8(mat[..., 1]) << 8) | (q8(mat[..., 2]) << 16) | (255 << 24)
    rough_metal = q8(mat[..., 3]) | (q8(mat[..., 4]) << 8)
    z = torch.zeros_like(normal)
    return dict(normal=_as_i32(torch.where(hit, normal, z)),
                albedo_occ=_as_i32(torch.where(hit, albedo_occ, z)),
                rough_metal=_as_i32(torch.where(hit, rough_metal, z)),
                depth=torch.where(hit, depth, torch.ones_like(depth)).to(torch.float32),
                width=width, height=height, y_offset=y0)


def scene_noise(width, height, seed, device='cpu', rows=None):
    """Scene B: every plane i.i.d. (gbuffer1 alpha = 0, gbuffer3 bytes 2,3 = 0, depth in [0,1))."""
    y0, y1 = rows if rows is not None else (0, height)
    g = torch.Generator(device='cpu')
    g.manual_seed(int(seed))
    full = lambda hi: torch.randint(0, hi, (height, width), generator=g, dtype=torch.int64)
    normal, albedo_occ, rough_metal = full(2 ** 30), full(2 ** 32), full(2 ** 16)
    depth = torch.rand((height, width), generator=g, dtype=torch.float32)
    sl = slice(y0, y1)
    return dict(normal=_as_i32(normal[sl]).to(device), albedo_occ=_as_i32(albedo_occ[sl]).to(device),
                rough_metal=_as_i32(rough_metal[sl]).to(device), depth=depth[sl].contiguous().to(device),
                width=width, height=height, y_offset=y0)


def point_light_field(n, seed, intensity=4.0):
    """EXTENSION E1 lights around Scene A: positions uniform in a box in front of
    the sphere wall, radius uniform [1,4], colour uniform [0.3,1]^3 (colour range as
    engine.rs:80-84)."""
    rng = np.random.RandomState(seed & 0x7FFFFFFF)
    lo = np.array([-1.0, 0.2, -4.0])
    hi = np.array([8.0, 12.0, 14.0])
    pos = lo + rng.rand(n, 3) * (hi - lo)
    radius = 1.0 + 3.0 * rng.rand(n)
    color = 0.3 + 0.7 * rng.rand(n, 3)
    return ubo.point_lights(pos, radius, color, intensity)


def shadow_setup(light_direction, size, device='cpu', center=(0.0, 6.0, 4.5), half_extent=18.0,
                 distance=40.0, near=0.1, far=90.0, bias=2e-3):
    """EXTENSION E3: orthographic depth map of Scene A seen along -light_direction.
    Returns dict(light_view_proj=16 floats column-major, map=(S,S) float32 tensor, bias, size)."""
    L = np.asarray(light_direction, np.float64)
    L = L / np.linalg.norm(L)
    eye = np.asarray(center, np.float64) + L * distance
    view = ubo.look_to_rh(eye, -L, (0.0, 1.0, 0.0)).astype(np.float64)
    lvp = ubo.ortho_zo(half_extent, near, far).astype(np.float64) @ view
    f64 = dict(dtype=torch.float64, device=device)
    s_ax, u_ax = torch.tensor(view[0, :3], **f64), torch.tensor(view[1, :3], **f64)
    fdir = torch.tensor(-L, **f64)
    tc = ((torch.arange(size, **f64) + 0.5) / size * 2.0 - 1.0) * half_extent
    o = torch.tensor(eye, **f64) + tc[None, :, None] * s_ax + tc[:, None, None] * u_ax
    d = fdir.view(1, 1, 3)
    rects = []
    for (c, _, _) in _sphere_list():
        rel = np.asarray(c, np.float64) - eye
        sx = (np.dot(rel, view[0, :3]) / half_extent * 0.5 + 0.5) * size
        sy = (np.dot(rel, view[1, :3]) / half_extent * 0.5 + 0.5) * size
        rad = SPHERE_RADIUS / half_extent * size / 2.0 + 3.0
        rects.append((max(0, int(sy - rad)), min(size, int(sy + rad) + 1),
                      max(0, int(sx - rad)), min(size, int(sx + rad) + 1)))
    t_best = torch.full((size, size), float('inf'), **f64)
    hit_id = torch.full((size, size), -1, dtype=torch.int64, device=device)
    nrm = torch.zeros(size, size, 3, **f64)
    _trace(o, d, t_best, hit_id, nrm, None, rects)
    z = ((t_best - near) / (far - near)).clamp(0.0, 1.0)
    z = torch.where(hit_id >= 0, z, torch.ones_like(z)).to(torch.float32)
    return dict(light_view_proj=ubo.columns(lvp.astype(np.float32)), map=z.contiguous(), bias=float(bias), size=size)


def to_numpy(gb):
    """Planes as the u32 / f32 NumPy arrays the CPU oracle takes."""
    return dict(normal=gb['normal'].cpu().numpy().view(np.uint32),
                albedo_occ=gb['albedo_occ'].cpu().numpy().view(np.uint32),
                rough_metal=gb['rough_metal'].cpu().numpy().view(np.uint32),
                depth=gb['depth'].cpu().numpy(),
                width=gb['width'], height=gb['height'], y_offset=gb['y_offset'])


def hosek_params_synthetic():
    """Synthetic HosekWilkieParams, (11, 4) float32 = the std140 block of render/hosek/mod.rs:222-256 (the model's
    dataset evaluation is CPU-side and out of scope): magnitudes of a turbidity-3 configuration, distinct per channel."""
    prm = np.zeros((11, 4), np.float32)
    prm[:, :3] = [[-1.1, -1.15, -1.2], [-0.2, -0.25, -0.3], [5.0, 4.0, 2.5], [1.0, 1.2, 0.9], [-3.0, -3.5, -2.5], [0.4, 0.3, 0.5],
                  [0.15, 0.2, 0.1], [0.7, 0.6, 0.65], [0.6, 0.7, 0.5], [12.0, 14.0, 18.0], [0.3, 0.8, 0.52]]
    prm[10, :3] /= np.linalg.norm(prm[10, :3])
    return prm


def transparency_layer(w, h, device):
    """accum (h, w, 4) / reveal (h, w) float16 of a weighted-blended OIT pass (render/mcguire13/mod.rs:22-23) with one
    translucent rectangle over the middle sixth of the frame; the rest holds the clear values (0, 1)."""
    import torch
    acc = torch.zeros((h, w, 4), dtype=torch.float16, device=device)
    rev = torch.ones((h, w), dtype=torch.float16, device=device)
    acc[h // 3: 2 * h // 3, w // 4: 3 * w // 4] = torch.tensor([0.5, 0.4, 0.3, 0.8], dtype=torch.float16, device=device)
    rev[h // 3: 2 * h // 3, w // 4: 3 * w // 4] = 0.4
    return acc, rev


def transparent_fragments(n, seed=0):
    """n fragments of a transparency accumulate draw (mil_transparent_fragments; numpy float32): positions in a 8 x 3 x 8
    box around the origin, unnormalised interpolated normals, gl_FragCoord.z in (0.2, 0.999), roughness >= 0.02 (the
    accumulate shader does not clamp it: 0 is NaN in the reference too)."""
    rng = np.random.RandomState(seed)
    f = np.float32
    pos = (rng.rand(n, 3) * np.array([8.0, 3.0, 8.0]) - np.array([4.0, 0.0, 4.0])).astype(f)
    z = (0.2 + 0.799 * rng.rand(n, 1)).astype(f)
    nrm = rng.randn(n, 3)
    nrm = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True) * (0.5 + rng.rand(n, 1))).astype(f)
    opacity = (0.05 + 0.95 * rng.rand(n, 1)).astype(f)
    albedo = rng.rand(n, 3).astype(f)
    rough = (0.02 + 0.98 * rng.rand(n, 1)).astype(f)
    return {'position_depth': np.concatenate([pos, z], 1), 'normal_opacity': np.concatenate([nrm, opacity], 1),
            'albedo_roughness': np.concatenate([albedo, rough], 1), 'metallic': rng.rand(n).astype(f)}
