"""Multi-GPU sharding of the lighting pass: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference is single-GPU (/root/reference/renderer/src/render/vulkan.rs:106-108 picks
one PhysicalDevice); the pass itself is embarrassingly parallel per pixel
(shaders/fs_deferred_lighting.glsl:40-67 reads only its own texel + uniforms), so there is
no exchange during shading and exactly one collective at the end: gather to rank 0.

  strips  one frame cut into horizontal strips on 16-row boundaries (tiles never straddle
          ranks); PositionFromDepth needs the global row, passed as y_offset.
  frames  a batch of frames dealt to ranks; each rank shades whole frames.

RCCL has no gather primitive; the gather is grouped point-to-point (every sender uses its
own xGMI link to GPU 0), expressed as dist.batch_isend_irecv, receiving straight into the
destination image on rank 0 (no staging copy).
"""
import math

import numpy as np
import torch
import torch.distributed as dist

from . import ubo

TILE = 16


def strip_bounds(height, n, align=TILE):
    """n contiguous strips, every cut a multiple of `align` rows: 2160 -> 7x272 + 256,
    4320 -> 7x544 + 512.  Trailing ranks may get an empty strip."""
    rows = int(math.ceil(height / n / align) * align)
    cuts = [min(height, i * rows) for i in range(n + 1)]
    return list(zip(cuts[:-1], cuts[1:]))


def strip_bounds_weighted(band_cost, n, height, align=TILE):
    """n contiguous strips with cuts on multiples of `align` rows, chosen so that the most expensive strip is as cheap
    as possible (cost-weighted boundaries).  band_cost[i] = cost of rows [i*align, (i+1)*align) - e.g. the
    previous frame's wave lifetimes per tile row (band_costs_from_stamps).  Exact for the bottleneck objective: binary
    search on the bottleneck, greedy feasibility.  Every rank computes the same cuts from the same costs."""
    c = np.asarray(band_cost, np.float64)
    nb = int(math.ceil(height / align))
    assert c.shape == (nb,) and n >= 1 and np.all(c >= 0)
    c = c + 1e-9 * (c.sum() + 1.0) / nb            # no zero-cost bands: keeps the cuts spread when costs are all 0

    def cuts_for(limit):
        cuts, acc = [0], 0.0
        for i, v in enumerate(c):
            if acc + v > limit and acc > 0.0:
                cuts.append(i)
                acc = 0.0
            acc += v
        return cuts

    lo, hi = float(c.max()), float(c.sum())
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if len(cuts_for(mid)) <= n:
            hi = mid
        else:
            lo = mid
    cuts = cuts_for(hi)
    cuts += [nb] * (n + 1 - len(cuts))               # fewer strips than ranks: trailing ranks get an empty strip
    rows = [min(height, k * align) for k in cuts]
    return list(zip(rows[:-1], rows[1:]))


def rescale_band_costs(band_cost, bounds, strip_times, align=TILE):
    """Feedback step of the balancing: the bands of every strip are rescaled so that they sum to the time that strip
    actually took (measured by its rank).  Iterating  cuts -> times -> rescale -> cuts  equalises the strip times even
    though wave lifetimes are not additive across strips."""
    c = np.asarray(band_cost, np.float64).copy()
    for (y0, y1), t in zip(bounds, strip_times):
        if y1 > y0:
            b0, b1 = y0 // align, -(-y1 // align)
            tot = c[b0:b1].sum()
            c[b0:b1] = c[b0:b1] * (t / tot) if tot > 0 else t / (b1 - b0)
    return c


def band_costs_from_stamps(stamps):
    """Per tile row: the sum of the wave lifetimes (end - start, 100 MHz ticks, wrap-safe) that a culled launch wrote
    into the (tiles_y, tiles_x, 4) stamp buffer (LightingPass.tile_stamps)."""
    s = stamps.detach().cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    return (((s[..., 1] - s[..., 0]) & 0xFFFFFFFF).sum(1)).astype(np.float64)


def exchange_band_costs(local_cost, y0, n_bands, group=None, device=None):
    """Every rank contributes the band costs of its own strip (which starts at row y0); all ranks get the whole frame's
    vector.  One all_gather of n_bands floats per rank: the only collective of the balancing step."""
    world = dist.get_world_size(group)
    mine = torch.zeros(n_bands, dtype=torch.float64, device=device)
    b0 = y0 // TILE
    lc = torch.as_tensor(np.asarray(local_cost, np.float64), device=device)
    mine[b0:b0 + lc.numel()] = lc
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.stack(parts).sum(0).cpu().numpy()


def frame_assignment(n_frames, world, rank):
    """Blocked deal of frame indices: rank r gets a contiguous run."""
    per = int(math.ceil(n_frames / world))
    return list(range(min(n_frames, rank * per), min(n_frames, (rank + 1) * per)))


def orbit_camera(frame_idx, n_frames, width, height):
    """Animated camera of the frame-parallel config: yaw += 2*pi/n_frames per frame around the
    sphere wall's centre, same projection as main.rs:70-78."""
    a = 2.0 * math.pi * frame_idx / n_frames
    centre = np.array([0.0, 7.5, 4.5])
    eye = centre + 9.0 * np.array([math.cos(a), 0.0, math.sin(a)])
    fwd = centre - eye
    return ubo.PerspectiveCamera(position=eye, forward=fwd / np.linalg.norm(fwd), up=(0.0, -1.0, 0.0),
                                 fov_deg=90.0, aspect_ratio=width / height, near=0.05, far=100.0)


def animated_point_lights(base, frame_idx, n_frames):
    """Lights of the frame-parallel config: every light of `base` ((n, 8) float32 rows
    {position, radius, colour, intensity}) translates sinusoidally, a closed form of the frame index - light k moves
    along x and z with amplitude 1.5 and phase k * golden angle, one period per n_frames."""
    pl = np.array(base, np.float32, copy=True).reshape(-1, 8)
    k = np.arange(pl.shape[0], dtype=np.float64)
    ph = 2.0 * math.pi * frame_idx / n_frames + k * 2.399963229728653
    pl[:, 0] = (pl[:, 0].astype(np.float64) + 1.5 * np.sin(ph)).astype(np.float32)
    pl[:, 2] = (pl[:, 2].astype(np.float64) + 1.5 * np.cos(ph)).astype(np.float32)
    return pl


def gather_strips(local, bounds, full=None, dst=0, group=None, async_op=False):
    """Gather row strips to rank `dst`.  local: this rank's (rows, width, ...) tensor;
    full (rank dst only): (height, width, ...) destination.  Returns the list of pending
    requests when async_op, else waits."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ops = []
    if rank == dst:
        y0, y1 = bounds[dst]
        if y1 > y0:
            full[y0:y1].copy_(local)
        for r in range(world):
            y0, y1 = bounds[r]
            if r != dst and y1 > y0:
                ops.append(dist.P2POp(dist.irecv, full[y0:y1], r, group))
    else:
        y0, y1 = bounds[rank]
        if y1 > y0:
            ops.append(dist.P2POp(dist.isend, local, dst, group))
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if not async_op:
        for q in reqs:
            q.wait()
        return []
    return reqs


def gather_frames(local, slots=None, dst=0, group=None, async_op=False):
    """Gather one finished frame per rank to rank `dst`.  slots (rank dst only): list of
    world tensors shaped like `local`; slots[dst] is filled by a device copy."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ops = []
    if rank == dst:
        slots[dst].copy_(local)
        ops = [dist.P2POp(dist.irecv, slots[r], r, group) for r in range(world) if r != dst]
    else:
        ops = [dist.P2POp(dist.isend, local, dst, group)]
    reqs = dist.batch_isend_irecv(ops) if ops else []
    if not async_op:
        for q in reqs:
            q.wait()
        return []
    return reqs


def render_strips(shade_fn, make_strip, height, full=None, dst=0, group=None):
    """One frame, strip-sharded.  make_strip(y0, y1) -> this rank's G-buffer strip;
    shade_fn(strip) -> (rows, width, ...) tensor.  Returns `full` on rank dst."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = strip_bounds(height, world)
    y0, y1 = bounds[rank]
    local = shade_fn(make_strip(y0, y1)) if y1 > y0 else None
    gather_strips(local, bounds, full, dst, group)
    return full if rank == dst else None
