"""world_size-2 gloo tests of the N>1 path (strip- and frame-sharding + gather to rank 0).
On CPU there is no HIP kernel, so the per-strip shade function injected here is the C
oracle acting as the stand-in compute; what is under test is the partition, the global-row
bookkeeping and the grouped point-to-point gather of renderer_amd/sharding.py."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from renderer_amd import sharding, synth, ubo

W, H = 64, 80


def test_strip_bounds():
    assert sharding.strip_bounds(2160, 8) == [(272 * i, min(2160, 272 * (i + 1))) for i in range(8)]
    b = sharding.strip_bounds(4320, 8)
    assert [y1 - y0 for y0, y1 in b] == [544] * 7 + [512]
    for h, n in [(2160, 1), (2160, 2), (2160, 4), (1080, 8), (720, 8), (90, 8), (16, 4)]:
        b = sharding.strip_bounds(h, n)
        assert len(b) == n and b[0][0] == 0 and b[-1][1] == h
        assert all(y0 % 16 == 0 for y0, y1 in b if y1 > y0) and all(a[1] == c[0] for a, c in zip(b[:-1], b[1:]))
    assert sharding.frame_assignment(64, 8, 3) == list(range(24, 32))
    assert sum(len(sharding.frame_assignment(10, 4, r)) for r in range(4)) == 10


def test_strip_bounds_weighted():
    rng = np.random.RandomState(7)
    for h, n in [(2160, 8), (4320, 8), (2160, 3), (720, 8), (90, 4), (16, 4)]:
        nb = -(-h // 16)
        for cost in (np.ones(nb), rng.rand(nb) ** 4, np.r_[np.zeros(nb // 2), np.ones(nb - nb // 2)], np.zeros(nb)):
            b = sharding.strip_bounds_weighted(cost, n, h)
            assert len(b) == n and b[0][0] == 0 and b[-1][1] == h
            assert all(y0 % 16 == 0 for y0, y1 in b if y1 > y0) and all(a[1] == c[0] for a, c in zip(b[:-1], b[1:]))
            strip_cost = lambda bb: max(cost[y0 // 16: -(-y1 // 16)].sum() for y0, y1 in bb if y1 > y0)
            assert strip_cost(b) <= strip_cost(sharding.strip_bounds(h, n)) + 1e-9       # never worse than equal rows
    # a frame whose top half is free (sky): the cuts move into the expensive half
    b = sharding.strip_bounds_weighted(np.r_[np.zeros(64), np.ones(71)], 8, 2160)
    assert b[0][1] >= 64 * 16 and max(y1 - y0 for y0, y1 in b[1:]) <= 10 * 16
    s = torch.zeros((3, 5, 4), dtype=torch.int32)
    s[..., 0] = 0x7FFFFFF0
    s[..., 1] = torch.tensor(-(2 ** 31) + 5)          # end wrapped past 2^31: lifetime 21 ticks
    assert sharding.band_costs_from_stamps(s).tolist() == [105.0, 105.0, 105.0]


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import coracle
    cam = synth.sweep_camera(W, H)
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
    pl = synth.point_light_field(12, 5)

    def shade(strip):
        g = synth.to_numpy(strip)
        hdr = coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'], frame, dl,
                            point_lights=pl, y_offset=g['y_offset'], full_height=H, nthreads=1)
        return torch.from_numpy(hdr)

    full = torch.full((H, W, 4), -1.0) if rank == 0 else None
    got = sharding.render_strips(shade, lambda y0, y1: synth.scene_sweep(W, H, cam, rows=(y0, y1)), H, full)
    # the asynchronous form bench.py --mode strips uses: shade, start the gather, wait later
    bounds = sharding.strip_bounds(H, world)
    y0, y1 = bounds[rank]
    full2 = torch.full((H, W, 4), -1.0) if rank == 0 else None
    reqs = sharding.gather_strips(shade(synth.scene_sweep(W, H, cam, rows=(y0, y1))), bounds, full2, dst=0, async_op=True)
    for q in reqs:
        q.wait()
    if rank == 0:
        assert torch.equal(full2, got)
    # frame-parallel: rank r shades frame r of an orbit, rank 0 collects both
    fcam = sharding.orbit_camera(rank, world, W, H)
    g = synth.to_numpy(synth.scene_sweep(W, H, fcam))
    mine = torch.from_numpy(coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'],
                                          fcam.frame_matrix_data(), dl, nthreads=1))
    slots = [torch.zeros(H, W, 4) for _ in range(world)] if rank == 0 else None
    reqs = sharding.gather_frames(mine, slots, async_op=True)
    for q in reqs:
        q.wait()
    # cost-weighted strips: every rank contributes its strip's band costs, all ranks derive the same cuts
    bounds = sharding.strip_bounds(H, world)
    y0, y1 = bounds[rank]
    local_cost = np.arange(y0 // 16, -(-y1 // 16), dtype=np.float64) ** 2 + 1.0    # stand-in for last frame's wave lifetimes
    cost = sharding.exchange_band_costs(local_cost, y0, -(-H // 16))
    assert np.array_equal(cost, np.arange(-(-H // 16), dtype=np.float64) ** 2 + 1.0)
    wb = sharding.strip_bounds_weighted(cost, world, H)
    assert wb[0][1] > bounds[0][1]                                               # the cheap top strip grows
    y0, y1 = wb[rank]
    full3 = torch.full((H, W, 4), -1.0) if rank == 0 else None
    sharding.gather_strips(shade(synth.scene_sweep(W, H, cam, rows=(y0, y1))), wb, full3, dst=0)
    if rank == 0:
        assert torch.equal(full3, got)
        np.savez(out_path, strips=got.numpy(), frames=torch.stack(slots).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_strips_and_frames_gather_world2(tmp_path, coracle):
    out = str(tmp_path / 'r0.npz')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    cam = synth.sweep_camera(W, H)
    g = synth.to_numpy(synth.scene_sweep(W, H, cam))
    dl = ubo.default_directional_lights()
    ref = coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'], cam.frame_matrix_data(), dl,
                        point_lights=synth.point_light_field(12, 5))
    np.testing.assert_array_equal(z['strips'].view(np.uint32), ref.view(np.uint32))   # sharded == monolithic, bit for bit
    for r in range(2):
        fcam = sharding.orbit_camera(r, 2, W, H)
        g = synth.to_numpy(synth.scene_sweep(W, H, fcam))
        ref = coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'], fcam.frame_matrix_data(), dl)
        np.testing.assert_array_equal(z['frames'][r].view(np.uint32), ref.view(np.uint32))
    assert z['frames'][0].max() > 0 and not np.array_equal(z['frames'][0], z['frames'][1])
