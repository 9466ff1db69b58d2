"""The multi-GPU layer behind the C ABI (include/mil.h mil_mg_*; SURVEY.md 8 b / 8 e) on ONE GPU:
  * the RCCL transport with a single rank whose rows loop back through ncclSend / ncclRecv (MIL_MG_FLAG_LOOPBACK): the
    library's own dlopen'ed RCCL path runs on real hardware, and the degenerate gather == the monolithic frame;
  * the peer-copy transport with 2..8 ranks all placed on device 0: every piece of the strip / band / chunk / piece logic
    that an 8-GPU node runs, bit for bit against the monolithic frame (RCCL itself refuses two ranks on one device, so
    the N > 1 RCCL run is the driver's: bench.py --gpus N).
The reference has one device (render/vulkan.rs:106-108): the bar is assembled frame == single-GPU frame, bit for bit."""
import numpy as np
import pytest
import torch

from renderer_amd import lighting, multigpu, sharding, synth, ubo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
AMB = (0.03, 0.03, 0.03)


def scene(w, h, n_point, shadow, cam=None):
    cam = cam or synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(n_point, 21) if n_point else None
    sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow else None
    return gb, cam.frame_matrix_data(), dl, pl, sh


def parts_for(mg, gb, pl, sh, h, fmt, with_out_for_rank0, band_rows):
    parts = []
    for r in range(mg.world):
        rows = torch.from_numpy(multigpu.frame_rows_of_rank(h, mg.world, r, band_rows=band_rows)).to(DEV)
        sub = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
        w = gb['depth'].shape[1]
        out = None
        if r != 0 or with_out_for_rank0:
            out = torch.full((len(rows), w, 4), -1.0, device=DEV) if fmt == lighting.OUT_HDR else torch.full((len(rows), w), -1, dtype=torch.int32, device=DEV)
        parts.append(dict(gb=sub, point_lights=pl, ambient=AMB, shadow=sh, out=out))
    return parts


def test_rccl_single_rank_loopback_equals_monolithic():
    lp = lighting.LightingPass(0, 4096)
    w, h = 320, 200
    gb, frame, dl, pl, sh = scene(w, h, 60, 128)
    for band_rows, chunks in [(32, 4), (0, 3), (16, 1)]:
        for flags_mg in (multigpu.FLAG_LOOPBACK, 0):
            mg = multigpu.MultiGpu.single_process([0], band_rows=band_rows, chunks=chunks, transport=multigpu.TRANSPORT_RCCL, flags=flags_mg)
            assert mg.world == 1 and mg.local == 1 and mg.rccl_ranks == 1
            for fmt in (lighting.OUT_SRGB8, lighting.OUT_HDR):
                for flags in (0, lighting.FLAG_STRICT):
                    want = lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=fmt, flags=flags)
                    full = torch.zeros_like(want)
                    parts = parts_for(mg, gb, pl, sh, h, fmt, bool(flags_mg), band_rows)
                    mg.render_strips(parts, frame, dl, w, h, fmt=fmt, flags=flags, full=full)
                    mg.sync()
                    assert torch.equal(full.view(torch.int32), want.view(torch.int32)), (band_rows, chunks, flags_mg, fmt, flags)
            # frames mode: the one rank's frame arrives in slot 0
            slot = torch.zeros((h, w), dtype=torch.int32, device=DEV)
            parts = parts_for(mg, gb, pl, sh, h, lighting.OUT_SRGB8, bool(flags_mg), 0)
            mg.render_frames(parts, [frame], dl, w, h, slots=[slot])
            mg.sync()
            assert torch.equal(slot, lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8))
            mg.close()
    lp.close()


@pytest.mark.parametrize('world,band_rows,chunks', [(8, 32, 4), (8, 0, 4), (3, 16, 2), (2, 64, 1), (5, 32, 7)])
def test_emulated_ranks_strips_equal_monolithic(world, band_rows, chunks):
    lp = lighting.LightingPass(0, 4096)
    mg = multigpu.MultiGpu.single_process([0] * world, band_rows=band_rows, chunks=chunks, transport=multigpu.TRANSPORT_PEER_COPY)
    assert mg.world == world and mg.local == world and mg.rccl_ranks == 0
    for (w, h), n_point, shadow in [((640, 360), 120, 256), ((130, 77), 0, 64)]:
        gb, frame, dl, pl, sh = scene(w, h, n_point, shadow)
        for fmt, flags in [(lighting.OUT_SRGB8, 0), (lighting.OUT_HDR, lighting.FLAG_STRICT)]:
            want = lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=fmt, flags=flags)
            fulls = [torch.zeros_like(want) for _ in range(2)]
            parts = parts_for(mg, gb, pl, sh, h, fmt, False, band_rows)
            for f in fulls:                                  # two frames back to back, no sync in between: the ranks' output
                mg.render_strips(parts, frame, dl, w, h, fmt=fmt, flags=flags, full=f)   # buffers are reused behind the gather
            mg.sync()
            for f in fulls:
                assert torch.equal(f.view(torch.int32), want.view(torch.int32)), (w, h, fmt, flags)
    # a rank whose planes do not hold the rows the layout gives it is refused before anything is enqueued
    parts[1]['gb'] = {k: (v[:-1] if torch.is_tensor(v) else v) for k, v in parts[1]['gb'].items()}
    with pytest.raises(lighting.MilError, match='-1'):
        mg.render_strips(parts, frame, dl, w, h, fmt=fmt, flags=flags, full=fulls[0])
    mg.close()
    lp.close()


def test_emulated_ranks_4k_headline_bands():
    """configs[2] cut over 8 emulated ranks in dealt 32-row bands, 4 chunks: == the monolithic 4K frame."""
    lp = lighting.LightingPass(0, 4096)
    w, h = 3840, 2160
    gb, frame, dl, pl, sh = scene(w, h, 256, 2048)
    want = lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8)
    mg = multigpu.MultiGpu.single_process([0] * 8, band_rows=32, chunks=4, transport=multigpu.TRANSPORT_PEER_COPY)
    full = torch.zeros_like(want)
    mg.render_strips(parts_for(mg, gb, pl, sh, h, lighting.OUT_SRGB8, False, 32), frame, dl, w, h, full=full)
    mg.sync()
    assert torch.equal(full, want)
    mg.close()
    lp.close()


def test_emulated_ranks_frames_mode():
    """One frame per rank (configs[4]): 3 ranks, each its own camera and moving lights, every frame lands in its slot."""
    lp = lighting.LightingPass(0, 4096)
    w, h, world = 384, 216, 3
    mg = multigpu.MultiGpu.single_process([0] * world, band_rows=32, chunks=3, transport=multigpu.TRANSPORT_PEER_COPY)
    dl = ubo.default_directional_lights()
    sh = synth.shadow_setup(dl[0, 0:3], 128, device=DEV)
    base = synth.point_light_field(50, 4)
    parts, frames, want = [], [], []
    for r in range(world):
        cam = sharding.orbit_camera(r * 5, 64, w, h)
        gb = synth.scene_sweep(w, h, cam, device=DEV)
        pl = sharding.animated_point_lights(base, r * 5, 64)
        frames.append(cam.frame_matrix_data())
        out = None if r == 0 else torch.zeros((h, w), dtype=torch.int32, device=DEV)
        parts.append(dict(gb=gb, point_lights=pl, ambient=AMB, shadow=sh, out=out))
        want.append(lp.light_ex(gb, frames[-1], dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8))
    slots = [torch.zeros((h, w), dtype=torch.int32, device=DEV) for _ in range(world)]
    mg.render_frames(parts, frames, dl, w, h, slots=slots)
    mg.sync()
    for r in range(world):
        assert torch.equal(slots[r], want[r]), r
    mg.close()
    lp.close()


def test_frames_in_flight_rotating_and_reused_buffers():
    """Several frames enqueued without a sync in between: with rotating output buffers (the gather of frame i may run under the
    shading of frame i + 1) and with ONE buffer reused by every frame (each frame must wait for the previous gather): every
    assembled frame == the monolithic launch of its own light set, bit for bit.  4 emulated ranks, 16-row bands."""
    w, h, n = 512, 288, 4
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    frame = cam.frame_matrix_data()
    dl = ubo.default_directional_lights()
    sh = synth.shadow_setup(dl[0, 0:3], 128, device=DEV)
    lights = [synth.point_light_field(1200, 0x5EED0100 + k) for k in range(6)]
    lp = lighting.LightingPass(0, 4096)
    want = [lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8) for pl in lights]
    mg = multigpu.MultiGpu.single_process([0] * n, max_point_lights=4096, band_rows=16, chunks=1, transport=multigpu.TRANSPORT_PEER_COPY)
    try:
        rows = [torch.from_numpy(multigpu.frame_rows_of_rank(h, n, r, band_rows=16)).to(DEV) for r in range(n)]
        parts_gb = [{k: (v[rows[r]].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()} for r in range(n)]
        for rotate in (True, False):
            n_buf = 3 if rotate else 1
            outs = [[None if r == 0 else torch.zeros((len(rows[r]), w), dtype=torch.int32, device=DEV) for r in range(n)] for _ in range(n_buf)]
            fulls = [torch.zeros((h, w), dtype=torch.int32, device=DEV) for _ in range(len(lights) if rotate else 1)]
            for k, pl in enumerate(lights):
                parts = [dict(gb=parts_gb[r], point_lights=pl, ambient=AMB, shadow=sh, out=outs[k % n_buf][r]) for r in range(n)]
                mg.render_strips(parts, frame, dl, w, h, fmt=lighting.OUT_SRGB8, full=fulls[k if rotate else 0])
                if not rotate:          # one destination: look at it before the next frame overwrites it
                    mg.sync()
                    assert torch.equal(fulls[0], want[k]), k
            mg.sync()
            if rotate:
                for k in range(len(lights)):
                    assert torch.equal(fulls[k], want[k]), k
    finally:
        mg.close()
        lp.close()


@pytest.mark.parametrize('lanes', [2, 1])
def test_reused_sender_buffers_and_no_sync(lanes):
    """A rank shades consecutive frames on two lanes (context + stream) in turn, so frame i + 1's launch may start under frame
    i's tail.  Here every frame has its own lights (each lane's context stages its own), enough of them for the two-kernel
    form (a workspace per lane), its own destination on rank 0 - and the SENDERS' row buffers are the same for every frame,
    with no sync in between: frame i + 1's shading must wait until frame i's rows have left the buffer (Local::touched).
    The senders' gather streams are held back by 1 ms per frame so that a missing wait shows: on a build without it the
    one-lane run (MIL_MG_LANES=1) gets 6 of 7 frames wrong.  The two-lane run passed on that build too - on one GPU the
    emulated ranks' dozen streams share four hardware queues, and an event wait of one stream holds back every later packet
    of its queue - so for two lanes this is the guard that the box allows, not a proof.  Every frame == its monolithic
    launch, bit for bit."""
    import os
    w, h, n = 512, 288, 2
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    frame = cam.frame_matrix_data()
    dl = ubo.default_directional_lights()
    sh = synth.shadow_setup(dl[0, 0:3], 128, device=DEV)
    lights = [synth.point_light_field(1400 + 100 * (k % 2), 0x5EED0200 + k) for k in range(7)]
    lp = lighting.LightingPass(0, 4096)
    want = [lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8) for pl in lights]
    before = os.environ.get('MIL_MG_LANES')
    os.environ['MIL_MG_LANES'] = str(lanes)          # read by mil_mg_create
    try:
        mg = multigpu.MultiGpu.single_process([0] * n, max_point_lights=4096, band_rows=16, chunks=1, transport=multigpu.TRANSPORT_PEER_COPY)
    finally:
        if before is None:
            del os.environ['MIL_MG_LANES']
        else:
            os.environ['MIL_MG_LANES'] = before
    try:
        rows = [torch.from_numpy(multigpu.frame_rows_of_rank(h, n, r, band_rows=16)).to(DEV) for r in range(n)]
        parts_gb = [{k: (v[rows[r]].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()} for r in range(n)]
        outs = [None if r == 0 else torch.zeros((len(rows[r]), w), dtype=torch.int32, device=DEV) for r in range(n)]
        fulls = [torch.zeros((h, w), dtype=torch.int32, device=DEV) for _ in lights]
        mg.reserve(w, h, lighting.OUT_SRGB8)
        for k, pl in enumerate(lights):
            mg.debug_delay_senders(1000)
            parts = [dict(gb=parts_gb[r], point_lights=pl, ambient=AMB, shadow=sh, out=outs[r]) for r in range(n)]
            mg.render_strips(parts, frame, dl, w, h, fmt=lighting.OUT_SRGB8, full=fulls[k])
        mg.sync()
        for k in range(len(lights)):
            assert torch.equal(fulls[k], want[k]), k
    finally:
        mg.close()
        lp.close()


def test_staging_slices_survive_a_held_back_scatter():
    """dealt bands of ranks 1.. travel into staging slices on rank 0's device and are scattered into the frame
    from there; with the peer-copy transport the copies INTO the slices run on the senders' streams, and nothing used to
    order frame i + 1's copies behind frame i's scatter - only timing hid it.  Here the scatter of every frame is held back
    (mil_mg_debug_delay_gather idles rank 0's gather stream in front of it) while the next frame, with different lights, is
    already shaded and sent: every assembled frame must still be its own.  (Checked against the unfixed build: with two
    ranks it FAILS there - frame k shows frame k + 1's rows; with four ranks the eight streams share the box's four hardware
    queues and the held-back queue happens to hold the senders back too, which is what had hidden the race.)"""
    w, h, n = 512, 288, 2          # two ranks = four streams: one hardware queue each (GPU_MAX_HW_QUEUES is 4), nothing serialises them by accident
    gb, frame, dl, _, sh = scene(w, h, 0, 128)
    lights = [synth.point_light_field(600, 0x5EED0200 + k) for k in range(5)]
    lp = lighting.LightingPass(0, 4096)
    want = [lp.light_ex(gb, frame, dl, pl, AMB, sh, fmt=lighting.OUT_SRGB8) for pl in lights]
    assert not torch.equal(want[0], want[1])
    mg = multigpu.MultiGpu.single_process([0] * n, max_point_lights=4096, band_rows=16, chunks=1, transport=multigpu.TRANSPORT_PEER_COPY)
    try:
        mg.reserve(w, h, lighting.OUT_SRGB8)
        rows = [torch.from_numpy(multigpu.frame_rows_of_rank(h, n, r, band_rows=16)).to(DEV) for r in range(n)]
        parts_gb = [{k: (v[rows[r]].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()} for r in range(n)]
        outs = [[None if r == 0 else torch.zeros((len(rows[r]), w), dtype=torch.int32, device=DEV) for r in range(n)] for _ in lights]
        fulls = [torch.zeros((h, w), dtype=torch.int32, device=DEV) for _ in lights]
        for k, pl in enumerate(lights):
            mg.debug_delay_gather(3000)            # 3 ms in front of frame k's receives and scatter; shading a frame takes ~0.1 ms
            parts = [dict(gb=parts_gb[r], point_lights=pl, ambient=AMB, shadow=sh, out=outs[k][r]) for r in range(n)]
            mg.render_strips(parts, frame, dl, w, h, fmt=lighting.OUT_SRGB8, full=fulls[k])
        mg.sync()
        for k in range(len(lights)):
            assert torch.equal(fulls[k], want[k]), k
    finally:
        mg.close()
        lp.close()
