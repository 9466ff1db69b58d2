"""mil_light_ex_noalloc, the one entry point of libmil.so that allocates nothing (include/mil.h; it has no counterpart in the
reference): its staging buffer is a static __device__ array instead of the context's ring.  Called through the C ABI and
compared bit for bit with mil_light_ex (the bar for integer / byte work and for two runs of the same FP32 program), on two
calls in a row, the second with a smaller input - so what the first call left in the shared array must not leak into the
second.  Collected last (the round's end runs pytest -x)."""
import ctypes

import numpy as np
import pytest
import torch

from renderer_amd import lighting, synth, ubo

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
AMB = (0.03, 0.03, 0.03)


def scene(w, h, n_point, shadow, seed):
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device=DEV)
    dl = ubo.default_directional_lights()
    pl = synth.point_light_field(n_point, seed) if n_point else np.zeros((0, 8), np.float32)
    sh = synth.shadow_setup(dl[0, 0:3], shadow, device=DEV) if shadow else None
    return gb, cam.frame_matrix_data(), dl, pl, sh


def raw_call(lp, fn, gb, frame, dl, pl_ptr, n_point, sh, out, fmt, flags):
    """One call through the C ABI exactly as a C host would make it."""
    frame = np.ascontiguousarray(frame, np.float32).reshape(68)
    dl = np.ascontiguousarray(dl, np.float32).reshape(-1, 8)
    push = ubo.push_constants(gb['width'], gb['height'], dl.shape[0])
    ex = lighting.Extras()
    ex.point_lights, ex.n_point = pl_ptr, n_point
    ex.ambient = (ctypes.c_float * 3)(*AMB)
    keep = None
    if sh is not None:
        keep = lighting.Shadow((ctypes.c_float * 16)(*[float(v) for v in sh['light_view_proj']]), sh['map'].data_ptr(), sh['map'].shape[0], float(sh['bias']))
        ex.shadow = ctypes.pointer(keep)
    g = lp._gb(gb)
    pitch = out.stride(0) // 4 if fmt == lighting.OUT_HDR else out.stride(0)
    stream = ctypes.c_void_p(torch.cuda.current_stream(lp.device).cuda_stream)
    return fn(lp.ctx, ctypes.byref(g), frame.ctypes.data, dl.ctypes.data, push.ctypes.data, ctypes.byref(ex), out.data_ptr(), pitch, fmt, flags, stream)


def test_noalloc_equals_light_ex_on_two_calls_in_a_row():
    lp = lighting.LightingPass(0, max_point_lights=4096)
    lib = lp.lib
    # the second call is smaller in every way: frame, point lights, directional lights beyond the inline records
    many_dir = ubo.directional_lights([((1, 2 + i, 0.5 * i), 0.1 + 0.05 * i, (1, 0.9, 0.8)) for i in range(9)])
    cases = [dict(size=(320, 180), n_point=300, shadow=128, dl=many_dir, seed=5),
             dict(size=(96, 54), n_point=40, shadow=64, dl=None, seed=6)]
    for fmt in (lighting.OUT_HDR, lighting.OUT_SRGB8_VIA_LDR):
        for flags in (0, lighting.FLAG_STRICT, lighting.FLAG_NO_CULL):
            got, want = [], []
            for c in cases:                                     # the two no-alloc calls back to back, nothing in between
                w, h = c['size']
                gb, frame, dl, pl, sh = scene(w, h, c['n_point'], c['shadow'], c['seed'])
                dl = dl if c['dl'] is None else c['dl']
                pl_dev = torch.from_numpy(pl).to(DEV)
                shape = (h, w, 4) if fmt == lighting.OUT_HDR else (h, w)
                out = torch.full(shape, -3, dtype=torch.float32 if fmt == lighting.OUT_HDR else torch.int32, device=DEV)
                assert raw_call(lp, lib.mil_light_ex_noalloc, gb, frame, dl, pl_dev.data_ptr(), pl.shape[0], sh, out, fmt, flags) == 0
                got.append(out)
                c['inputs'] = (gb, frame, dl, pl, sh, pl_dev)
            for c in cases:                                     # the existing entry point on the same inputs
                gb, frame, dl, pl, sh, _ = c['inputs']
                h, w = gb['depth'].shape
                shape = (h, w, 4) if fmt == lighting.OUT_HDR else (h, w)
                out = torch.empty(shape, dtype=torch.float32 if fmt == lighting.OUT_HDR else torch.int32, device=DEV)
                assert raw_call(lp, lib.mil_light_ex, gb, frame, dl, pl.ctypes.data, pl.shape[0], sh, out, fmt, flags) == 0
                want.append(out)
            torch.cuda.synchronize()
            for i, (a, b) in enumerate(zip(got, want)):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (fmt, flags, i)
    # no point light and <= 4 directional lights: every record travels in the kernarg, the array is not touched
    gb, frame, dl, pl, sh = scene(64, 48, 0, 0, 1)
    a = lp.light_ex_noalloc(gb, frame, dl, None, AMB, None)
    assert torch.equal(a.view(torch.int32), lp.light_ex(gb, frame, dl, None, AMB, None).view(torch.int32))
    # sizes beyond the static array are refused with a clear status, before anything is enqueued
    big = torch.zeros((lighting.NOALLOC_MAX_POINT_LIGHTS + 1, 8), dtype=torch.float32, device=DEV)
    with pytest.raises(lighting.MilError, match='-5'):
        lp.light_ex_noalloc(gb, frame, dl, big, AMB, None)
    with pytest.raises(lighting.MilError, match='-2'):
        lp.light_ex_noalloc(gb, frame, np.zeros((101, 8), np.float32), None, AMB, None)
    # the full capacity works, and a context whose own capacity is smaller may still use the static array
    small = lighting.LightingPass(0, max_point_lights=8)
    gb, frame, dl, pl, sh = scene(128, 72, lighting.NOALLOC_MAX_POINT_LIGHTS, 0, 9)
    a = small.light_ex_noalloc(gb, frame, dl, torch.from_numpy(pl).to(DEV), AMB, None)
    assert torch.equal(a.view(torch.int32), lp.light_ex(gb, frame, dl, pl, AMB, None).view(torch.int32))
    small.close()
    lp.close()


def test_noalloc_two_kernel_lists_two_streams_and_graph_capture():
    """(1) the two-kernel path of the entry point on REAL lists (capacity 4096, threshold 1, 300 lights: k_bin's
    lists are read through the static array's cull / prepared records), == mil_light_ex bit for bit; (2) it alternates with
    mil_light_ex of the same context on another stream (one workspace per context: the calls are serialised by the
    workspace's event, which the entry point waits for and records - neither allocates); (3) the use it exists for: the
    call is captured into a HIP graph and replayed; hipMemGetInfo does not move across the calls."""
    lp = lighting.LightingPass(0, max_point_lights=4096)
    lp.set_two_kernel_threshold(1)
    gb, frame, dl, pl, sh = scene(320, 180, 300, 128, 11)
    pl_dev = torch.from_numpy(pl).to(DEV)
    want = lp.light_ex(gb, frame, dl, pl, AMB, sh)
    out = torch.empty_like(want)
    lp.light_ex_noalloc(gb, frame, dl, pl_dev, AMB, sh, out=out)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        out.fill_(-1.0)
        lp.light_ex_noalloc(gb, frame, dl, pl_dev, AMB, sh, out=out)
        assert torch.equal(out.view(torch.int32), want.view(torch.int32))
    assert torch.cuda.mem_get_info()[0] == free0 and lp.workspace_info()['frame_call_allocations'] == 0
    assert lp.workspace_info()['last_pool_entries'] > 0                     # lists were built: not the brute-force fallback
    # (2) two streams, alternating entry points, a second scene so that a stale workspace would show
    gb2, frame2, dl2, pl2, _ = scene(256, 144, 500, 0, 12)
    pl2_dev = torch.from_numpy(pl2).to(DEV)
    want2 = lp.light_ex(gb2, frame2, dl2, pl2, AMB, None)
    sa, sb = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    outs = []
    for i in range(6):
        o1, o2 = torch.empty_like(want), torch.empty_like(want2)
        lp.light_ex(gb, frame, dl, pl, AMB, sh, out=o1, stream=sa)
        lp.light_ex_noalloc(gb2, frame2, dl2, pl2_dev, AMB, None, out=o2, stream=sb)
        outs.append((o1, o2))
    torch.cuda.synchronize()
    for o1, o2 in outs:
        assert torch.equal(o1.view(torch.int32), want.view(torch.int32)) and torch.equal(o2.view(torch.int32), want2.view(torch.int32))
    # (3) under capture: nothing is allocated, waited for or recorded outside the graph; the replay writes the same bits
    out.fill_(-1.0)
    graph = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream(device=DEV)
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=cs):
        lp.light_ex_noalloc(gb, frame, dl, pl_dev, AMB, sh, out=out, stream=cs)
    torch.cuda.synchronize()
    assert torch.all(out == -1.0)                                            # captured, not run
    for _ in range(2):
        out.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out.view(torch.int32), want.view(torch.int32))
    # the workspace is pinned now: it may not move under the graph
    with pytest.raises(lighting.MilError, match='-5'):
        lp.set_list_pool(1 << 20)
    assert torch.equal(lp.light_ex(gb, frame, dl, pl, AMB, sh).view(torch.int32), want.view(torch.int32))
    del graph
    lp.close()
