#!/usr/bin/env python3
"""Benchmark of the MI355X lighting pass on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1 (the configuration BASELINE.json's metric is quoted on): a step = one pass of the hot
path over one 3840x2160 synthetic G-buffer (Scene A): the lighting kernel (2
directional lights of main.rs:80-91, light 0 with 2048^2 3x3-PCF shadow, 256 point lights
culled per 16x16 tile into LDS lists, ambient) writing the RGBA32F hdr attachment, then the
tonemap/gamma kernel (ACES + sRGB8).  Inputs are resident in HBM before the timed region;
three buffer sets (> 256 MiB) rotate so the Infinity Cache cannot hold a frame.

N > 1 (north_star's multi-GPU claim): by default ONE 7680x4320 frame with 4096 point lights
(BASELINE.json configs[3]) per step, cut over the ranks in dealt 32-row bands by libmil's
multi-GPU layer (mil_mg_create_rank + mil_mg_render_strips: one process per GPU, RCCL
communicator inside the library, grouped ncclSend / ncclRecv per chunk beside the shading of
the next chunk) and ASSEMBLED ON GPU 0 inside the timed region: "scaling": "strong", `value`
= frame pixels x steps / time with the gather included.  The same frame on rank 0 alone
through the same code path is timed in the same run and reported as
config.one_gpu_same_workload (the N = 1 default line is a different, smaller workload, so
value(N) / value(1) across the two is NOT a scaling efficiency).  --mode frames: one 4K
frame per rank per step (configs[4], "weak"), every frame gathered into its slot on GPU 0
inside the timed region; the gather-free rate is reported beside it in config.
--gather torch keeps the round-1 Python gather (torch.distributed batch_isend_irecv).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: FP32 vector peak
# Algorithmic FLOP constants frozen in SURVEY.md 8(d)
C_PIX, C_DIR_MIN, C_POINT_MIN = 75, 80, 100

WORKLOADS = {
    # name: (width, height, n_dir, n_point, shadow_size, ambient)
    'headline': (3840, 2160, 2, 256, 2048, True),
    'k1_1light': (3840, 2160, 1, 0, 0, False),
    'cfg2_1080p': (1920, 1080, 1, 0, 2048, False),
    'cfg4_8k': (7680, 4320, 2, 4096, 2048, True),
    'valu_4k_4096': (3840, 2160, 2, 4096, 2048, True),      # north_star: the 4K / 4096-light pass (FP32 VALU target)
}


TWO_KERNEL_MIN = 1280     # include/mil.h MIL_TWO_KERNEL_DEFAULT: point lights from which mil_light_ex takes k_bin -> k_shade


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default=None, choices=sorted(WORKLOADS),
                    help='default: headline at N = 1 and for --mode frames, cfg4_8k for N > 1 strips')
    ap.add_argument('--sets', type=int, default=3, help='rotating buffer sets (cold-cache protocol)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-extra', action='store_true', help='skip the 4K/1-light HBM-bound side measurement')
    ap.add_argument('--mode', default='strips', choices=['frames', 'strips'],
                    help='N > 1: strips = one frame cut over the ranks (strong scaling, the default); frames = every rank '
                         'shades a whole frame per step (weak scaling); both gathered to GPU 0 inside the timed region')
    ap.add_argument('--gather', default='mil_mg', choices=['mil_mg', 'torch'],
                    help='N > 1: mil_mg = the multi-GPU layer of libmil.so (RCCL inside the library); torch = the Python gather '
                         '(torch.distributed P2P; the only choice for the one-GPU gloo rehearsal)')
    ap.add_argument('--band-rows', type=int, default=16, help='mil_mg strips: rows per dealt band (0 = contiguous strips)')
    ap.add_argument('--chunks', type=int, default=1, help='mil_mg: launches per rank and frame (1 = best throughput: frames are enqueued back to back and the gather of frame i hides under the shading of frame i + 1; 2 = lower single-frame latency)')
    ap.add_argument('--pipelined', action='store_true',
                    help='N = 1: tonemap of frame i on a second stream beside the lighting of frame i + 1 (+3 %%; off by default so '
                         'that per-kernel durations under rocprofv3 are those of kernels running alone)')
    ap.add_argument('--balance', action='store_true',
                    help='--mode strips: cut the strips by cost (wave lifetimes of one calibration frame, exchanged by one all_gather) instead of by rows')
    ap.add_argument('--step', default='dual', choices=['dual', 'two', 'fused'],
                    help='N = 1 step: dual = ONE launch writes the hdr image and the tonemapped sRGB8 image (mil_extras.ldr_out; default); '
                         'two = lighting -> hdr, then the tonemap kernel -> sRGB8; fused = one launch -> sRGB8 only (no hdr image)')
    ap.add_argument('--fused', action='store_true', help='same as --step fused')
    ap.add_argument('--flags', type=int, default=0, help='extra MIL_FLAG_* bits (experiments)')
    args = ap.parse_args()
    if args.fused:
        args.step = 'fused'
    return args


def build_inputs(torch, synth, ubo, wl, device, n_sets, frame_idx=None, n_frames=64, rows=None, noise=False, all_lit=False):
    from renderer_amd import sharding
    w, h, n_dir, n_point, shadow_size, ambient = WORKLOADS[wl]
    cam = synth.sweep_camera(w, h) if frame_idx is None else sharding.orbit_camera(frame_idx, n_frames, w, h)
    if all_lit:
        cam = synth.all_lit_camera(w, h)
    gb0 = synth.scene_sweep(w, h, cam, device=device, rows=rows)       # rows: this rank's strip (global rows, y_offset set)
    if noise:        # Scene B: every plane i.i.d. random, every texel lit (nothing for an early-out to skip)
        gb0 = synth.scene_noise(w, h, 0x5EED0000, device=device)
    sets = [gb0] + [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in gb0.items()} for _ in range(n_sets - 1)]
    dl = ubo.default_directional_lights()[:n_dir]
    pl = synth.point_light_field(n_point, 0x5EED0003) if n_point else None
    sh = synth.shadow_setup(dl[0, 0:3], shadow_size, device=device) if shadow_size else None
    amb = (0.03, 0.03, 0.03) if ambient else None
    return dict(w=w, h=h, cam=cam, frame=cam.frame_matrix_data(), sets=sets, dl=dl, pl=pl, sh=sh, amb=amb,
                n_dir=n_dir, n_point=n_point, shadow_size=shadow_size, rows=rows)


def algorithmic(inp, pairs, mode, lit_px=None):
    """Per-launch algorithmic bytes and FLOPs of the lighting kernel.  FLOPs are counted for
    LIT pixels only (occlusion > 0): a cleared / background texel is exactly 0 whatever the lights are, and
    the kernel skips it, so counting it would flatter the VALU fraction."""
    px = inp['w'] * inp.get('lh', inp['h'])
    lit = px if lit_px is None else lit_px
    bytes_ = px * (16 + {'two': 16, 'dual': 20, 'fused': 4}[mode]) + 32 * (inp['n_dir'] + inp['n_point']) + 272
    if inp['shadow_size']:
        bytes_ += inp['shadow_size'] ** 2 * 4
    flops = lit * C_PIX + lit * inp['n_dir'] * C_DIR_MIN + pairs * C_POINT_MIN
    return bytes_, flops


def kernel_burst(torch, lighting, lp, inp, n, mode, flags, which='light'):
    """Average duration of ONE kernel: n back-to-back launches (rotating buffer sets) between two HIP events
    on the stream the kernel is launched on.  An event pair around a single launch adds several microseconds of
    command-processor latency; a burst amortises it and agrees with rocprofv3's per-kernel average."""
    dev = lp.device
    n_sets = len(inp['sets'])
    w, h = inp['w'], inp.get('lh', inp['h'])
    if h == 0:
        return float('nan')
    hdr = [torch.empty((h, w, 4), dtype=torch.float32, device=dev) for _ in range(n_sets)]
    out = [torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(n_sets)]

    def one(i):
        s = i % n_sets
        if which == 'tonemap':
            lp.tonemap(hdr[s], lighting.OUT_SRGB8, out=out[s])
        elif mode == 'fused':
            lp.light_ex(inp['sets'][s], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_SRGB8,
                        flags=flags, out=out[s])
        else:
            lp.light_ex(inp['sets'][s], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_HDR,
                        flags=flags, out=hdr[s], ldr_out=(out[s] if mode == 'dual' else None))
    # untimed: ~40 ms of the same launches, so that the burst runs at the sustained clock (tools/clock_ramp.py)
    t_settle = time.perf_counter()
    while True:
        for i in range(20):
            one(i)
        torch.cuda.synchronize()
        if time.perf_counter() - t_settle > 0.04:
            break
    ts = []
    for rnd in range(5):           # the median of five bursts (one burst of 20 x 0.1 ms wanders by +-4 % from run to run)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            one(i)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n)
    return sorted(ts)[2]


def two_in_flight(torch, lighting, lp, inp, n, mode, flags):
    """Frames alternating over two contexts on two streams (each context has its own light staging and workspace)."""
    dev = lp.device
    n_sets = len(inp['sets'])
    w, h = inp['w'], inp['h']
    lps = [lp, lighting.LightingPass(int(str(dev).split(':')[1]), max_point_lights=4096)]
    # four output sets: set o is always written on stream o % 2 (inputs are only read, any stream may share them)
    hdr = [torch.empty((h, w, 4), dtype=torch.float32, device=dev) for _ in range(4)]
    out = [torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(4)]
    # HIP deals streams over the process's hardware queues (4 here) in creation order, and two streams that share a queue run
    # their kernels one after the other: of four fresh streams the pair that overlaps best is used (probed below)
    pool = [torch.cuda.Stream(device=dev) for _ in range(4)]
    streams = pool[:2]
    cur = torch.cuda.current_stream(dev)

    def run(m):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for st in streams:
            st.wait_event(e0)
        for i in range(m):
            s, o, k = i % n_sets, i % 4, i % 2
            if mode == 'fused':
                lps[k].light_ex(inp['sets'][s], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_SRGB8,
                                flags=flags, out=out[o], stream=streams[k])
            else:
                lps[k].light_ex(inp['sets'][s], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_HDR,
                                flags=flags, out=hdr[o], ldr_out=(out[o] if mode == 'dual' else None), stream=streams[k])
        for st in streams:
            cur.wait_stream(st)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / m
    run(2 * n)
    probe = {}
    for a, b in ((0, 1), (0, 2), (0, 3), (1, 2)):
        streams = [pool[a], pool[b]]
        run(8)
        probe[(a, b)] = min(run(16) for _ in range(2))
    a, b = min(probe, key=probe.get)
    streams = [pool[a], pool[b]]
    ms = sorted(run(2 * n) for _ in range(5))[2]
    lps[1].close()
    px = w * h
    return dict(ms_per_frame=round(ms, 4), value=round(px / ms / 1e3, 1), unit='Mpix/s', streams=2, contexts=2,
                stream_pairs_probed={f'{a}+{b}': round(v, 4) for (a, b), v in probe.items()},
                note='same launches and buffers as `value`, consecutive frames on alternating streams; median of five bursts')


def frame_chain(torch, lighting, synth, lp, inp, n):
    dev = lp.device
    w, h, n_sets = inp['w'], inp['h'], len(inp['sets'])
    prm = synth.hosek_params_synthetic()
    acc, rev = synth.transparency_layer(w, h, dev)
    hdr = [torch.empty((h, w, 4), dtype=torch.float32, device=dev) for _ in range(n_sets)]
    ldr = [torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(n_sets)]
    out = [torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(n_sets)]
    args = (inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'])

    def chain(i):
        s = i % n_sets
        lp.light_ex(inp['sets'][s], *args, out=hdr[s])
        lp.sky_hosek(inp['sets'][s], inp['frame'], prm, hdr[s])
        lp.transparency_resolve(acc, rev, hdr[s])
        lp.tonemap(hdr[s], lighting.OUT_LDR, out=ldr[s])
        lp.fxaa(ldr[s], out=out[s])

    def one(i):
        s = i % n_sets
        lp.shade_frame(inp['sets'][s], *args, sky=prm, accum=acc, reveal=rev, fmt=lighting.OUT_LDR, out=ldr[s])
        lp.fxaa(ldr[s], out=out[s])

    def shade_only(i):
        s = i % n_sets
        lp.shade_frame(inp['sets'][s], *args, sky=prm, accum=acc, reveal=rev, fmt=lighting.OUT_LDR, out=ldr[s])

    def burst(fn):
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    c, o, so = burst(chain), burst(one), burst(shade_only)
    px = w * h
    return dict(five_call_chain_ms=round(c, 4), shade_frame_plus_fxaa_ms=round(o, 4), shade_frame_ms=round(so, 4),
                mpix_s=round(px / o / 1e3, 1), passes='2 dir + PCF + 256 point + ambient, sky on 49 % of the texels, resolve over 1/6')


def accumulate_burst(torch, synth, lp, inp, n, n_frag):
    """mil_transparency_accumulate over n_frag fragments (one full-screen translucent layer at 4K), the frame's two directional
    lights: 52 B read + 20 B written per fragment."""
    fr = synth.transparent_fragments(n_frag, seed=7)
    sets = [{k: torch.from_numpy(v).to(lp.device) for k, v in fr.items()} for _ in range(2)]
    acc = [torch.empty((n_frag, 4), dtype=torch.float32, device=lp.device) for _ in range(2)]
    rev = [torch.empty((n_frag,), dtype=torch.float32, device=lp.device) for _ in range(2)]

    def one(i):
        lp.transparency_accumulate(sets[i % 2], inp['frame'], inp['dl'], accum=acc[i % 2], reveal=rev[i % 2])
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    b = 72 * n_frag
    return dict(kernel='k_accumulate', fragments=n_frag, ms=round(ms, 4), mfrag_s=round(n_frag / ms / 1e3, 1), algorithmic_bytes=b,
                hbm_GBs=round(b / ms / 1e6, 1), hbm_frac=round(b / ms / 1e6 / HBM_PEAK_GBS, 4))


def run_workload(torch, lighting, lp, inp, steps, warmup, mode, flags, dist=None, world=1, rank=0, pipelined=True, gather=True):
    """Returns seconds for `steps` steps [max over ranks]."""
    from renderer_amd import sharding
    dev = lp.device
    n_sets = len(inp['sets'])
    w, h = inp['w'], inp['h']
    strips = inp.get('rows') is not None
    bounds = inp.get('bounds') or (sharding.strip_bounds(h, world) if strips else None)
    lh = (inp['rows'][1] - inp['rows'][0]) if strips else h            # rows this rank shades
    fused = mode == 'fused'
    hdr = [None if fused else torch.empty((lh, w, 4), dtype=torch.float32, device=dev) for _ in range(n_sets)]
    out = [torch.empty((lh, w), dtype=torch.int32, device=dev) for _ in range(n_sets)]
    if strips:      # rank 0 assembles the whole frame
        slots = [torch.empty((h, w), dtype=torch.int32, device=dev) if rank == 0 else None for _ in range(n_sets)]
    else:
        slots = [[torch.empty((h, w), dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(n_sets)] \
            if (world > 1 and rank == 0) else [None] * n_sets
    pending = [[] for _ in range(n_sets)]
    # Two HIP streams: the tonemap of frame i (HBM-bound) runs beside the lighting of frame i + 1 (VALU-bound).
    main = torch.cuda.current_stream(dev)
    post = torch.cuda.Stream(device=dev) if pipelined else main
    lit_done = [torch.cuda.Event() for _ in range(n_sets)]
    tone_done = [torch.cuda.Event() for _ in range(n_sets)]
    used = [False] * n_sets

    def step(i, ev=None):
        s = i % n_sets
        for q in pending[s]:          # the set's previous gather must be done before its buffers are rewritten
            q.wait()
        pending[s] = []
        gb = inp['sets'][s]
        if lh == 0:
            pass                                         # more ranks than 16-row strips: nothing to shade here
        elif fused:
            lp.light_ex(gb, inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_SRGB8,
                        flags=flags, out=out[s])
        elif mode == 'dual':
            lp.light_ex(gb, inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_HDR,
                        flags=flags, out=hdr[s], ldr_out=out[s])
        else:
            if used[s]:
                main.wait_event(tone_done[s])            # hdr[s] is still being read by the tonemap of n_sets steps ago
            lp.light_ex(gb, inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_HDR,
                        flags=flags, out=hdr[s], stream=main)
            lit_done[s].record(main)
            post.wait_event(lit_done[s])
            lp.tonemap(hdr[s], lighting.OUT_SRGB8, out=out[s], stream=post)
            tone_done[s].record(post)
            used[s] = True
        if world > 1 and gather:
            with torch.cuda.stream(post):
                if strips:
                    pending[s] = sharding.gather_strips(out[s], bounds, slots[s], dst=0, async_op=True)
                else:
                    pending[s] = sharding.gather_frames(out[s], slots[s], dst=0, async_op=True)

    def drain():
        for s in range(n_sets):
            for q in pending[s]:
                q.wait()
            pending[s] = []
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    drain()
    if dist is not None and world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    drain()
    if dist is not None and world > 1:
        dist.barrier()
    t1 = time.perf_counter()
    dt = t1 - t0
    if dist is not None and world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def cpu_baseline(inp, target_s=12.0):
    """The FP32 C restatement (oracle/mil_oracle.c, "port") on a bounded band of rows of the same
    workload, all host threads OpenMP gives us.  Reported beside the GPU number, not a target."""
    from oracle import coracle
    from renderer_amd import synth
    coracle.build()
    threads = coracle.max_threads()
    w, h = inp['w'], inp['h']
    sh = None if inp['sh'] is None else dict(inp['sh'], map=inp['sh']['map'].cpu().numpy())

    def run(rows, reps=1):
        # `rows` rows of the frame as up to eight bands spread evenly from top to bottom (one band through the
        # sphere wall is the most expensive part of the frame, which sold the CPU short); the whole frame when rows == h
        nonlocal threads
        nb = 1 if rows >= h else max(1, min(8, rows // 16))
        per = rows // nb
        bands = []
        for i in range(nb):
            y0 = 0 if nb == 1 and rows >= h else min(h - per, max(0, int((i + 0.5) * h / nb) - per // 2))
            strip = {k: (v[y0:y0 + per].cpu() if hasattr(v, 'cpu') else v) for k, v in inp['sets'][0].items()}
            bands.append((y0, synth.to_numpy(strip)))
        t0 = time.perf_counter()
        for _ in range(reps):
            for y0, g in bands:
                hdr = coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'], inp['frame'], inp['dl'],
                                    point_lights=inp['pl'], ambient=inp['amb'], shadow=sh, y_offset=y0, full_height=h,
                                    nthreads=threads)
                coracle.tonemap(hdr, 2, nthreads=threads)
        return time.perf_counter() - t0, nb * per

    def sample(budget_s):
        t, rows = run(16)
        rows_full = int(min(h, max(rows, rows * budget_s / max(t, 1e-6)))) // 16 * 16
        if rows_full > rows:
            t, rows = run(rows_full)
        reps = 1
        if t < budget_s:                   # a whole frame is quicker than the sample budget: repeat it
            reps = int(min(200, max(1, budget_s / max(t, 1e-6))))
            t, rows = run(rows, reps)
        return reps, rows, t

    all_threads = threads
    reps, rows, t = sample(target_s)
    res = dict(value=round(reps * rows * w / t / 1e6, 4), unit='Mpix/s', cores=all_threads, kind='port',
               sample=f'{reps} x {rows} of {h} rows x {w} px ({"the whole frame" if rows >= h else "bands spread evenly over the frame"}), same lights/shadow/ambient, '
                      f'lighting + tonemap, {t:.1f} s wall, oracle/mil_oracle.c brute-force over all lights, OpenMP')
    threads = 1                            # one thread beside all cores
    reps, rows, t = sample(target_s / 2)
    res['one_thread'] = dict(value=round(reps * rows * w / t / 1e6, 4), unit='Mpix/s', cores=1,
                             sample=f'{reps} x {rows} rows x {w} px, {t:.1f} s wall, same spread of bands and lights')
    return res


def run_mg(args, torch, dist, world, rank, local_rank):
    """N > 1 through libmil's multi-GPU layer (one process per GPU, RCCL inside the library).  Returns the result dict on
    rank 0, None elsewhere."""
    from renderer_amd import lighting, multigpu, sharding, synth, ubo
    dev = f'cuda:{local_rank}'
    strips = args.mode == 'strips'
    wl = args.workload or ('cfg4_8k' if strips else 'headline')
    w, h, n_dir, n_point, shadow_size, ambient = WORKLOADS[wl]
    px = w * h
    # the 128-byte RCCL id: made by rank 0 inside libmil (ncclGetUniqueId), carried by one torch.distributed broadcast
    idt = torch.zeros(multigpu.UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
    if rank == 0:
        idt.copy_(torch.frombuffer(bytearray(multigpu.MultiGpu.unique_id()), dtype=torch.uint8))
    dist.broadcast(idt, 0)
    band_rows = args.band_rows if strips else 32
    mg = multigpu.MultiGpu.per_process(world, rank, local_rank, idt.cpu().numpy().tobytes(), max_point_lights=4096,
                                       band_rows=band_rows, chunks=args.chunks)
    dl = ubo.default_directional_lights()[:n_dir]
    base_pl = synth.point_light_field(n_point, 0x5EED0003) if n_point else None
    sh = synth.shadow_setup(dl[0, 0:3], shadow_size, device=dev) if shadow_size else None
    amb = (0.03, 0.03, 0.03) if ambient else None
    n_sets = args.sets
    if strips:
        cam = synth.sweep_camera(w, h)
        pl = base_pl
        gb_full = synth.scene_sweep(w, h, cam, device=dev)
        rows = torch.from_numpy(multigpu.frame_rows_of_rank(h, world, rank, band_rows=band_rows)).to(dev)
        mine = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb_full.items()}
        if rank != 0:
            del gb_full
    else:
        cam = sharding.orbit_camera(rank, 64, w, h)
        pl = sharding.animated_point_lights(base_pl, rank, 64) if base_pl is not None else None
        mine = synth.scene_sweep(w, h, cam, device=dev)
        gb_full = mine
    frame = cam.frame_matrix_data()
    lrows = mine['depth'].shape[0]
    sets = [mine] + [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in mine.items()} for _ in range(n_sets - 1)]
    outs = [None if rank == 0 else torch.empty((lrows, w), dtype=torch.int32, device=dev) for _ in range(n_sets)]
    parts = [[dict(gb=sets[s], point_lights=pl, ambient=amb, shadow=sh, out=outs[s])] for s in range(n_sets)]
    fulls = slots = None
    if rank == 0:
        if strips:
            fulls = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(n_sets)]
        else:
            slots = [[torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(world)] for _ in range(n_sets)]

    mg.reserve(w, h, lighting.OUT_SRGB8)     # staging area + workspaces ahead of the loop: no per-frame call allocates

    def step(i):
        s = i % n_sets
        if strips:
            mg.render_strips(parts[s], frame, dl, w, h, fmt=lighting.OUT_SRGB8, flags=args.flags, full=fulls[s] if rank == 0 else None)
        else:
            mg.render_frames(parts[s], [frame], dl, w, h, fmt=lighting.OUT_SRGB8, flags=args.flags, slots=slots[s] if rank == 0 else None)

    def timed(fn, sync, steps, warmup, everyone=True):
        for i in range(warmup):
            fn(i)
        sync()
        if everyone:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(warmup + i)
        sync()
        if everyone:
            dist.barrier()
        dt = time.perf_counter() - t0
        if everyone:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # shading only, every rank for itself (no gather): what the transfer costs on top
    lp = lighting.LightingPass(local_rank, max_point_lights=4096)
    tmp = torch.empty((max(lrows, 1), w), dtype=torch.int32, device=dev)

    def shade_only(i):
        if lrows == 0:
            return
        gb = sets[i % n_sets]
        if strips and band_rows:
            lp.shade_bands(gb, frame, dl, (band_rows, world, rank, 0), pl, amb, sh, fmt=lighting.OUT_SRGB8, flags=args.flags, out=tmp)
        elif strips:
            lp.light_ex(dict(gb, y_offset=int(rows[0].item())), frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, flags=args.flags, out=tmp)
        else:
            lp.light_ex(gb, frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, flags=args.flags, out=tmp)
    # this leg first: it also brings every GPU to its sustained clock before the timed steps (tools/clock_ramp.py)
    dt_shade = timed(shade_only, torch.cuda.synchronize, args.steps, max(args.warmup, 40))
    dt = timed(step, mg.sync, args.steps, args.warmup)
    rccl_ranks = mg.rccl_ranks

    res = None
    if rank == 0:
        one = None
        equal = None
        if strips:
            # the same frame on this GPU alone through the same code path (one rank: no transfer at all)
            mg1 = multigpu.MultiGpu.single_process([local_rank], max_point_lights=4096, band_rows=band_rows, chunks=args.chunks,
                                                   transport=multigpu.TRANSPORT_PEER_COPY)
            full1 = torch.zeros((h, w), dtype=torch.int32, device=dev)
            p1 = [dict(gb=gb_full, point_lights=pl, ambient=amb, shadow=sh, out=None)]
            dt1 = timed(lambda i: mg1.render_strips(p1, frame, dl, w, h, fmt=lighting.OUT_SRGB8, flags=args.flags, full=full1),
                        mg1.sync, max(5, args.steps // 2), 2, everyone=False)
            n1 = max(5, args.steps // 2)
            one = dict(value=round(n1 * px / dt1 / 1e6, 1), unit='Mpix/s', ms_per_step=round(dt1 / n1 * 1e3, 4),
                       note='the whole frame on rank 0 alone, same library path (mil_mg with one rank), same run')
            equal = bool(torch.equal(full1, fulls[(args.warmup + args.steps - 1) % n_sets]))
            mg1.close()
        total_px = px if strips else world * px
        res = dict(metric='shaded Mpix/s at 3840x2160, 256 point lights + PCF shadows; % HBM/VALU roofline',
                   value=round(args.steps * total_px / dt / 1e6, 1), unit='Mpix/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True, scaling=('strong' if strips else 'weak'),
                   vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload=f'{wl}: {w}x{h} Scene A sweep, {n_dir} directional (light 0 PCF {shadow_size}^2) + {n_point} point lights'
                                        + (' + ambient' if amb else '') + ('' if strips else ', one frame per rank per step (orbiting camera, moving lights)'),
                               kernels=[('k_bin -> k_order -> k_shade' if n_point >= TWO_KERNEL_MIN else 'k_tiled') +
                                        ' + tonemap fused -> sRGB8 (mil_light_ex semantics inside mil_mg)'],
                               parallelism=(f'{world} ranks, one process per GPU; ' +
                                            (f'frame dealt in {band_rows}-row bands' if (strips and band_rows) else
                                             ('contiguous 16-row-aligned strips' if strips else 'one frame per rank')) +
                                            f', {args.chunks} chunks per rank, grouped ncclSend/ncclRecv to GPU 0 per chunk on a second stream (dealt bands: one message per rank into a staging slice, scattered into the frame by a strided copy)'),
                               gather='inside the timed region: `value` counts a frame when it is assembled on GPU 0',
                               gather_path='libmil mil_mg (RCCL bound inside the library)', rccl_ranks=rccl_ranks,
                               shading_only=dict(value=round(args.steps * total_px / dt_shade / 1e6, 1), unit='Mpix/s',
                                                 ms_per_step=round(dt_shade / args.steps * 1e3, 4),
                                                 note='every rank shading its rows with no transfer (max over ranks)'),
                               buffer_sets=n_sets))
        if one is not None:
            res['config']['one_gpu_same_workload'] = one
            res['config']['assembled_equals_single_gpu_bitwise'] = equal
            res['config']['note'] = ('the N = 1 default line of this bench is BASELINE.json\'s metric configuration (4K, 256 lights, two kernels): '
                                     'compare this value with config.one_gpu_same_workload, not with that line')
        # roofline of the lighting launch over this rank's rows (from 1280 point lights on: k_bin, k_order and k_shade together;
        # k_shade is > 85 % of it), HIP-event burst on its stream
        if lrows:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(3):
                shade_only(i)
            torch.cuda.synchronize()
            e0.record()
            nb = max(args.steps, 10)
            for i in range(nb):
                shade_only(i)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / nb
            abytes = lrows * w * 20 + 32 * (n_dir + n_point) + 272 + (shadow_size ** 2 * 4 if shadow_size else 0)
            hbm_roof = dict(bound='hbm', achieved=round(abytes / ms / 1e6, 1), peak=HBM_PEAK_GBS, unit='GB/s',
                                   frac=round(abytes / ms / 1e6 / HBM_PEAK_GBS, 4), traffic=None,
                                   kernel=('k_bin + k_order + k_shade' if n_point >= TWO_KERNEL_MIN else 'k_tiled') + ' (rank 0, its rows, fused sRGB8 store)',
                                   ms=round(ms, 4), algorithmic_bytes=abytes,
                                   )
            res['roofline'] = hbm_roof
            if strips and n_point:
                # FLOPs of the WHOLE frame (pairs counted on rank 0's copy of it) / world: the bands are dealt evenly
                gbf = dict(gb_full, y_offset=0)
                pairs = lp.count_pairs(gbf, frame, pl)
                lit = int((((gb_full['albedo_occ'] >> 24) & 0xFF) != 0).sum().item())
                flops = (lit * C_PIX + lit * n_dir * C_DIR_MIN + pairs * C_POINT_MIN) / world
                valu_roof = dict(bound='valu', achieved=round(flops / ms / 1e9, 2), peak=VALU_PEAK_TFLOPS, unit='TFLOP/s',
                                 frac=round(flops / ms / 1e9 / VALU_PEAK_TFLOPS, 4), traffic=None, kernel=hbm_roof['kernel'], ms=round(ms, 4),
                                 algorithmic_flops=int(flops), note='FLOPs of the whole frame / ranks (bands dealt evenly), lit pixels only')
                if valu_roof['frac'] > hbm_roof['frac']:
                    res['roofline'], res['roofline_other'] = valu_roof, hbm_roof
                else:
                    res['roofline_other'] = valu_roof
    lp.close()
    mg.close()
    return res


def emit(res, saved_stdout):
    """The ONE line of stdout."""
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    print(json.dumps(res), flush=True)
    os.dup2(2, 1)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start N ranks as CHILD processes of this one - which has made no
    GPU call and imports nothing that does - with the driver's own command line, relay rank 0's one JSON line, return the
    launcher's exit code.  No exec: the parent stays and only waits.  The ranks get MIL_BENCH_TIMEOUT seconds (default 1500);
    at the limit their process group is killed and nothing else is started.  If the ranks of the default path (mil_mg: RCCL
    inside libmil.so) END with an error that is not "no GPU", the same command runs once more over the Python gather
    (--gather torch) and its line says so (`fallback`): a scaling line over the older path beats none."""
    import signal
    import socket
    import subprocess
    import threading

    def run_ranks(extra):
        with socket.socket() as so:          # a free rendezvous port on the loopback address
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:] + extra
        print(f'bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks as child processes: {" ".join(cmd)}', file=sys.stderr, flush=True)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), MIL_BENCH_CHILD='1')
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, start_new_session=True)
        seen = dict(no_device=False)

        def relay():                         # the ranks' stderr goes through line by line (progress stays visible)
            for raw in child.stderr:
                ln = raw.decode(errors='replace')
                seen['no_device'] |= 'MIL_ERR_NO_DEVICE' in ln
                sys.stderr.write(ln)
                sys.stderr.flush()
        th = threading.Thread(target=relay, daemon=True)
        th.start()
        timed_out = False
        try:
            out, _ = child.communicate(timeout=float(os.environ.get('MIL_BENCH_TIMEOUT', '1500')))
        except subprocess.TimeoutExpired:
            timed_out = True
            os.killpg(child.pid, signal.SIGKILL)     # exactly the group started above
            out, _ = child.communicate()
        th.join(timeout=5)
        line = None
        for ln in out.decode(errors='replace').splitlines():
            try:
                if 'metric' in json.loads(ln):
                    line = ln
            except ValueError:
                print(ln, file=sys.stderr)          # anything else a rank wrote to stdout
        return child.returncode, line, timed_out, seen['no_device']

    rc, line, timed_out, no_device = run_ranks([])
    if rc == 0 and line is not None:
        print(line, flush=True)
        return 0
    why = 'time limit' if timed_out else f'launcher exit code {rc}' + ('' if line else ', no result line')
    print(f'bench.py: the {args.gpus}-rank run failed ({why})', file=sys.stderr)
    if timed_out or no_device or args.gather != 'mil_mg' or os.environ.get('MIL_BENCH_NO_FALLBACK') == '1':
        return rc or 1
    print('bench.py: once more over the Python gather (--gather torch)', file=sys.stderr)
    rc2, line2, _, _ = run_ranks(['--gather', 'torch'])
    if rc2 == 0 and line2 is not None:
        res = json.loads(line2)
        res['fallback'] = f'the mil_mg (RCCL inside libmil.so) run of this command failed ({why}); this line is the --gather torch path'
        print(json.dumps(res), flush=True)
        return 0
    return rc2 or rc or 1


def main():
    args = parse()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args))
    # stdout carries exactly one JSON line: everything else that lands on file descriptor 1 while the bench runs (RCCL prints a
    # version banner there when a communicator is created, inside torch as well as inside libmil) goes to stderr
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    from renderer_amd import lighting, synth, ubo

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        args.gpus = world
    if os.environ.get('MIL_BENCH_FAIL_MG') == '1' and world > 1 and args.gather == 'mil_mg':
        sys.exit(3)                          # test hook (tests/test_host_logic.py): the mil_mg ranks end with an error
    n_dev = torch.cuda.device_count()       # counting devices does not initialise the GPU
    rehearse_env = os.environ.get('MIL_BENCH_REHEARSE') == '1'
    if n_dev == 0 or (local_rank >= n_dev and not rehearse_env):
        sys.exit(f'bench.py rank {rank}: MIL_ERR_NO_DEVICE (-4): local rank {local_rank} has no GPU ({n_dev} visible); there is no CPU path')
    dist = None
    # rehearsal on a one-GPU box only: MIL_BENCH_REHEARSE=1 puts every rank on device 0 and joins them over gloo
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearse = os.environ.get('MIL_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_mg = os.environ.get('MIL_BENCH_FORCE_MG') == '1'       # smoke test of the mil_mg bench path with a world of one
    if world > 1 or force_mg:
        import torch.distributed as dist
        if force_mg and 'RANK' not in os.environ:          # world of one without a launcher: the rendezvous is with oneself
            os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=os.environ.get('MASTER_PORT', '29531'))
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = f'cuda:{local_rank}'

    if (world > 1 or force_mg) and args.gather == 'mil_mg' and not rehearse:
        res = run_mg(args, torch, dist, world, rank, local_rank)
        if rank == 0:
            emit(res, saved_stdout)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.workload is None:
        args.workload = 'cfg4_8k' if (world > 1 and args.mode == 'strips') else 'headline'

    lp = lighting.LightingPass(local_rank, max_point_lights=4096)
    strips = args.mode == 'strips' and world > 1          # one GPU: the whole frame, nothing to cut
    if strips:
        from renderer_amd import sharding
        y0, y1 = sharding.strip_bounds(WORKLOADS[args.workload][1], world)[rank]
        inp = build_inputs(torch, synth, ubo, args.workload, dev, args.sets, rows=(y0, y1))
        inp['lh'] = y1 - y0
        if args.balance and world > 1 and inp['pl'] is not None:
            # calibration frame on the equal-row strips: per-tile-row wave lifetimes -> all ranks -> the same weighted cuts
            hgt = WORKLOADS[args.workload][1]
            cost_local = np.zeros(0)
            if y1 > y0:
                st = lp.tile_stamps(inp['sets'][0])
                lp.light_ex(inp['sets'][0], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'])
                torch.cuda.synchronize()
                lp.tile_stamps(inp['sets'][0], enable=False)
                cost_local = sharding.band_costs_from_stamps(st)
            cost = sharding.exchange_band_costs(cost_local, y0, -(-hgt // 16), device=(None if rehearse else dev))
            bounds = sharding.strip_bounds_weighted(cost, world, hgt)
            y0, y1 = bounds[rank]
            del inp
            inp = build_inputs(torch, synth, ubo, args.workload, dev, args.sets, rows=(y0, y1))
            inp['lh'] = y1 - y0
            inp['bounds'] = bounds
    else:
        inp = build_inputs(torch, synth, ubo, args.workload, dev, args.sets, frame_idx=(rank if world > 1 else None))
    px = inp['w'] * inp['h']
    pairs = lp.count_pairs(inp['sets'][0], inp['frame'], inp['pl']) if (inp['pl'] is not None and inp.get('lh', 1)) else 0

    # The roofline leg (bursts of the kernel between HIP events) runs FIRST, on every rank: an idle MI355X takes ~20 ms of
    # work to reach its sustained clock (tools/clock_ramp.py: the first 100 launches after idle run 5-10 % slower), and W
    # is 5 in the driver's command - the timed steps below then measure the sustained rate, as the bursts do.
    light_ms = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), args.step, args.flags) if inp.get('lh', 1) else 0.0
    tone_burst_ms = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), 'two', 0, 'tonemap') if (args.step == 'two' and inp.get('lh', 1)) else 0.0
    # N > 1: the gather to GPU 0 is inside the timed region (a frame counts when it is on GPU 0)
    dt = run_workload(torch, lighting, lp, inp, args.steps, args.warmup, args.step, args.flags, dist, world, rank,
                      pipelined=(args.pipelined and world == 1), gather=True)
    dt_nogather = None
    if world > 1 and not strips:       # the same frames left on the GPUs that shaded them: what the transfer costs
        dt_nogather = run_workload(torch, lighting, lp, inp, args.steps, args.warmup, args.step, args.flags, dist, world, rank,
                                   pipelined=False, gather=False)
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    occ = (inp['sets'][0]['albedo_occ'] >> 24) & 0xFF
    lit_px = int((occ != 0).sum().item())
    abytes, aflops = algorithmic(inp, pairs, args.step, lit_px)
    hbm = abytes / (light_ms * 1e-3) / 1e9
    valu = aflops / (light_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    kname = 'k_tiled' if (inp['n_point'] or (inp['shadow_size'] and args.flags & 4)) else 'k_stream'
    if inp['n_point'] >= TWO_KERNEL_MIN and not (args.flags & 7):
        kname = 'k_bin+k_order+k_shade'
    if os.path.exists(tpath) and not strips:      # the PMC passes were taken on whole-frame launches
        t = json.load(open(tpath)).get(f'{args.workload}{dict(two="", dual="_dual", fused="_fused")[args.step]}:{kname}')
        if t:
            traffic = t['hbm_bytes_per_launch']
    roof_hbm = dict(bound='hbm', achieved=round(hbm, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(hbm / HBM_PEAK_GBS, 4),
                    traffic=traffic, traffic_source=(None if traffic is None else 'profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE '
                                                     'passes of tools/profile_round.sh on this kernel and workload, committed; not measured in this run)'),
                    kernel=kname, ms=round(light_ms, 4), algorithmic_bytes=abytes)
    roof_valu = dict(bound='valu', achieved=round(valu, 2), peak=VALU_PEAK_TFLOPS, unit='TFLOP/s',
                     frac=round(valu / VALU_PEAK_TFLOPS, 4), traffic=traffic, kernel=kname, ms=round(light_ms, 4),
                     algorithmic_flops=aflops, pairs_in_radius=pairs, lit_pixels=lit_px)
    # the binding roofline is the one the kernel sits closer to
    primary, other = (roof_hbm, roof_valu) if roof_hbm['frac'] >= roof_valu['frac'] else (roof_valu, roof_hbm)

    res = dict(metric='shaded Mpix/s at 3840x2160, 256 point lights + PCF shadows; % HBM/VALU roofline',
               value=round((1 if strips else world) * args.steps * px / dt / 1e6, 1), unit='Mpix/s', n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True,
               lit_mpix_s=round((1 if strips else world) * args.steps * lit_px / dt / 1e6, 1),
               scaling=('strong' if strips else 'weak'),
               vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload=f'{args.workload}: {inp["w"]}x{inp["h"]} Scene A sweep, {inp["n_dir"]} directional '
                                    f'(light 0 PCF {inp["shadow_size"]}^2) + {inp["n_point"]} point lights'
                                    + (' + ambient' if inp['amb'] else ''),
                           kernels=dict(fused=[f'{kname} + tonemap -> sRGB8 (no hdr image)'], two=[f'{kname} -> hdr RGBA32F', 'k_tonemap -> sRGB8'],
                                        dual=[f'{kname} -> hdr RGBA32F and, from the same registers, tonemapped sRGB8 (mil_extras.ldr_out): the lighting and '
                                              'tonemap subpasses of render/mod.rs:201-216 + 297-309 as one pass over the pixels, both images written'])[args.step],
                           buffer_sets=args.sets, parallelism=('single GPU' if world == 1 else (f'{world} horizontal strips (16-row aligned{", cost-weighted " + str([b - a for a, b in inp["bounds"]]) if inp.get("bounds") else ""}) + P2P gather to GPU 0' if strips else f'frame-parallel x{world}, every frame gathered to GPU 0 inside the timed region')),
                           tonemap_ms=round(tone_burst_ms, 4),
                           streams=('1 (serial)' if (not args.pipelined or args.step != 'two' or world > 1) else '2: tonemap of frame i beside lighting of frame i+1'),
                           kernel_ms_method='median of five bursts of back-to-back launches between two HIP events on the launch stream',
                           leg_order='kernel bursts (roofline) first, then the W warm-up and K timed steps: all at the sustained clock'),
               roofline=primary, roofline_other=other)
    if world > 1:
        res['config']['gather'] = 'inside the timed region: `value` counts a frame when it is on GPU 0'
        res['config']['gather_path'] = 'torch.distributed batch_isend_irecv (' + ('gloo rehearsal on one GPU: the numbers mean nothing' if rehearse else 'nccl = RCCL') + ')'
    if dt_nogather is not None:
        res['config']['shading_only'] = dict(value=round(world * args.steps * px / dt_nogather / 1e6, 1), unit='Mpix/s',
                                             ms_per_step=round(dt_nogather / args.steps * 1e3, 4),
                                             note='same loop with every frame left on the GPU that shaded it')

    if not args.no_extra and world == 1 and args.workload == 'headline':
        # side measurement: the HBM-bound 4K / 1-light lighting pass (BASELINE.md target >= 70 % of HBM roofline)
        inp1 = build_inputs(torch, synth, ubo, 'k1_1light', dev, args.sets)
        l1 = kernel_burst(torch, lighting, lp, inp1, max(args.steps, 20), 'two', 0)
        b1, f1 = algorithmic(inp1, 0, 'two')
        res['extra'] = {'k1_4k_1light': dict(kernel='k_stream -> hdr RGBA32F', ms=round(l1, 4), mpix_s=round(px / l1 / 1e3, 1),
                                             hbm_GBs=round(b1 / l1 / 1e6, 1), hbm_frac=round(b1 / l1 / 1e6 / HBM_PEAK_GBS, 4),
                                             algorithmic_bytes=b1)}
        del inp1
        # the same pass on Scene B (noise, every texel lit): no wave can skip anything, so this is the pure streaming rate
        inp1 = build_inputs(torch, synth, ubo, 'k1_1light', dev, args.sets, noise=True)
        l1 = kernel_burst(torch, lighting, lp, inp1, max(args.steps, 20), 'two', 0)
        res['extra']['k1_4k_1light_all_lit'] = dict(kernel='k_stream -> hdr RGBA32F, Scene B', ms=round(l1, 4), mpix_s=round(px / l1 / 1e3, 1),
                                                    hbm_GBs=round(b1 / l1 / 1e6, 1), hbm_frac=round(b1 / l1 / 1e6 / HBM_PEAK_GBS, 4),
                                                    note='tools/probe/stream_probe.hip: the same four-plane access pattern with no arithmetic takes 43.6 us (6.09 TB/s), a float4 copy 41.6 us')
        del inp1
        # side measurement: the VALU-bound 4K / 4096-light lighting pass (north_star target >= 50 % of FP32 VALU peak): k_bin ->
        # k_order -> k_shade, hdr out.  The FLOP fraction is quoted against the spec peak and against the packed-FMA rate
        # measured on this part (tools/probe: 4.8 cycles per dependent v_pk_fma_f32 at 4 waves per SIMD -> 131 TFLOP/s)
        inpv = build_inputs(torch, synth, ubo, 'valu_4k_4096', dev, args.sets)
        lv = kernel_burst(torch, lighting, lp, inpv, max(args.steps // 2, 10), 'two', args.flags)
        pairs_v = lp.count_pairs(inpv['sets'][0], inpv['frame'], inpv['pl'])
        occ_v = (inpv['sets'][0]['albedo_occ'] >> 24) & 0xFF
        bv, fv = algorithmic(inpv, pairs_v, 'two', int((occ_v != 0).sum().item()))
        res['extra']['k2_4k_4096'] = dict(kernel='k_bin + k_order + k_shade -> hdr RGBA32F', ms=round(lv, 4), mpix_s=round(px / lv / 1e3, 1),
                                          pairs_in_radius=pairs_v, algorithmic_flops=fv, tflops=round(fv / lv / 1e9, 2),
                                          valu_frac=round(fv / lv / 1e9 / VALU_PEAK_TFLOPS, 4), valu_frac_of_measured_131tf=round(fv / lv / 1e9 / 131.0, 4))
        del inpv, occ_v
        # side measurement: the headline step on an ALL-LIT view of the same scene (camera pitched onto floor + spheres: no texel
        # keeps the cleared G-buffer, so nothing is skipped by the occlusion-0 early-out that covers 49 % of the headline frame)
        inpl = build_inputs(torch, synth, ubo, 'headline', dev, args.sets, all_lit=True)
        dtl = run_workload(torch, lighting, lp, inpl, args.steps, args.warmup, args.step, args.flags, pipelined=False)
        ll = kernel_burst(torch, lighting, lp, inpl, max(args.steps, 20), args.step, args.flags)
        pairs_l = lp.count_pairs(inpl['sets'][0], inpl['frame'], inpl['pl'])
        bl, fl = algorithmic(inpl, pairs_l, args.step, px)
        res['extra']['headline_all_lit'] = dict(value=round(args.steps * px / dtl / 1e6, 1), unit='Mpix/s', ms_per_step=round(dtl / args.steps * 1e3, 4),
                                                k_tiled_ms=round(ll, 4), lit_pixels=px, pairs_in_radius=pairs_l,
                                                hbm_frac=round(bl / ll / 1e6 / HBM_PEAK_GBS, 4), valu_frac=round(fl / ll / 1e9 / VALU_PEAK_TFLOPS, 4),
                                                note='same lights, shadow, ambient and step as `value`; camera (7, 9, 4.5) pitched 63 degrees down')
        del inpl
        # side measurements: the same two images by two launches (lighting -> hdr, then k_tonemap reads it back -> sRGB8), and
        # the sRGB8 image alone by one launch (hdr stays in registers; what mil_shade_frame does)
        if args.step == 'dual':
            dt2 = run_workload(torch, lighting, lp, inp, args.steps, args.warmup, 'two', args.flags, pipelined=False)
            l2 = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), 'two', args.flags)
            t2 = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), 'two', 0, 'tonemap')
            res['extra']['headline_two_launches'] = dict(value=round(args.steps * px / dt2 / 1e6, 1), unit='Mpix/s', ms_per_step=round(dt2 / args.steps * 1e3, 4),
                                                         lighting_ms=round(l2, 4), tonemap_ms=round(t2, 4),
                                                         note='the step `value` timed before mil_extras.ldr_out existed: same hdr and sRGB8 images, bit for bit')
        # side measurement: the same step with TWO frames in flight - frames alternate over two contexts on two HIP streams, so
        # the ramp of one launch runs under the tail of the other (a launch spends its last ~12 us below half occupancy:
        # tools/stamps.py).  Throughput of a frame sequence, not the duration of a kernel; `value` stays the serial figure so
        # that it agrees with rocprofv3's per-kernel average.
        res['extra']['two_frames_in_flight'] = two_in_flight(torch, lighting, lp, inp, max(args.steps, 20), args.step, args.flags)
        if args.step != 'fused':
            lf = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), 'fused', args.flags)
            res['extra']['headline_one_launch'] = dict(kernel='k_tiled+tonemap -> sRGB8 only', ms=round(lf, 4), mpix_s=round(px / lf / 1e3, 1))
        # side measurement: the reference's whole frame, G-buffer -> swapchain image (render/mod.rs:189-329: lighting, sky,
        # transparency resolve, tonemap, FXAA), as the five-call chain and as mil_shade_frame + mil_fxaa (hdr never stored)
        res['extra']['frame_4k_gbuffer_to_swapchain'] = frame_chain(torch, lighting, synth, lp, inp, max(args.steps, 20))
        # side measurement: the accumulate draw's fragment shader over one full-screen translucent layer
        res['extra']['transparency_accumulate_4k_layer'] = accumulate_burst(torch, synth, lp, inp, max(args.steps, 20), px)
    if not args.no_cpu and world == 1:
        res['cpu_baseline'] = cpu_baseline(inp)
    emit(res, saved_stdout)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
