#!/usr/bin/env python3
"""Benchmark of the MI355X lighting pass on BASELINE.json's headline configuration.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over one 3840x2160 synthetic G-buffer (Scene A, SURVEY.md
8d): the lighting kernel (2 directional lights of main.rs:80-91, light 0 with 2048^2 3x3-PCF
shadow, 256 point lights culled per 16x16 tile into LDS lists, ambient) writing the
RGBA32F hdr attachment, then the fused tonemap/gamma kernel (ACES + sRGB8).  Inputs are
resident in HBM before the timed region; three buffer sets (> 256 MiB) rotate so the
Infinity Cache cannot hold a frame.  N > 1: frame-parallel (each rank shades its own frames
of an orbiting camera; the frames are independent units and no data-path collective joins
them); per-GPU work is fixed, so scaling is "weak".  The same loop is then timed once more
with every packed frame gathered to GPU 0 by overlapped grouped P2P (north_star's
presentation step; bound by one xGMI link per sender, 33 MB per frame) and reported beside
`value` as config.gather_to_gpu0.  --mode strips always gathers (one frame is assembled).
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
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='headline', choices=sorted(WORKLOADS))
    ap.add_argument('--sets', type=int, default=3, help='rotating buffer sets (cold-cache protocol)')
    ap.add_argument('--no-cpu', action='store_true', help='skip the cpu_baseline leg')
    ap.add_argument('--no-extra', action='store_true', help='skip the 4K/1-light HBM-bound side measurement')
    ap.add_argument('--mode', default='frames', choices=['frames', 'strips'],
                    help='N > 1: frames = every rank shades whole frames (weak scaling); strips = one frame cut into '
                         'horizontal strips on 16-row boundaries (strong scaling), both gathered to GPU 0')
    ap.add_argument('--pipelined', action='store_true',
                    help='N = 1: tonemap of frame i on a second stream beside the lighting of frame i + 1 (+3 %%; off by default so '
                         'that per-kernel durations under rocprofv3 are those of kernels running alone)')
    ap.add_argument('--balance', action='store_true',
                    help='--mode strips: cut the strips by cost (wave lifetimes of one calibration frame, exchanged by one all_gather) instead of by rows')
    ap.add_argument('--fused', action='store_true', help='one kernel: lighting + tonemap -> sRGB8 (20 B/px)')
    ap.add_argument('--flags', type=int, default=0, help='extra MIL_FLAG_* bits (experiments)')
    return ap.parse_args()


def build_inputs(torch, synth, ubo, wl, device, n_sets, frame_idx=None, n_frames=64, rows=None, noise=False):
    from renderer_amd import sharding
    w, h, n_dir, n_point, shadow_size, ambient = WORKLOADS[wl]
    cam = synth.sweep_camera(w, h) if frame_idx is None else sharding.orbit_camera(frame_idx, n_frames, w, h)
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


def algorithmic(inp, pairs, fused, lit_px=None):
    """Per-launch algorithmic bytes and FLOPs of the lighting kernel.  FLOPs are counted for
    LIT pixels only (occlusion > 0): a cleared / background texel is exactly 0 whatever the lights are, and
    the kernel skips it, so counting it would flatter the VALU fraction."""
    px = inp['w'] * inp.get('lh', inp['h'])
    lit = px if lit_px is None else lit_px
    bytes_ = px * (16 + (4 if fused else 16)) + 32 * (inp['n_dir'] + inp['n_point']) + 272
    if inp['shadow_size']:
        bytes_ += inp['shadow_size'] ** 2 * 4
    flops = lit * C_PIX + lit * inp['n_dir'] * C_DIR_MIN + pairs * C_POINT_MIN
    return bytes_, flops


def kernel_burst(torch, lighting, lp, inp, n, fused, flags, which='light'):
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
        elif fused:
            lp.light_ex(inp['sets'][s], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_SRGB8,
                        flags=flags, out=out[s])
        else:
            lp.light_ex(inp['sets'][s], inp['frame'], inp['dl'], inp['pl'], inp['amb'], inp['sh'], fmt=lighting.OUT_HDR,
                        flags=flags, out=hdr[s])
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        one(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


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


def run_workload(torch, lighting, lp, inp, steps, warmup, fused, flags, dist=None, world=1, rank=0, pipelined=True, gather=True):
    """Returns seconds for `steps` steps [max over ranks]."""
    from renderer_amd import sharding
    dev = lp.device
    n_sets = len(inp['sets'])
    w, h = inp['w'], inp['h']
    strips = inp.get('rows') is not None
    bounds = inp.get('bounds') or (sharding.strip_bounds(h, world) if strips else None)
    lh = (inp['rows'][1] - inp['rows'][0]) if strips else h            # rows this rank shades
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
        y0 = h // 2 - rows // 2          # band through the sphere wall
        strip = {k: (v[y0:y0 + rows].cpu() if hasattr(v, 'cpu') else v) for k, v in inp['sets'][0].items()}
        g = synth.to_numpy(strip)
        t0 = time.perf_counter()
        for _ in range(reps):
            hdr = coracle.shade(g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'], inp['frame'], inp['dl'],
                                point_lights=inp['pl'], ambient=inp['amb'], shadow=sh, y_offset=y0, full_height=h,
                                nthreads=threads)
            coracle.tonemap(hdr, 2, nthreads=threads)
        return time.perf_counter() - t0

    rows = 16
    t = run(rows)
    rows_full = int(min(h, max(rows, rows * target_s / max(t, 1e-6)))) // 16 * 16
    if rows_full > rows:
        rows, t = rows_full, run(rows_full)
    reps = 1
    if t < target_s:                       # a whole frame is quicker than the sample budget: repeat it
        reps = int(min(200, max(1, target_s / max(t, 1e-6))))
        t = run(rows, reps)
    return dict(value=round(reps * rows * w / t / 1e6, 4), unit='Mpix/s', cores=threads, kind='port',
                sample=f'{reps} x {rows} of {h} rows x {w} px (band centred on the sphere wall), same lights/shadow/ambient, '
                       f'lighting + tonemap, {t:.1f} s wall, oracle/mil_oracle.c brute-force over all lights, OpenMP')


def main():
    args = parse()
    import torch
    from renderer_amd import lighting, synth, ubo

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit('for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N '
                     '--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...')
        args.gpus = world
    dist = None
    # rehearsal on a one-GPU box only: MIL_BENCH_REHEARSE=1 puts every rank on device 0 and joins them over gloo
    # (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    rehearse = os.environ.get('MIL_BENCH_REHEARSE') == '1'
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        if rehearse:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    dev = f'cuda:{local_rank}'

    lp = lighting.LightingPass(local_rank, max_point_lights=4096)
    strips = args.mode == 'strips'
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

    dt = run_workload(torch, lighting, lp, inp, args.steps, args.warmup, args.fused, args.flags, dist, world, rank,
                      pipelined=(args.pipelined and world == 1), gather=strips)
    dt_gather = None
    gather_error = None
    if world > 1 and not strips:       # the same frames, each gathered to GPU 0 (xGMI-bound)
        try:
            dt_gather = run_workload(torch, lighting, lp, inp, args.steps, args.warmup, args.fused, args.flags, dist, world, rank,
                                     pipelined=False, gather=True)
        except Exception as e:          # the headline measurement above must still be reported
            gather_error = f'{type(e).__name__}: {e}'[:300]
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    occ = (inp['sets'][0]['albedo_occ'] >> 24) & 0xFF
    lit_px = int((occ != 0).sum().item())
    abytes, aflops = algorithmic(inp, pairs, args.fused, lit_px)
    light_ms = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), args.fused, args.flags)
    tone_burst_ms = 0.0 if args.fused else kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), False, 0, 'tonemap')
    hbm = abytes / (light_ms * 1e-3) / 1e9
    valu = aflops / (light_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    kname = 'k_tiled' if (inp['n_point'] or (inp['shadow_size'] and args.flags & 4)) else 'k_stream'
    if os.path.exists(tpath) and not strips:      # the PMC passes were taken on whole-frame launches
        t = json.load(open(tpath)).get(f'{args.workload}{"_fused" if args.fused else ""}:{kname}')
        if t:
            traffic = t['hbm_bytes_per_launch']
    roof_hbm = dict(bound='hbm', achieved=round(hbm, 1), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(hbm / HBM_PEAK_GBS, 4),
                    traffic=traffic, kernel=kname, ms=round(light_ms, 4), algorithmic_bytes=abytes)
    roof_valu = dict(bound='valu', achieved=round(valu, 2), peak=VALU_PEAK_TFLOPS, unit='TFLOP/s',
                     frac=round(valu / VALU_PEAK_TFLOPS, 4), traffic=traffic, kernel=kname, ms=round(light_ms, 4),
                     algorithmic_flops=aflops, pairs_in_radius=pairs, lit_pixels=lit_px)
    # the binding roofline is the one the kernel sits closer to
    primary, other = (roof_hbm, roof_valu) if roof_hbm['frac'] >= roof_valu['frac'] else (roof_valu, roof_hbm)

    res = dict(metric='shaded Mpix/s at 3840x2160, 256 point lights + PCF shadows; % HBM/VALU roofline',
               value=round((1 if strips else world) * args.steps * px / dt / 1e6, 1), unit='Mpix/s', n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 4), higher_is_better=True,
               scaling=('strong' if strips else 'weak'),
               vs_baseline=None, dtype='f32', data='synthetic',
               config=dict(workload=f'{args.workload}: {inp["w"]}x{inp["h"]} Scene A sweep, {inp["n_dir"]} directional '
                                    f'(light 0 PCF {inp["shadow_size"]}^2) + {inp["n_point"]} point lights'
                                    + (' + ambient' if inp['amb'] else ''),
                           kernels=(['k_tiled+tonemap fused -> sRGB8'] if args.fused else [f'{kname} -> hdr RGBA32F', 'k_tonemap -> sRGB8']),
                           buffer_sets=args.sets, parallelism=('single GPU' if world == 1 else (f'{world} horizontal strips (16-row aligned{", cost-weighted " + str([b - a for a, b in inp["bounds"]]) if inp.get("bounds") else ""}) + P2P gather to GPU 0' if strips else f'frame-parallel x{world}, frames stay on their GPUs')),
                           tonemap_ms=round(tone_burst_ms, 4),
                           streams=('1 (serial)' if (not args.pipelined or args.fused or world > 1) else '2: tonemap of frame i beside lighting of frame i+1'),
                           kernel_ms_method='burst of back-to-back launches between two HIP events on the launch stream'),
               roofline=primary, roofline_other=other)
    if gather_error is not None:
        res['config']['gather_to_gpu0'] = dict(error=gather_error)
    if dt_gather is not None:
        res['config']['gather_to_gpu0'] = dict(value=round(world * args.steps * px / dt_gather / 1e6, 1), unit='Mpix/s',
                                               ms_per_step=round(dt_gather / args.steps * 1e3, 4),
                                               note='same loop with every packed sRGB8 frame (33.2 MB) sent to GPU 0 by grouped P2P over xGMI')

    if not args.no_extra and world == 1 and args.workload == 'headline':
        # side measurement: the HBM-bound 4K / 1-light lighting pass (BASELINE.md target >= 70 % of HBM roofline)
        inp1 = build_inputs(torch, synth, ubo, 'k1_1light', dev, args.sets)
        l1 = kernel_burst(torch, lighting, lp, inp1, max(args.steps, 20), False, 0)
        b1, f1 = algorithmic(inp1, 0, False)
        res['extra'] = {'k1_4k_1light': dict(kernel='k_stream -> hdr RGBA32F', ms=round(l1, 4), mpix_s=round(px / l1 / 1e3, 1),
                                             hbm_GBs=round(b1 / l1 / 1e6, 1), hbm_frac=round(b1 / l1 / 1e6 / HBM_PEAK_GBS, 4),
                                             algorithmic_bytes=b1)}
        del inp1
        # the same pass on Scene B (noise, every texel lit): no wave can skip anything, so this is the pure streaming rate
        inp1 = build_inputs(torch, synth, ubo, 'k1_1light', dev, args.sets, noise=True)
        l1 = kernel_burst(torch, lighting, lp, inp1, max(args.steps, 20), False, 0)
        res['extra']['k1_4k_1light_all_lit'] = dict(kernel='k_stream -> hdr RGBA32F, Scene B', ms=round(l1, 4), mpix_s=round(px / l1 / 1e3, 1),
                                                    hbm_GBs=round(b1 / l1 / 1e6, 1), hbm_frac=round(b1 / l1 / 1e6 / HBM_PEAK_GBS, 4),
                                                    note='tools/probe/stream_probe.hip: the same four-plane access pattern with no arithmetic takes 43.6 us (6.09 TB/s), a float4 copy 41.6 us')
        del inp1
        # side measurement: lighting + tonemap as ONE launch (hdr stays in registers; what mil_shade_frame does when the frame has
        # no sky / transparency subpass): same output as the two-kernel step that `value` times
        if not args.fused:
            lf = kernel_burst(torch, lighting, lp, inp, max(args.steps, 20), True, args.flags)
            res['extra']['headline_one_launch'] = dict(kernel='k_tiled+tonemap -> sRGB8', ms=round(lf, 4), mpix_s=round(px / lf / 1e3, 1))
        # side measurement: the reference's whole frame, G-buffer -> swapchain image (render/mod.rs:189-329: lighting, sky,
        # transparency resolve, tonemap, FXAA), as the five-call chain and as mil_shade_frame + mil_fxaa (hdr never stored)
        res['extra']['frame_4k_gbuffer_to_swapchain'] = frame_chain(torch, lighting, synth, lp, inp, max(args.steps, 20))
        # side measurement: the accumulate draw's fragment shader over one full-screen translucent layer
        res['extra']['transparency_accumulate_4k_layer'] = accumulate_burst(torch, synth, lp, inp, max(args.steps, 20), px)
    if not args.no_cpu and world == 1:
        res['cpu_baseline'] = cpu_baseline(inp)
    print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
