"""Strip / band balance on ONE GPU: every rank's share of one frame is timed on its own (lighting fused with
the sRGB8 encode, what mil_mg_render_strips launches), for contiguous 16-row-aligned strips and for bands dealt cyclically.
The slowest rank bounds the strong-scaling speed-up before the gather.  Also: the cost of cutting a rank's launch into chunks
(mil_mg's gather overlap).  usage: band_balance.py [W H NPT]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, multigpu, synth, ubo
W, H, NPT = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7680, 4320, 4096)
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3

def timed(f, n=8):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

out_full = torch.empty((H, W), dtype=torch.int32, device=dev)
full = timed(lambda: lp.light_ex(gb, frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out_full))
print(f'{W}x{H}, {NPT} point lights, fused sRGB8 store: whole frame {full:.1f} us')
for n in (2, 4, 8):
    for band in (0, 16, 32, 64, 128):
        ts = []
        for r in range(n):
            rows = torch.from_numpy(multigpu.frame_rows_of_rank(H, n, r, band_rows=band)).to(dev)
            sub = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
            out = torch.empty((len(rows), W), dtype=torch.int32, device=dev)
            if band:
                sub['y_offset'] = 0
                ts.append(timed(lambda: lp.shade_bands(sub, frame, dl, (band, n, r, 0), pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out)))
            else:
                sub['y_offset'] = int(rows[0].item())
                ts.append(timed(lambda: lp.light_ex(sub, frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out)))
        print(f'  n={n} {"contiguous strips" if not band else f"bands of {band:3d} rows "}: slowest rank {max(ts):7.1f} us, fastest {min(ts):7.1f} -> '
              f'speed-up before gather {full / max(ts):4.2f}x')
# chunking cost for rank 0 of 8, bands of 32: the rank's rows as 1, 2, 4, 8 launches on one stream
n, band, r = 8, 32, 0
rows = torch.from_numpy(multigpu.frame_rows_of_rank(H, n, r, band_rows=band)).to(dev)
sub = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
sub['y_offset'] = 0
out = torch.empty((len(rows), W), dtype=torch.int32, device=dev)
nb = len(rows) // band
for chunks in (1, 2, 4, 8):
    cuts = [nb * k // chunks for k in range(chunks + 1)]
    def f():
        for a, b in zip(cuts[:-1], cuts[1:]):
            part = {k: (v[a * band:b * band] if torch.is_tensor(v) else v) for k, v in sub.items()}
            lp.shade_bands(part, frame, dl, (band, n, r, a), pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out[a * band:b * band])
    print(f'  rank 0 of 8, bands of 32 rows, {chunks} launch(es) on one stream: {timed(f):7.1f} us')
