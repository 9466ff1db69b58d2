"""Strip balance on ONE GPU: every strip of an n-way split is timed on its own (lighting -> hdr + tonemap -> sRGB8),
for equal-row cuts and for cuts weighted by the previous frame's wave lifetimes (sharding.strip_bounds_weighted).
The slowest strip bounds the strong-scaling speed-up of bench.py --mode strips before the gather."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, synth, ubo, sharding
W, H, NPT = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (7680, 4320, 4096)
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdr = torch.empty((H, W, 4), device=dev); out = torch.empty((H, W), dtype=torch.int32, device=dev)

def timed(y0, y1, n=10):
    strip = {k: (v[y0:y1] if torch.is_tensor(v) else v) for k, v in gb.items()}
    strip['y_offset'] = y0; strip['width'] = W; strip['height'] = H
    def f():
        lp.light_ex(strip, frame, dl, pl, amb, sh, out=hdr[y0:y1]); lp.tonemap(hdr[y0:y1], 2, out=out[y0:y1])
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

full = timed(0, H)
st = lp.tile_stamps(gb); lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr); torch.cuda.synchronize(); lp.tile_stamps(gb, enable=False)
cost = sharding.band_costs_from_stamps(st)
print(f'{W}x{H}, {NPT} point lights: whole frame {full:.1f} us')
for n in (2, 4, 8):
    c = cost
    for name, it in (('equal rows', -1), ('weighted 0', 0), ('weighted 1', 1), ('weighted 2', 2), ('weighted 3', 3)):
        b = sharding.strip_bounds(H, n) if it < 0 else sharding.strip_bounds_weighted(c, n, H)
        t = [timed(y0, y1) if y1 > y0 else 0.0 for y0, y1 in b]
        print(f'  n={n} {name}: slowest strip {max(t):8.1f} us (fastest {min(t):7.1f}) -> speed-up before gather {full / max(t):4.2f}x   rows {[y1 - y0 for y0, y1 in b]}')
        if it >= 0: c = sharding.rescale_band_costs(c, b, t)          # what every rank would do with its own measured time
