"""Fixed cost against marginal cost of the headline launch: the dual step on the lit rows of the 4K frame as one strip, as two
halves and as four quarters.  If a launch were a fixed ramp + tail `a` plus a rate times its tiles, the parts add up to the
whole plus (parts - 1) * a."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
pl = synth.point_light_field(256, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
lit = (((gb['albedo_occ'] >> 24) & 0xFF) != 0)
lit_tiles_row = (lit.view(H // 16, 16, W // 16, 16).sum((1, 3)) > 0).sum(1).cpu().numpy()
rows = lit_tiles_row > 0
t0 = int(rows.argmax()); t1 = len(rows) - int(rows[::-1].argmax())
print(f'lit tile rows {t0}..{t1}, lit tiles {int(lit_tiles_row.sum())}')
hdr = torch.empty((H, W, 4), device=dev); out = torch.empty((H, W), dtype=torch.int32, device=dev)
dual = os.environ.get('DUAL', '1') == '1'


def strip_fn(a, b):
    r0, r1 = a * 16, b * 16
    s = {k: (v[r0:r1].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
    s['y_offset'] = r0
    if dual:
        return lambda: lp.light_ex(s, frame, dl, pl, amb, sh, out=hdr[r0:r1], ldr_out=out[r0:r1])
    return lambda: lp.light_ex(s, frame, dl, pl, amb, sh, out=hdr[r0:r1])


def time(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    return statistics.median(ts)


whole = time(strip_fn(t0, t1))
print(f'whole strip {whole:7.1f} us')
for parts in (2, 4, 8):
    cuts = [t0 + (t1 - t0) * i // parts for i in range(parts + 1)]
    ts = [time(strip_fn(cuts[i], cuts[i + 1])) for i in range(parts)]
    tl = [int(lit_tiles_row[cuts[i]:cuts[i + 1]].sum()) for i in range(parts)]
    a = (sum(ts) - whole) / (parts - 1)
    print(f'{parts} parts: ' + ' '.join(f'{t:.1f}us/{n}' for t, n in zip(ts, tl)) + f'  sum {sum(ts):.1f}  fixed cost per launch {a:.1f} us')
print(f'whole strip again {time(strip_fn(t0, t1)):7.1f} us')
