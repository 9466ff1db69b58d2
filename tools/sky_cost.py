"""What the sky tiles cost the headline launch: the dual step on the whole 4K frame against the same step on the strip of tile
rows that hold a lit pixel (the sky rows simply not launched)."""
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
lit_rows = ((((gb['albedo_occ'] >> 24) & 0xFF) != 0).sum(1).view(-1, 16).sum(1) > 0).cpu().numpy()
r0 = int(lit_rows.argmax()) * 16
r1 = (len(lit_rows) - int(lit_rows[::-1].argmax())) * 16
print(f'tile rows with a lit pixel: {int(lit_rows.sum())} of {len(lit_rows)}; frame rows {r0}..{r1}')
strip = {k: (v[r0:r1].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
strip['y_offset'] = r0
hdr = torch.empty((H, W, 4), device=dev); out = torch.empty((H, W), dtype=torch.int32, device=dev)
def full(): lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr, ldr_out=out)
def part(): lp.light_ex(strip, frame, dl, pl, amb, sh, out=hdr[r0:r1], ldr_out=out[r0:r1])
for name, fn in (('whole frame', full), ('lit rows only', part), ('whole frame', full), ('lit rows only', part)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f'{name:14s} {statistics.median(ts):7.1f} us')
