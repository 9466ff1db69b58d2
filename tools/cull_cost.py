"""Ad-hoc: what does the culling itself cost?  4096 point lights moved out of reach (empty lists) vs none vs real."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
def t(pl, n=20):
    def f(i): lp.light_ex(gbs[i % 3], frame, dl, pl, amb, sh, out=hdrs[i % 3])
    for i in range(4): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): f(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for n in (256, 1024, 4096):
    pl = synth.point_light_field(n, 0x5EED0003)
    far = pl.copy(); far[:, 1] += 1000.0
    one = far[:1].copy()
    print(f'{n} lights: real {t(pl):.1f} us; all out of reach {t(far):.1f} us; a single light out of reach {t(one):.1f} us')
