"""Ad-hoc: K1 (1 directional light -> hdr) at 4K on Scene B (noise: every texel lit, nothing to skip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
lp = lighting.LightingPass(0, 16)
cam = synth.sweep_camera(W, H); frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()[:int(os.environ.get('NDIR', '1'))]
for name, mk in (('sweep', lambda s: synth.scene_sweep(W, H, cam, device='cuda:0')), ('noise', lambda s: synth.scene_noise(W, H, 0x5EED0000 + s, device='cuda:0'))):
    gbs = [mk(i) for i in range(3)]
    hdrs = [torch.empty((H, W, 4), device='cuda:0') for _ in range(3)]
    outs = [torch.empty((H, W), dtype=torch.int32, device='cuda:0') for _ in range(3)]
    for fmt in (0, 1):
        dst = hdrs if fmt == 0 else outs
        for i in range(5): lp.light_ex(gbs[i % 3], frame, dl, fmt=fmt, out=dst[i % 3])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(30): lp.light_ex(gbs[i % 3], frame, dl, fmt=fmt, out=dst[i % 3])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        b = W * H * (16 + (16 if fmt == 0 else 4))
        print(f'{name} fmt {fmt}: {us:.1f} us  {b / us / 1e6:.2f} TB/s algorithmic')
