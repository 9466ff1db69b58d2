"""Sum of the half-tile list lengths the two-kernel path builds (sizes the list pool of the workspace): upper bound
2 x the per-tile union count k_bin reports, for the 4K and 8K frames at 256 / 1024 / 4096 lights."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo

lp = lighting.LightingPass(0, max_point_lights=4096)
lp.set_two_kernel_threshold(1)
for (w, h) in ((3840, 2160), (7680, 4320)):
    cam = synth.sweep_camera(w, h)
    gb = synth.scene_sweep(w, h, cam, device='cuda:0')
    frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
    for n in (256, 1024, 4096):
        pl = synth.point_light_field(n, 0x5EED0003)
        cnt = lp.tile_counts(gb)
        lp.light_ex(gb, frame, dl, pl, (0.03, 0.03, 0.03), None, fmt=lighting.OUT_SRGB8)
        torch.cuda.synchronize()
        c = cnt.cpu().numpy().astype('int64')
        print(f'{w}x{h} {n} lights: union entries per tile sum {c.sum()} max {c.max()} tiles {c.size} lit tiles {(c>0).sum()}', flush=True)
        lp.tile_counts(gb, enable=False)
