"""Times mil_shade_frame (lighting + sky + resolve + tonemap -> R11G11B10F, one call) on the 4K sweep frame with NPT point lights,
with the two-kernel form allowed (default threshold) and forbidden.  usage: time_frame.py [NPT]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
NPT = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
prm = synth.hosek_params_synthetic(); ta, tr = synth.transparency_layer(W, H, dev)
out = torch.empty((H, W), dtype=torch.int32, device=dev)
def one(): lp.shade_frame(gb, frame, dl, pl, amb, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_LDR, out=out)
res = {}
for name, thr in (('two-kernel allowed', 1024), ('k_tiled only', 0xFFFFFFFF)):
    lp.set_two_kernel_threshold(thr)
    for _ in range(3): one()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): one()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
    res[name] = best
    ref = out.clone() if name.startswith('two') else ref
    if not name.startswith('two'): print('identical:', bool(torch.equal(ref, out)))
print(f'mil_shade_frame 4K, {NPT} point lights:', ', '.join(f'{k} {v:.1f} us' for k, v in res.items()))
