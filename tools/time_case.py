"""Times the lighting launch (HIP-event burst) on the 4K sweep frame: NPT point lights, optional MIL_DEBUG/MIL_TUNE env.
usage: time_case.py [NPT] [fmt: hdr|srgb8] [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
NPT = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fmt = lighting.OUT_SRGB8 if (len(sys.argv) > 2 and sys.argv[2] == 'srgb8') else lighting.OUT_HDR
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (3840, 2160)
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
outs = [torch.empty((H, W, 4), device=dev) if fmt == lighting.OUT_HDR else torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(3)]
def one(i): lp.light_ex(gbs[i % 3], frame, dl, pl, amb, sh, fmt=fmt, out=outs[i % 3])
for i in range(5): one(i)
torch.cuda.synchronize()
res = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30 if NPT <= 512 else 10
    e0.record()
    for i in range(n): one(i)
    e1.record(); torch.cuda.synchronize()
    res.append(e0.elapsed_time(e1) / n * 1e3)
print(f'NPT={NPT} {W}x{H} tune={os.environ.get("MIL_TUNE", "0")}: {min(res):.1f} us (runs {[round(r, 1) for r in res]})')
