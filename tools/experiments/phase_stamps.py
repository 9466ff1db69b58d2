import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(256, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdr = torch.empty((H, W, 4), device=dev)
for _ in range(3): lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr)
torch.cuda.synchronize()
st = lp.tile_stamps(gb)
lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr)
torch.cuda.synchronize()
s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
t0, t1 = s[..., 0].ravel(), s[..., 1].ravel()
tA, tB = (s[..., 2].ravel() & 0xFFFF) / 100.0, (s[..., 2].ravel() >> 16) / 100.0
tC, tD = (s[..., 3].ravel() & 0xFFFF) / 100.0, (s[..., 3].ravel() >> 16) / 100.0
life = ((t1 - t0) & 0xFFFFFFFF) / 100.0
lit = tB > 0
print(f'lit tiles {lit.sum()}: lifetime {life[lit].mean():.2f} us = albedo round trip {tA[lit].mean():.2f} + other planes {np.mean(tB[lit]-tA[lit]):.2f} + box {np.mean(tC[lit]-tB[lit]):.2f} + cull {np.mean(tD[lit]-tC[lit]):.2f} + two passes and stores {np.mean(life[lit]-tD[lit]):.2f}')
print(f'unlit tiles {(~lit).sum()}: lifetime {life[~lit].mean():.2f} us')
