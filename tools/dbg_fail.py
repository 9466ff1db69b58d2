"""Prints the pixels of the K1 noise-scene fast-tolerance test that fall outside the tolerance."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import mil_oracle as mo
from renderer_amd import lighting, synth, ubo
w, h = 256, 144
cam = synth.sweep_camera(w, h)
gb = synth.scene_noise(w, h, 0x5EED0000, device='cuda:0')
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
lp = lighting.LightingPass(0, 64)
out = lp.light_directional(gb, frame, dl).cpu().numpy()
st = lp.light_directional(gb, frame, dl, strict=True).cpu().numpy()
g = synth.to_numpy(gb)
planes = (g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'])
ref, cond = mo.shade(*planes, frame, dl, want_cond=True)
tol = 1e-3 + 1e-4 * np.abs(ref[..., :3]) + 8 * 2.0 ** -24 * cond
err = np.abs(out[..., :3].astype(np.float64) - ref[..., :3])
errs = np.abs(st[..., :3].astype(np.float64) - ref[..., :3])
bad = np.argwhere((err > tol).any(-1))
print('n bad', len(bad), 'strict-vs-f64 bad', ((errs > tol).any(-1)).sum())
worst = np.argsort((err / tol).max(-1).ravel())[-8:]
for k in worst:
    y, x = divmod(int(k), w)
    print(y, x, 'ref', ref[y, x, :3], 'fast', out[y, x, :3], 'strict', st[y, x, :3], 'tol', tol[y, x], 'cond', cond[y, x],
          'n=%08x a=%08x rm=%08x d=%r' % (planes[0][y, x], planes[1][y, x], planes[2][y, x], planes[3][y, x]))
