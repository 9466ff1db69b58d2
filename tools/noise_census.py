"""Tolerance census of the fast path on Scene B (i.i.d. noise planes: every roughness byte incl. 0 -> alpha 1e-8, normals and depths
unrelated): pixels outside the bare 1e-3 + 1e-4 |b| of the float64 oracle, and how many of them are the fast path's own
(the reference-order FP32 result inside the bar)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import coracle
from renderer_amd import lighting, synth, ubo
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
coracle.build()
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_noise(W, H, 0x5EED0001, device='cuda:0')
g = synth.to_numpy(gb)
planes = (g['normal'], g['albedo_occ'], g['rough_metal'], g['depth'])
frame, dl = cam.frame_matrix_data(), ubo.default_directional_lights()
for n in (0, 64):
    pl = synth.point_light_field(n, 5) if n else None
    r64, cond = coracle.shade_f64(*planes, frame, dl, point_lights=pl, ambient=(0.03,) * 3, want_cond=True)
    ref32 = coracle.shade(*planes, frame, dl, point_lights=pl, ambient=(0.03,) * 3)
    fast = lp.light_ex(gb, frame, dl, pl, (0.03,) * 3, None).cpu().numpy()
    b = np.abs(r64[..., :3])
    fin = np.isfinite(r64[..., :3]).all(-1) & np.isfinite(ref32[..., :3]).all(-1)
    err = np.abs(fast[..., :3].astype(np.float64) - r64[..., :3])
    bare = (err > 1e-3 + 1e-4 * b).any(-1) & fin
    refbad = (np.abs(ref32[..., :3].astype(np.float64) - r64[..., :3]) > 1e-3 + 1e-4 * b).any(-1) & fin
    withc = (err > 1e-3 + 1e-4 * b + 8 * 2.0 ** -24 * cond).any(-1) & fin
    print(f'{W}x{H} noise, {n} point lights: finite {int(fin.sum())}; fast outside bare {int(bare.sum())}; reference-order FP32 outside bare {int(refbad.sum())}; '
          f'own {int((bare & ~refbad).sum())}; outside bar + 8 ulp cond {int(withc.sum())}; non-finite fast {int((~np.isfinite(fast)).any(-1).sum())}', flush=True)
