"""Ad-hoc: how conservative are K2's tile lists on the headline frame?  Per (tile, list entry): does any pixel of the tile
lie in the light's radius, and how many."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
pl = synth.point_light_field(256, 0x5EED0003)
counts = lp.tile_counts(gb)
lp.light_ex(gb, frame, dl, pl, (0.03,) * 3, None)
torch.cuda.synchronize()
cnt = counts.cpu().numpy().astype(np.int64).reshape(H // 16, W // 16)
lp.tile_counts(gb, enable=False)
# world positions in float64 on the GPU, as PositionFromDepth
f = torch.tensor(np.asarray(frame, np.float64), device=dev)
ip, iv = f[32:48].reshape(4, 4).T, f[48:64].reshape(4, 4).T
ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64), indexing='ij')
clip = torch.stack([(xs + 0.5) / W * 2 - 1, (ys + 0.5) / H * 2 - 1, gb['depth'].double(), torch.ones_like(xs)], -1)
v = clip @ ip.T
v = v / v[..., 3:4]
P = (v @ iv.T)[..., :3]
occ = ((gb['albedo_occ'] >> 24) & 0xFF) > 0
lit = (gb['depth'] < 1.0) & occ
tiles_hit = torch.zeros((H // 16, W // 16), dtype=torch.int64, device=dev)
pairs = 0
hist = torch.zeros(257, dtype=torch.int64, device=dev)
for l in pl:
    pos = torch.tensor(l[0:3].astype(np.float64), device=dev)
    r2 = float(np.float32(l[3]) * np.float32(l[3]))
    inr = (((P - pos) ** 2).sum(-1) < r2) & lit
    per_tile = inr.reshape(H // 16, 16, W // 16, 16).sum((1, 3))
    tiles_hit += (per_tile > 0)
    pairs += int(inr.sum())
    hist += torch.bincount(per_tile.reshape(-1), minlength=257)
hit = tiles_hit.cpu().numpy()
print(f'pairs in radius {pairs}; list entries {cnt.sum()}; (tile, light) with any pixel in radius {hit.sum()} -> {cnt.sum() / max(hit.sum(), 1):.2f} x conservative')
print(f'tiles with a list {np.count_nonzero(cnt)}, mean length {cnt[cnt > 0].mean():.1f}, max {cnt.max()}; pixel fill of the hit (tile, light) pairs: {pairs / (256.0 * hit.sum()):.2f}')
lit_tiles = lit.reshape(H // 16, 16, W // 16, 16).any(3).any(1).cpu().numpy()
print(f'lit tiles {lit_tiles.sum()} of {lit_tiles.size}; entries in lists of tiles with no lit pixel {cnt[~lit_tiles].sum()}')
