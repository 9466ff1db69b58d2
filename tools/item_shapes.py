"""Lane efficiency of k_shade's items by item SHAPE (4K, 4096 lights): an item (128 pixels = one wave, two pixels per lane)
evaluates a light for all its pixels when one is in range.  efficiency = pairs in range / (128 x (item, light) pairs with any
pixel in range); list entries = those pairs (a lower bound of what culling by boxes lists)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import synth, ubo
W, H = 3840, 2160
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = 'cuda:0'
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data()
pl = synth.point_light_field(N, 0x5EED0003)
f = torch.tensor(np.asarray(frame, np.float64), device=dev)
ip, iv = f[32:48].reshape(4, 4).T, f[48:64].reshape(4, 4).T
ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float64), torch.arange(W, device=dev, dtype=torch.float64), indexing='ij')
clip = torch.stack([(xs + 0.5) / W * 2 - 1, (ys + 0.5) / H * 2 - 1, gb['depth'].double(), torch.ones_like(xs)], -1)
v = clip @ ip.T
v = v / v[..., 3:4]
P = (v @ iv.T)[..., :3].float()
lit = (((gb['albedo_occ'] >> 24) & 0xFF) > 0)
shapes = [(16, 8), (32, 4), (64, 2), (8, 16), (16, 16), (128, 1)]
items_hit = {s: 0 for s in shapes}
pairs = 0
for l in pl:
    pos = torch.tensor(l[0:3], device=dev)
    r2 = float(np.float32(l[3]) * np.float32(l[3]))
    inr = (((P - pos) ** 2).sum(-1) < r2) & lit
    pairs += int(inr.sum())
    for (sw, shh) in shapes:
        items_hit[(sw, shh)] += int(inr.reshape(H // shh, shh, W // sw, sw).any(3).any(1).sum())
print(f'{N} lights: pairs in range {pairs}')
for (sw, shh) in shapes:
    n = items_hit[(sw, shh)]
    print(f'item {sw:3d} x {shh:2d}: (item, light) pairs {n:9d}  lane efficiency {pairs / (n * sw * shh):.3f}  lane-evaluations {n * sw * shh / 1e6:8.1f} M')
