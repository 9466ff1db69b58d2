"""Wave-lifetime diagnostics of k_shade (two-kernel tiled path): per half-tile start / end stamps. Run with MIL_DEBUG=1 MIL_TUNE=2048."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
NPT = int(os.environ.get('NPT', '256'))
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdr = torch.empty((H, W, 4), device=dev)
for _ in range(3): lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr)
torch.cuda.synchronize()
nt = (H // 16) * (W // 16)
st = torch.zeros((2 * nt, 4), dtype=torch.int32, device=dev)
lp._check(lp.lib.mil_set_tile_stamp_buffer(lp.ctx, st.data_ptr(), 2 * nt))
lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr)
torch.cuda.synchronize()
lp._check(lp.lib.mil_set_tile_stamp_buffer(lp.ctx, None, 0))
s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
used = s[:, 1] != 0
s = s[used]
t0, t1, ln, idx = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
base = t0.min()
a, b = (t0 - base) / 100.0, (t1 - base) / 100.0
life = b - a
print(f'items {len(a)}; span {b.max():.1f} us; lifetime mean {life.mean():.2f} p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} p99 {np.percentile(life, 99):.2f} max {life.max():.2f}')
print(f'sum of lifetimes {life.sum():.0f} wave-us -> mean concurrency {life.sum() / b.max():.0f}')
o = np.argsort(idx)
print('by launch order (deciles): start time / lifetime / list length')
for q in range(10):
    sl = o[len(o) * q // 10: len(o) * (q + 1) // 10]
    print(f'   {q}: start {a[sl].mean():7.1f}  life {life[sl].mean():7.2f}  len {ln[sl].mean():6.1f}')
edges = np.arange(0, b.max() + 1, b.max() / 40)
ev = np.concatenate([np.stack([a, np.ones_like(a)], 1), np.stack([b, -np.ones_like(b)], 1)])
ev = ev[np.argsort(ev[:, 0], kind='stable')]
c = np.cumsum(ev[:, 1]); tt = ev[:, 0]
conc = [c[np.searchsorted(tt, e, side='right') - 1] if e >= tt[0] else 0 for e in edges]
print('concurrency at 40 points:', ' '.join(str(int(v)) for v in conc))
