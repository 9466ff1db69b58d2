"""Per-XCD view of a k_tiled launch (persistent work-queue experiment): when each XCD's waves ran out of tiles."""
import os, sys
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
st = lp.tile_stamps(gb)
lp.light_ex(gb, frame, dl, pl, amb, sh, out=hdr)
torch.cuda.synchronize()
lp.tile_stamps(gb, enable=False)
s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
t0, t1, cnt, hw = s[..., 0].ravel(), s[..., 1].ravel(), s[..., 2].ravel(), s[..., 3].ravel()
base = t0.min(); a, b = (t0 - base) / 100.0, (t1 - base) / 100.0
xcc = (hw >> 16) & 0xF
print(f'span {b.max():.1f} us')
for x in range(8):
    m = xcc == x
    if not m.any(): continue
    life = (b - a)[m]
    print(f'XCD {x}: tiles {m.sum():6d}  lit-with-lights {int((cnt[m] > 0).sum()):5d}  sum of tile times {life.sum():8.0f} us  first start {a[m].min():6.1f}  last end {b[m].max():6.1f}  90 % of tiles done at {np.percentile(b[m], 90):6.1f}')
slot = (xcc << 16) | (hw & 0xFFFF)
ends = {}
for sl, e in zip(slot.tolist(), b.tolist()): ends[sl] = max(ends.get(sl, 0), e)
e = np.array(sorted(ends.values()))
print('wave slots', len(e), 'last-tile end per slot: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f' % tuple(np.percentile(e, [10, 50, 90, 99, 100])))

ctr = (hw >> 20) & 0x3F
lit = hw != 0
print('per counter (lit tiles only): tiles, first start, last end')
for c in range(32):
    m = lit & (ctr == c)
    if m.any(): print(f'  counter {c:2d}: {m.sum():5d} tiles, xcc {sorted(set(xcc[m].tolist()))}, last end {b[m].max():6.1f}, sum {((b - a)[m]).sum():7.0f} us')
