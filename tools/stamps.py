"""Wave-lifetime diagnostics of k_tiled on the headline frame (per-tile start/end stamps, 100 MHz)."""
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
t0, t1, cnt = s[..., 0].ravel(), s[..., 1].ravel(), s[..., 2].ravel()
base = t0.min()
a, b = (t0 - base) / 100.0, (t1 - base) / 100.0      # us
life = b - a
print(f'kernel span {b.max():.1f} us; waves {len(a)}; lifetime mean {life.mean():.2f} p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} p99 {np.percentile(life, 99):.2f} max {life.max():.2f} us')
print(f'sum of lifetimes {life.sum():.0f} wave-us -> mean concurrency {life.sum() / b.max():.0f} waves ({life.sum() / b.max() / 256:.1f} per CU)')
for lo, hi in [(0, 0), (1, 4), (5, 16), (17, 64), (65, 9999)]:
    m = (cnt >= lo) & (cnt <= hi)
    if m.any(): print(f'  list {lo:3d}..{hi:4d}: {m.sum():6d} tiles, lifetime mean {life[m].mean():6.2f} us, share of wave-time {life[m].sum() / life.sum() * 100:5.1f} %')
edges = np.arange(0, b.max() + 4, 4.0)
conc = [((a < e + 4) & (b > e)).sum() for e in edges]
print('concurrency per 4 us bin:', ' '.join(str(c) for c in conc))
ev = np.concatenate([np.stack([a, np.ones_like(a)], 1), np.stack([b, -np.ones_like(b)], 1)])
ev = ev[np.argsort(ev[:, 0], kind='stable')]
c = np.cumsum(ev[:, 1]); tt = ev[:, 0]
exact = [np.average(c[:-1][(tt[:-1] >= e) & (tt[:-1] < e + 4)], weights=np.maximum(np.diff(tt)[(tt[:-1] >= e) & (tt[:-1] < e + 4)], 1e-9)) if ((tt[:-1] >= e) & (tt[:-1] < e + 4)).any() else 0 for e in edges]
print('exact mean concurrency per 4 us bin:', ' '.join(f'{v:.0f}' for v in exact), '| peak', int(c.max()))
hw = s[..., 3].ravel()
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh_ = (hw >> 12) & 1; simd = (hw >> 4) & 3; wid = hw & 0xF
xcc = (hw >> 16) & 0xF
slot = (xcc << 16) | (hw & 0xFFFF)
order = np.lexsort((a, slot))
ss, aa, bb = slot[order], a[order], b[order]
same = ss[1:] == ss[:-1]
gaps = (aa[1:] - bb[:-1])[same]
print(f'slots seen {len(set(slot.tolist()))}; relaunch gap per slot: mean {gaps.mean():.2f} p50 {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} min {gaps.min():.2f} us; negative {np.mean(gaps < 0):.3f}')
print('distinct (se,sh,cu):', len(set(zip(se.tolist(), sh_.tolist(), cu.tolist()))), 'wave_id max', wid.max(), 'simd values', sorted(set(simd.tolist())))
print('start time by tile row (first, last row starts):', a.reshape(s.shape[:2])[0].mean().round(1), a.reshape(s.shape[:2])[-1].mean().round(1))
rows = life.reshape(s.shape[:2]).mean(1)
print('mean lifetime per tile row (every 9th):', ' '.join(f'{v:.1f}' for v in rows[::9]))
