"""One rank's share of a frame (bands dealt cyclically), N launches: for rocprofv3 passes and wave-lifetime stamps.
usage: rank_case.py W H NPT n r band [iters] [stamps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, multigpu, synth, ubo
W, H, NPT, n, r, band = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 8
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
rows = torch.from_numpy(multigpu.frame_rows_of_rank(H, n, r, band_rows=band)).to(dev)
sub = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
sub['y_offset'] = 0
del gb
out = torch.empty((len(rows), W), dtype=torch.int32, device=dev)
def one(): lp.shade_bands(sub, frame, dl, (band, n, r, 0), pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out)
for _ in range(iters): one()
torch.cuda.synchronize()
if len(sys.argv) > 8:
    nt = (len(rows) // 16 + 1) * (W // 16)
    st = torch.zeros((2 * nt, 4), dtype=torch.int32, device=dev)
    lp._check(lp.lib.mil_set_tile_stamp_buffer(lp.ctx, st.data_ptr(), 2 * nt))
    one(); torch.cuda.synchronize()
    lp._check(lp.lib.mil_set_tile_stamp_buffer(lp.ctx, None, 0))
    s = st.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    s = s[s[:, 1] != 0]
    t0, t1, ln, idx = s[:, 0], s[:, 1], s[:, 2], s[:, 3]
    a, b = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0
    life = b - a
    print(f'items {len(a)}; span {b.max():.1f} us; lifetime mean {life.mean():.2f} p50 {np.median(life):.2f} p90 {np.percentile(life, 90):.2f} p99 {np.percentile(life, 99):.2f} max {life.max():.2f}; sum {life.sum():.0f} wave-us')
    o = np.argsort(idx)
    for q in range(10):
        sl = o[len(o) * q // 10: len(o) * (q + 1) // 10]
        print(f'   decile {q}: start {a[sl].mean():7.1f}  end {b[sl].max():7.1f} life {life[sl].mean():7.2f} max {life[sl].max():7.2f} len {ln[sl].mean():6.1f} maxlen {ln[sl].max()}')
    top = np.argsort(-life)[:8]
    print('longest items:', [(round(float(a[i]), 1), round(float(life[i]), 1), int(ln[i]), int(idx[i])) for i in top])
