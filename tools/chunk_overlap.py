"""One rank's rows (rank 0 of 8, bands of 32 rows, 8K / 4096 lights) as several launches on SEPARATE streams of descending
priority: do the chunks overlap without a tail between them, and when does the first chunk finish?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, multigpu, synth, ubo
W, H, NPT = 7680, 4320, 4096
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device=dev)
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
n, band, r = 8, 32, 0
rows = torch.from_numpy(multigpu.frame_rows_of_rank(H, n, r, band_rows=band)).to(dev)
sub = {k: (v[rows].contiguous() if torch.is_tensor(v) else v) for k, v in gb.items()}
sub['y_offset'] = 0
del gb
out = torch.empty((len(rows), W), dtype=torch.int32, device=dev)
nb = len(rows) // band
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (1, -1)
print('priority range (least, greatest):', lo, hi)
# HIP on this box has two levels (least 0, greatest -1); a priority outside the range hangs stream creation: never pass one
for chunks, prios in [(1, [0]), (2, [0, 0]), (2, [-1, 0]), (3, [-1, 0, 0]), (4, [-1, 0, 0, 0])]:
    assert all(hi <= p <= lo for p in prios)
    streams = [torch.cuda.Stream(device=dev, priority=p) for p in prios]
    streams = [streams[0]] + [streams[1]] * (chunks - 1) if chunks > 1 else streams      # chunk 0 on the high-priority stream, the rest in order on the other
    cuts = [nb * k // chunks for k in range(chunks + 1)]
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(chunks)] for _ in range(6)]
    start = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    main = torch.cuda.current_stream()
    def f(it):
        start[it].record(main)
        for c, (a, b) in enumerate(zip(cuts[:-1], cuts[1:])):
            part = {k: (v[a * band:b * band] if torch.is_tensor(v) else v) for k, v in sub.items()}
            st = streams[c]
            st.wait_event(start[it])
            lp.shade_bands(part, frame, dl, (band, n, r, a), pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out[a * band:b * band], stream=st)
            evs[it][c].record(st)
        for c in range(chunks):
            main.wait_event(evs[it][c])
    for it in range(6):
        f(it)
    torch.cuda.synchronize()
    t = np.array([[start[it].elapsed_time(evs[it][c]) * 1e3 for c in range(chunks)] for it in range(2, 6)])
    print(f'{chunks} chunk(s), priorities {prios}: chunk completion times (us, mean of 4) {np.round(t.mean(0), 1)}')
