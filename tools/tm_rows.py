"""Ad-hoc: tonemap kernel timing for the rows-per-lane variants (env MIL_TONEMAP_ROWS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting
W, H = 3840, 2160
lp = lighting.LightingPass(0, 16)
hdrs = [torch.rand((H, W, 4), device='cuda:0') * 4 for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.int32, device='cuda:0') for _ in range(3)]
for fmt in (1, 2, 3):
    for i in range(5): lp.tonemap(hdrs[i % 3], fmt, out=outs[i % 3])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(30): lp.tonemap(hdrs[i % 3], fmt, out=outs[i % 3])
    e1.record(); torch.cuda.synchronize()
    print(f'rows {os.environ.get("MIL_TONEMAP_ROWS", "default")} fmt {fmt}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us')
