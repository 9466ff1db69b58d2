"""Throughput of the 4K / NPT-light lighting launch with frames alternating over TWO contexts on two streams (k_bin of one frame
can run under the tail of k_shade of the other) against one context on one stream.  usage: two_ctx_overlap.py [NPT]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
NPT = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
W, H = 3840, 2160
dev = 'cuda:0'
lps = [lighting.LightingPass(0, 4096), lighting.LightingPass(0, 4096)]
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
outs = [torch.empty((H, W, 4), device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
def run(n, two):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for s in streams: s.wait_event(e0)
    for i in range(n):
        k = i % 2 if two else 0
        lps[k].light_ex(gbs[k], frame, dl, pl, amb, sh, out=outs[k], stream=streams[k])
    cur = torch.cuda.current_stream()
    for s in streams: cur.wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for two in (False, True):
    run(6, two)
    print(f'NPT={NPT}: {"two contexts on two streams" if two else "one context, one stream"}: {min(run(20, two) for _ in range(3)):.1f} us per frame')
