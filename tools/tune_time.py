"""Times one case under several MIL_TUNE values in ONE process (contexts read MIL_TUNE at mil_create).  usage: tune_time.py <case> <tune> [<tune> ...]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['MIL_DEBUG'] = '1'
import torch
from renderer_amd import lighting, synth, ubo
case = sys.argv[1]; tunes = [int(t, 0) for t in sys.argv[2:]]
lps = []
for t in tunes:
    os.environ['MIL_TUNE'] = str(t)
    lps.append(lighting.LightingPass(0, 4096))
W, H = 3840, 2160
dev = 'cuda:0'
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
pl = synth.point_light_field(4096 if case == 'l4096' else 256, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(3)]
def run(lp, i):
    k = i % 3
    if case == 'dual': lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k], ldr_out=outs[k])
    else: lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k])
res = [[] for _ in tunes]
for rnd in range(5):
    for a, lp in enumerate(lps):
        for i in range(3): run(lp, i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(20): run(lp, i)
        e1.record(); torch.cuda.synchronize()
        res[a].append(e0.elapsed_time(e1) / 20 * 1e3)
for t, r in zip(tunes, res):
    print(f'{case} tune {t:#x}: median {statistics.median(r):.1f} us ({min(r):.1f}-{max(r):.1f})', flush=True)
