"""Runs ONE kernel configuration N times (for rocprofv3 passes).  usage: run_case.py <case> [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
case = sys.argv[1]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
if os.environ.get('TWO'): lp.set_two_kernel_threshold(int(os.environ['TWO']))
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(4096 if case == 'l4096' else int(os.environ.get('NPT', '256')), 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
far = pl.copy(); far[:, 0:3] += 1.0e4; far1 = far[:1]      # point lights far outside the scene (budget cases)
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(3)]
frs = None
if case == 'accumulate':
    fr = synth.transparent_fragments(W * H, seed=7)
    frs = [{k: torch.from_numpy(v).to(dev) for k, v in fr.items()} for _ in range(2)]
    acc = [torch.empty((W * H, 4), device=dev) for _ in range(2)]; rev = [torch.empty((W * H,), device=dev) for _ in range(2)]
ldrs = [lp.tonemap(lp.light_ex(gbs[0], frame, dl, pl, amb, sh), 1) for _ in range(3)] if case == 'fxaa' else None
torch.cuda.synchronize()
for i in range(iters):
    k = i % 3
    if case == 'k1': lp.light_directional(gbs[k], frame, dl[:1], out=hdrs[k])
    elif case == 'k1_2': lp.light_directional(gbs[k], frame, dl, out=hdrs[k])
    elif case in ('headline', 'l4096'): lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k])      # l4096: run with NPT=4096
    elif case == 'dual': lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k], ldr_out=outs[k])
    elif case == 'headline_lds': lp.light_ex(gbs[k], frame, dl, pl, amb, sh, flags=4, out=hdrs[k])
    elif case == 'hl_noshadow': lp.light_ex(gbs[k], frame, dl, pl, amb, None, out=hdrs[k])
    elif case == 'hl_nopoints': lp.light_ex(gbs[k], frame, dl, None, amb, sh, flags=4, out=hdrs[k])
    elif case == 'hl_nodir': lp.light_ex(gbs[k], frame, dl[:0], pl, amb, None, out=hdrs[k])
    elif case == 'points': lp.light_ex(gbs[k], frame, dl[:0], pl, None, None, out=hdrs[k])
    elif case == 'b_base': lp.light_ex(gbs[k], frame, dl[:0], far1, None, None, out=hdrs[k])       # k_tiled with nothing to add: loads, box, two setups, stores
    elif case == 'b_cull': lp.light_ex(gbs[k], frame, dl[:0], far, None, None, out=hdrs[k])        # + culling NPT lights that reach nothing
    elif case == 'tonemap': lp.tonemap(hdrs[k], 2, out=outs[k])
    elif case == 'fxaa': lp.fxaa(ldrs[k], out=outs[k])
    elif case == 'accumulate': lp.transparency_accumulate(frs[i % 2], frame, dl, accum=acc[i % 2], reveal=rev[i % 2])
    else: raise SystemExit('unknown case')
torch.cuda.synchronize()
