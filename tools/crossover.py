"""k_tiled against k_bin -> k_order -> k_shade by light count (4K, hdr out, same box, alternating): where MIL_TWO_KERNEL_DEFAULT belongs."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
dev = 'cuda:0'
one, two = lighting.LightingPass(0, 4096), lighting.LightingPass(0, 4096)
one.set_two_kernel_threshold(0xFFFFFFFF); two.set_two_kernel_threshold(1)
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
for n in [int(a) for a in sys.argv[1:]] or [256, 512, 768, 1024, 1536, 2048, 4096]:
    pl = synth.point_light_field(n, 0x5EED0003)
    res = [[], []]
    for rnd in range(4):
        for a, lp in enumerate((one, two)):
            for i in range(3): lp.light_ex(gbs[i % 3], frame, dl, pl, amb, sh, out=hdrs[i % 3])
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(15): lp.light_ex(gbs[i % 3], frame, dl, pl, amb, sh, out=hdrs[i % 3])
            e1.record(); torch.cuda.synchronize()
            res[a].append(e0.elapsed_time(e1) / 15 * 1e3)
    print(f'{n:5d} lights: k_tiled {statistics.median(res[0]):7.1f} us   two-kernel {statistics.median(res[1]):7.1f} us', flush=True)
