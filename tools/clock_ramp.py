"""Step time against time since the GPU was idle: the dual step in bursts of 20 for 1.5 s after IDLE seconds (default 2) of
sleep.  Measured: 113 us/step in the second burst, 101 from ~20 ms on (the clock ramp bench.py orders its legs around)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from renderer_amd import lighting, synth, ubo
W, H = 3840, 2160
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
pl = synth.point_light_field(256, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(3)]
torch.cuda.synchronize()
time.sleep(float(os.environ.get('IDLE', '2')))
t_start = time.perf_counter()
res = []
i = 0
while time.perf_counter() - t_start < 1.5:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        k = i % 3; i += 1
        lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k], ldr_out=outs[k])
    e1.record(); torch.cuda.synchronize()
    res.append((time.perf_counter() - t_start, e0.elapsed_time(e1) / 20 * 1e3))
for j in list(range(0, 12)) + list(range(12, len(res), max(1, len(res) // 30))):
    print(f'{res[j][0]*1e3:8.1f} ms  {res[j][1]:7.1f} us/step')
