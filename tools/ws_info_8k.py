"""8K / 4096 lights through the two-kernel path: workspace use (mil_workspace_info) and, under rocprofv3, per-kernel times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo
W, H = (7680, 4320) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
if os.environ.get('MIL_LIB'):      # an older build of the library for A/B timing: symbols it lacks are stubbed
    import ctypes
    class Shim(ctypes.CDLL):
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                if name.startswith('mil_'):
                    return type('Missing', (), {})()
                raise
    lighting.ctypes.CDLL = Shim
    lighting.load_library(os.environ['MIL_LIB'])
lp = lighting.LightingPass(0, 4096)
if os.environ.get('POOL'): lp.set_list_pool(int(os.environ['POOL']))
cam = synth.sweep_camera(W, H)
gb = synth.scene_sweep(W, H, cam, device='cuda:0')
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(4096, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device='cuda:0'); amb = (0.03,) * 3
out = torch.empty((H, W), dtype=torch.int32, device='cuda:0')
for i in range(8):
    lp.light_ex(gb, frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out)
torch.cuda.synchronize()
print(lp.workspace_info() if not os.environ.get('MIL_LIB') else 'old lib')
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10):
    lp.light_ex(gb, frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, out=out)
e1.record(); torch.cuda.synchronize()
print('ms per frame', e0.elapsed_time(e1) / 10)
