"""Ad-hoc: host time of one call through the C ABI (tiny frame, so the GPU never limits), split into the Python binding
and the library itself (the raw ctypes call with pre-marshalled arguments)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, synth, ubo
W, H = 64, 64
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H); gb = synth.scene_sweep(W, H, cam, device='cuda:0')
frame = np.ascontiguousarray(cam.frame_matrix_data(), np.float32).reshape(68); dl = ubo.default_directional_lights()
hdr = torch.empty((H, W, 4), device='cuda:0')
def t(fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
print(f'mil_version() through ctypes      {t(lambda: lp.lib.mil_version()):6.2f} us')
print(f'light_directional (python binding) {t(lambda: lp.light_directional(gb, frame, dl, out=hdr)):6.2f} us')
g = lp._gb(gb); push = ubo.push_constants(W, H, len(dl)); lights = np.ascontiguousarray(dl, np.float32)
a = (lp.ctx, ctypes.byref(g), ctypes.c_void_p(frame.ctypes.data), ctypes.c_void_p(lights.ctypes.data), ctypes.c_void_p(push.ctypes.data), ctypes.c_void_p(hdr.data_ptr()), W, 0, None)
print(f'mil_light_directional (raw call)   {t(lambda: lp.lib.mil_light_directional(*a)):6.2f} us')
for n in (256, 4096):
    pl = np.ascontiguousarray(synth.point_light_field(n, 3), np.float32)
    ex = lighting.Extras(); ex.point_lights, ex.n_point = pl.ctypes.data, n
    b = (lp.ctx, ctypes.byref(g), ctypes.c_void_p(frame.ctypes.data), ctypes.c_void_p(lights.ctypes.data), ctypes.c_void_p(push.ctypes.data), ctypes.byref(ex), ctypes.c_void_p(hdr.data_ptr()), W, 0, 0, None)
    print(f'mil_light_ex, {n:4d} static point lights (raw call) {t(lambda: lp.lib.mil_light_ex(*b), 500):6.2f} us')
out = torch.empty((H, W), dtype=torch.int32, device='cuda:0')
c = (lp.ctx, ctypes.c_void_p(hdr.data_ptr()), W, W, H, ctypes.c_void_p(out.data_ptr()), W, 2, 0, None)
print(f'mil_tonemap (raw call)             {t(lambda: lp.lib.mil_tonemap(*c)):6.2f} us')
