"""Same-box A/B timing of two builds of libmil.so (boxes of the pool differ by 10 % and more): both libraries are loaded into
one process, the cases alternate A, B, A, B ... in bursts between HIP events, medians are printed.
usage: python tools/ab_time.py <libA.so> <libB.so> [case ...]   cases: dual hdr fused tonemap fxaa l4096 k1 frame"""
import ctypes, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from renderer_amd import lighting, synth, ubo


class Shim(ctypes.CDLL):          # an older build lacks newer symbols: stub them
    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name.startswith('mil_'):
                return type('Missing', (), {})()
            raise


lighting.ctypes.CDLL = Shim
paths = sys.argv[1:3]
cases = sys.argv[3:] or ['dual', 'hdr', 'fused', 'tonemap', 'fxaa', 'l4096']
lps = []
for pth in paths:
    lighting._lib = None
    lighting.load_library(os.path.abspath(pth))
    lps.append(lighting.LightingPass(0, 4096))
W, H = 3840, 2160
dev = 'cuda:0'
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2):
    gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights()
pl = synth.point_light_field(256, 0x5EED0003); pl4 = synth.point_light_field(4096, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(3)]
ldrs = [lps[0].tonemap(lps[0].light_ex(gbs[0], frame, dl, pl, amb, sh), 1) for _ in range(3)]
prm = synth.hosek_params_synthetic()
ta, tr = synth.transparency_layer(W, H, dev)


def run(lp, case, i):
    k = i % 3
    if case == 'dual': lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k], ldr_out=outs[k])
    elif case == 'hdr': lp.light_ex(gbs[k], frame, dl, pl, amb, sh, out=hdrs[k])
    elif case == 'fused': lp.light_ex(gbs[k], frame, dl, pl, amb, sh, fmt=lighting.OUT_SRGB8, out=outs[k])
    elif case == 'tonemap': lp.tonemap(hdrs[k], 2, out=outs[k])
    elif case == 'fxaa': lp.fxaa(ldrs[k], out=outs[k])
    elif case == 'l4096': lp.light_ex(gbs[k], frame, dl, pl4, amb, sh, out=hdrs[k])
    elif case == 'k1': lp.light_directional(gbs[k], frame, dl[:1], out=hdrs[k])
    elif case == 'frame': lp.shade_frame(gbs[k], frame, dl, pl, amb, sh, sky=prm, accum=ta, reveal=tr, fmt=lighting.OUT_LDR, out=outs[k])
    else: raise SystemExit('unknown case ' + case)


for case in cases:
    res = [[], []]
    for rnd in range(5):
        for a in (0, 1):
            lp = lps[a]
            for i in range(3): run(lp, case, i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for i in range(n): run(lp, case, i)
            e1.record(); torch.cuda.synchronize()
            res[a].append(e0.elapsed_time(e1) / n * 1e3)
    ma, mb = statistics.median(res[0]), statistics.median(res[1])
    print(f'{case:8s} A {ma:8.1f} us  B {mb:8.1f} us  B/A {mb / ma:.3f}   (A {min(res[0]):.1f}-{max(res[0]):.1f}, B {min(res[1]):.1f}-{max(res[1]):.1f})', flush=True)
