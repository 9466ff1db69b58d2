"""Ad-hoc timing of kernel variants on the GPU box (not part of the product or the tests)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from renderer_amd import lighting, synth, ubo

def timeit(fn, n=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us

W, H = (3840, 2160) if len(sys.argv) < 3 else (int(sys.argv[1]), int(sys.argv[2]))
NPT = int(os.environ.get('NPT', '256'))
dev = 'cuda:0'
lp = lighting.LightingPass(0, 4096)
cam = synth.sweep_camera(W, H)
gbs = [synth.scene_sweep(W, H, cam, device=dev)]
for _ in range(2): gbs.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in gbs[0].items()})
frame = cam.frame_matrix_data(); dl = ubo.default_directional_lights(); pl = synth.point_light_field(NPT, 0x5EED0003)
sh = synth.shadow_setup(dl[0, :3], 2048, device=dev); amb = (0.03,) * 3
hdrs = [torch.empty((H, W, 4), device=dev) for _ in range(3)]
outs = [torch.empty((H, W), dtype=torch.int32, device=dev) for _ in range(3)]
it = [0]
def run(dlx, plx, ambx, shx, fmt=0, flags=0):
    def f():
        i = it[0] % 3; it[0] += 1
        lp.light_ex(gbs[i], frame, dlx, plx, ambx, shx, fmt=fmt, flags=flags, out=(hdrs[i] if fmt == 0 else outs[i]))
    return f
cnt = lp.tile_counts(gbs[0]); run(dl, pl, amb, sh)(); torch.cuda.synchronize(); lp.tile_counts(gbs[0], enable=False)
c = cnt.float(); print(f'tile list length: mean {c.mean():.1f} max {c.max():.0f} p50 {c.median():.0f}; pairs/px {lp.count_pairs(gbs[0], frame, pl)/(W*H):.2f}')
cases = [
 ('headline tiled (2dir+shadow+%dpt+amb)' % NPT, run(dl, pl, amb, sh)),
 ('  shadow via LDS', run(dl, pl, amb, sh, flags=4)),
 ('  no shadow', run(dl, pl, amb, None)),
 ('  no point lights (tiled, shadow LDS)', run(dl, None, amb, sh, flags=4)),
 ('  no point, shadow direct (stream)', run(dl, None, amb, sh)),
 ('  point only, no dir', run(dl[:0], pl, None, None)),
 ('  NO_CULL brute force', run(dl, pl, amb, sh, flags=2)),
 ('  fused -> sRGB8', run(dl, pl, amb, sh, fmt=2)),
 ('k1 1 light stream HDR', run(dl[:1], None, None, None)),
 ('k1 2 lights stream HDR', run(dl, None, None, None)),
 ('k1 1 light fused sRGB8', run(dl[:1], None, None, None, fmt=2)),
 ('k1 1 light fused LDR', run(dl[:1], None, None, None, fmt=1)),
 ('strict headline', run(dl, pl, amb, sh, flags=1)),
 ('strict k1 1 light', run(dl[:1], None, None, None, flags=1)),
]
if os.environ.get('K1ONLY'): cases = [c for c in cases if c[0].startswith('k1')]
if os.environ.get('HLONLY'): cases = cases[:4]
for name, f in cases:
    print(f'{name:50s} {timeit(f):9.1f} us')
def tm(fmt):
    def f():
        i = it[0] % 3; it[0] += 1
        lp.tonemap(hdrs[i], fmt, out=outs[i])
    return f
if not os.environ.get('K1ONLY') and not os.environ.get('HLONLY'):
  for fmt in (1, 2, 3): print(f'tonemap fmt {fmt:38d} {timeit(tm(fmt)):9.1f} us')
ldrs = [lp.tonemap(hdrs[i], 1) for i in range(3)]
def fx(strict=False):
    def f():
        i = it[0] % 3; it[0] += 1
        lp.fxaa(ldrs[i], out=outs[i], strict=strict)
    return f
if not os.environ.get('K1ONLY') and not os.environ.get('HLONLY'):
    print(f'{"fxaa R11G11B10F -> sRGB8":50s} {timeit(fx()):9.1f} us')
    print(f'{"fxaa strict":50s} {timeit(fx(True)):9.1f} us')
if not os.environ.get('K1ONLY') and not os.environ.get('HLONLY'):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
    from test_oracle_kat import sky_test_params
    prm = sky_test_params()
    acc = torch.zeros((H, W, 4), dtype=torch.float16, device=dev); rev = torch.ones((H, W), dtype=torch.float16, device=dev)
    acc[H // 3: 2 * H // 3, W // 4: 3 * W // 4] = torch.tensor([0.5, 0.4, 0.3, 0.8], dtype=torch.float16, device=dev); rev[H // 3: 2 * H // 3, W // 4: 3 * W // 4] = 0.4
    def sk():
        i = it[0] % 3; it[0] += 1
        lp.sky_hosek(gbs[i], frame, prm, hdrs[i])
    def rs():
        i = it[0] % 3; it[0] += 1
        lp.transparency_resolve(acc, rev, hdrs[i])
    def chain():
        i = it[0] % 3; it[0] += 1
        lp.light_ex(gbs[i], frame, dl, pl, amb, sh, out=hdrs[i]); lp.sky_hosek(gbs[i], frame, prm, hdrs[i])
        lp.transparency_resolve(acc, rev, hdrs[i]); lp.tonemap(hdrs[i], 1, out=ldrs[i]); lp.fxaa(ldrs[i], out=outs[i])
    print(f'{"sky (49 % of the texels)":50s} {timeit(sk):9.1f} us')
    print(f'{"transparency resolve (1/6 of the frame covered)":50s} {timeit(rs):9.1f} us')
    print(f'{"whole frame: light+sky+resolve+tonemap+fxaa":50s} {timeit(chain):9.1f} us')
    def one(sky=True, tr=True, fmt=1):
        def f():
            i = it[0] % 3; it[0] += 1
            lp.shade_frame(gbs[i], frame, dl, pl, amb, sh, sky=prm if sky else None, accum=acc if tr else None, reveal=rev if tr else None,
                           fmt=fmt, out=(ldrs[i] if fmt else hdrs[i]))
        return f
    def fused():
        i = it[0] % 3; it[0] += 1
        lp.shade_frame(gbs[i], frame, dl, pl, amb, sh, sky=prm, accum=acc, reveal=rev, fmt=1, out=ldrs[i]); lp.fxaa(ldrs[i], out=outs[i])
    print(f'{"shade_frame light+sky+resolve -> ldr (1 launch)":50s} {timeit(one()):9.1f} us')
    print(f'{"shade_frame light+sky -> ldr":50s} {timeit(one(tr=False)):9.1f} us')
    print(f'{"shade_frame light+resolve -> ldr":50s} {timeit(one(sky=False)):9.1f} us')
    print(f'{"shade_frame light -> ldr":50s} {timeit(one(False, False)):9.1f} us')
    print(f'{"shade_frame light+sky+resolve -> hdr":50s} {timeit(one(fmt=0)):9.1f} us')
    print(f'{"whole frame: shade_frame + fxaa (2 launches)":50s} {timeit(fused):9.1f} us')
# memcpy reference: copy 16 B/px in, 16 B/px out
if os.environ.get('K1ONLY') or os.environ.get('HLONLY'): sys.exit(0)
a = torch.empty(W * H * 4, device=dev); b = torch.empty_like(a)
print(f'{"torch copy 133MB->133MB":50s} {timeit(lambda: b.copy_(a)):9.1f} us')
