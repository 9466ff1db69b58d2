import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, 'libbwprobe.so'))
W, H = 3840, 2160
dev = 'cuda:0'
sets = []
for _ in range(3):
    sets.append([torch.randint(0, 2**31 - 1, (H, W), dtype=torch.int32, device=dev) for _ in range(4)] + [torch.empty((H, W, 4), device=dev)])
names = {0: 'A 1px nt/nt', 1: 'A plain/plain', 2: 'A plain ld/nt st', 3: 'A nt ld/plain st', 4: 'B 4px x4 nt', 5: 'B 4px x4 plain',
         6: 'C 4px LDS-transposed nt', 7: 'C LDS-transposed plain', 8: 'D 2px nt', 9: 'E gridstride 2048 nt', 10: 'E gridstride 4096 nt',
         11: 'E gridstride 2048 plain', 12: 'R read 133MB only', 13: 'W write 133MB only', 20: 'V +0 fma', 21: 'V +100 fma', 22: 'V +200 fma', 23: 'V +300 fma', 24: 'V +400 fma', 25: 'V +600 fma', 26: 'V +800 fma'}
it = [0]
def call(v):
    s = sets[it[0] % 3]; it[0] += 1
    rc = lib.probe(v, *[ctypes.c_void_p(t.data_ptr()) for t in s], W, H, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc
sel = [int(a) for a in sys.argv[1:]] or list(names)
for v in sel:
    for _ in range(5): call(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): call(v)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    mb = 133 if v >= 12 else 265.4
    print(f'{names[v]:32s} {us:8.1f} us  {mb / us * 1e6 / 1e6:7.2f} TB/s')
