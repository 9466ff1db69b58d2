import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
print('torch', torch.__version__, torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device='cuda:0'); torch.cuda.synchronize()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'renderer_amd', 'libmil.so'))
for l in open('/proc/self/maps'):
    if ('amdhip' in l or 'hsa-runtime' in l or 'libmil' in l) and 'r-xp' in l: print(l.strip())
h = ctypes.CDLL('libamdhip64.so.7')
n = ctypes.c_int(-1); rc = h.hipGetDeviceCount(ctypes.byref(n)); print('hipGetDeviceCount rc', rc, 'n', n.value)
v = ctypes.c_int(0); h.hipRuntimeGetVersion(ctypes.byref(v)); print('runtime version', v.value)
c = ctypes.c_void_p(); rc = lib.mil_create(ctypes.byref(c), 0, 16); print('mil_create', rc)
print({k: v for k, v in os.environ.items() if 'VISIBLE' in k or 'HIP' in k or 'ROCR' in k or 'HSA' in k})
