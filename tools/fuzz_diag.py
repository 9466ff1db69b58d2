import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['FUZZ_DIAG'] = '1'
import fuzz_parity as fz
from renderer_amd import lighting
fz.coracle.build()
lp = lighting.LightingPass(0, max_point_lights=512)
for s in sys.argv[1:]:
    print(fz.one(lp, int(s)))
