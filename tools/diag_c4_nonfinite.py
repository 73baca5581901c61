"""Diagnostic: non-finite pixels of the C4 step (1920x1080, 256 spp) and what the oracle returns for their samples."""
import sys, numpy as np
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import oracle_lib
from pathtracer_amd import hip_renderer, _abi
from pathtracer_amd.synthetic import synthetic_spheres
o = oracle_lib.Oracle()
s = synthetic_spheres(100000, 0x5EED, 1920, 1080, 256)
x, y = 662, 292
xys = np.stack([np.full(256, x), np.full(256, y), np.arange(256)], 1)
for depth in (1, 2, 3, 4, 5):
    ref = o.paths(s, xys, seed=0, iterative=True, max_depth=depth)
    k = int(np.argmax(ref.max(axis=1)))
    print('depth', depth, 'oracle max sample', k, ref[k])
    for fl, nm in ((0, 'queue'), (_abi.FLAG_MEGAKERNEL, 'mega'), (_abi.FLAG_NO_BVH, 'brute')):
        one = hip_renderer.render_accum(s, k, 1, seed=0, max_depth=depth, flags=fl)
        print('   hip', nm, one[y, x])
