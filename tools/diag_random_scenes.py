import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib
from pathtracer_amd import hip_renderer
import test_gpu_random_scenes as T
o=oracle_lib.Oracle()
for case in T.CASES:
    s = T.random_scene(*case)
    for mult in (1, 4):
        s2 = s.with_render(s.width, s.height, s.samples*mult)
        a = hip_renderer.render_accum(s2, seed=case[0]); b = o.render_accum(s2, seed=case[0])
        ta, tb = a.sum(dtype=np.float64), b.sum(dtype=np.float64)
        err = np.abs(a.astype(np.float64)-b).max(axis=2)/np.maximum(np.abs(b).max(axis=2),1e-3*s2.samples)
        print(case[:4], s2.samples, 'total rel', f'{abs(ta-tb)/abs(tb):.2e}', 'q50 %.1e q80 %.1e q95 %.1e' % tuple(np.quantile(err,[.5,.8,.95])))
