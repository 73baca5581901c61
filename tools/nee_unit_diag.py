import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import oracle_lib
from helpers import unit_vectors
from pathtracer_amd import hip_renderer
hip=hip_renderer.unit_ops(0); oracle=oracle_lib.Oracle()
N=1<<16
rng = np.random.default_rng(3)
n = unit_vectors(rng, N); v = unit_vectors(rng, N)
v[np.einsum("ij,ij->i", n, v) < 0] *= -1
l = unit_vectors(rng, N)
dist2 = rng.uniform(0.5, 30.0, N).astype(np.float32)
mat = np.zeros((N, 8), np.float32)
mat[:, 0:3] = rng.uniform(0.05, 0.95, (N, 3)); mat[:, 3] = rng.choice([0.0, 1.0, 0.3], N)
mat[:, 4] = np.exp(rng.uniform(np.log(0.01), 0.0, N)); mat[:, 5] = 1.0
light = dict(pos=(0, 3, 4), u=(-1, 0, 0), v=(0, 0.3, -1), intensity=(25, 20, 15))
a = hip.nee_eval(n, v, l, dist2, mat, light); b = oracle.unit.nee_eval(n, v, l, dist2, mat, light)
err = np.linalg.norm(a.astype(np.float64) - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-3)
h = v.astype(np.float64) + l; hlen=np.linalg.norm(h,axis=1); h /= hlen[:,None]
nh = np.maximum(np.einsum("ij,ij->i", n.astype(np.float64), h), 0.0)
q = nh ** 2 * (mat[:, 4].astype(np.float64) ** 2 - 1) + 1
nv = np.einsum("ij,ij->i", n.astype(np.float64), v); nl = np.einsum("ij,ij->i", n.astype(np.float64), l)
print("max err", err.max(), "q99", np.quantile(err,.99), "q999", np.quantile(err,.999))
lit = b.sum(1)>0
for reg,name in (((hlen>0.2)&(nv>0.01)&(nl>0.01),"reg A"),((hlen>0.5)&(nv>0.05)&(nl>0.05),"reg B"),(np.ones(N,bool),"all")):
    r=reg&lit
    print(name, "max err*q", (err[r]*q[r]).max(), "max err*q*hlen*min(nv,nl)", (err[r]*q[r]*hlen[r]*np.minimum(nv[r],nl[r])).max(), "max err", err[r].max())
i=np.argmax(err); print("worst", err[i], q[i], hlen[i], nv[i], nl[i], mat[i,4], mat[i,3])
