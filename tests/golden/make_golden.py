"""Generates the golden fixtures from the CPU oracle (never from reference code — the
reference is Rust and cannot be run here). Inputs are seeded numpy draws;
expected outputs are the oracle's. Run: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import oracle_lib  # noqa: E402
from helpers import example_rays, unit_vectors  # noqa: E402
from pathtracer_amd import Scene  # noqa: E402

N = 4096


def main():
    o = oracle_lib.Oracle()
    ex = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"))
    rng = np.random.default_rng(20250629)
    out = {}
    ro, rd = example_rays(rng, N)
    t, n, idx = o.unit.intersect(ex, ro, rd)
    out.update(isect_ro=ro, isect_rd=rd, isect_t=t, isect_n=n, isect_idx=idx)
    dist2 = rng.uniform(0.01, 60.0, N).astype(np.float32)
    out.update(occ_dist2=dist2, occ=o.unit.occluded(ex, ro, rd, dist2))

    nn, v, l = unit_vectors(rng, N), unit_vectors(rng, N), unit_vectors(rng, N)
    v[np.einsum("ij,ij->i", nn, v) < 0] *= -1
    mat = np.zeros((N, 8), np.float32)
    mat[:, 0:3] = rng.uniform(0.05, 0.95, (N, 3))
    mat[:, 3] = rng.choice([0.0, 1.0, 0.3], N)
    mat[:, 4] = np.exp(rng.uniform(np.log(0.05), 0.0, N))
    mat[:, 5] = 1.0
    d2 = rng.uniform(0.5, 30.0, N).astype(np.float32)
    light = {k: [float(x) for x in ex.lights[0][k]] for k in ("pos", "u", "v", "intensity")}
    out.update(nee_n=nn, nee_v=v, nee_l=l, nee_d2=d2, nee_mat=mat, nee_rgb=o.unit.nee_eval(nn, v, l, d2, mat, light))

    rough = np.exp(rng.uniform(np.log(0.01), 0.0, N)).astype(np.float32)
    r1, r2 = rng.random(N, np.float32), rng.random(N, np.float32)
    out.update(gs_n=nn, gs_rough=rough, gs_r1=r1, gs_r2=r2, gs_h=o.unit.ggx_sample(nn, rough, r1, r2))
    out.update(dif_dir=o.unit.diffuse_sample(nn, r1, r2))
    ior = rng.choice([1.33, 1.5, 3.0], N).astype(np.float32)
    out.update(gl_rd=l, gl_ior=ior, gl_u=r1, gl_dir=o.unit.glass(l, nn, ior, r1))
    g = rng.choice([0.0, 0.3, -0.4, 0.8], N).astype(np.float32)
    c, d = o.unit.hg(l, g, r1, r2)
    out.update(hg_g=g, hg_cos=c, hg_dir=d)
    xy = np.stack([rng.integers(0, ex.width, N), rng.integers(0, ex.height, N)], 1).astype(np.uint32)
    u4 = rng.random((N, 4), np.float32)
    cro, crd = o.unit.camera_ray(ex, xy, u4)
    out.update(cam_xy=xy, cam_u4=u4, cam_ro=cro, cam_rd=crd)
    rgb = np.exp(rng.uniform(np.log(1e-4), np.log(50.0), (N, 3))).astype(np.float32)
    out.update(tm_rgb=rgb, tm_u8=o.unit.tonemap(rgb))
    out.update(rng_seed0_pixel5_sample9=o.unit.rng(0, 5, 9, 64))
    np.savez_compressed(os.path.join(HERE, "unit_vectors.npz"), **out)

    # (v) sample-exact image target + per-pixel variance; (vi) image means + event counters
    s = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 64, 48, 256)
    parts = [o.render_accum(s, 32 * k, 32, seed=0) for k in range(8)]
    acc = np.sum(parts, axis=0, dtype=np.float32)
    var = np.var(np.stack(parts) / 32.0, axis=0) * 32.0       # per-sample variance estimate from 8 batch means
    np.savez_compressed(os.path.join(HERE, "example_64x48_256spp.npz"), accum=o.render_accum(s, 0, 256, seed=0),
                        accum_from_parts=acc, var=var.astype(np.float32))
    s2 = Scene.load(os.path.join(ROOT, "scenes", "scene.example.json"), 320, 240, 16)       # BASELINE config C1
    a2, c2 = o.render_accum(s2, seed=0, counters=True)
    s3 = Scene.load(os.path.join(ROOT, "scenes", "glass_fog.json"), 160, 120, 16)
    a3, c3 = o.render_accum(s3, seed=0, counters=True)
    meta = {
        "c1_example_320x240_16spp": {"mean_radiance": [float(x) for x in a2.mean(axis=(0, 1), dtype=np.float64) / 16],
                                     "counters": c2},
        "glass_fog_160x120_16spp": {"mean_radiance": [float(x) for x in a3.mean(axis=(0, 1), dtype=np.float64) / 16],
                                    "counters": c3},
    }
    with open(os.path.join(HERE, "image_stats.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden written")


if __name__ == "__main__":
    main()
