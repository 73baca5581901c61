"""Algorithmic FLOP model of the trace loop.

Convention: add/sub/mul/div/sqrt/transcendental = 1, FMA = 2; abs/neg/min/max/
compare/select/int/RNG = 0; per-vertex and per-light invariants hoisted. Event
counts come from the device counters (HIPPT_FLAG_COUNTERS); extend tests are
credited in full (rays x objects), shadow tests as executed.
"""

F_SPH = 29      # sphere test (sphere.rs:26-33)
F_PL = 33       # plane test, u.u / v.v hoisted (plane.rs:25-52)
F_ANY = 2       # t*t, dist2*0.999 per shadow test (renderer.rs:46)
F_FIN = 13      # hit point + normal per extend hit
F_CAM = 70      # camera ray (renderer.rs:170-178, algebra.rs:67-73)
F_VTX = 19      # n.v, f0, diffuse colour per NEE vertex
F_SS = 38       # light point, l, dist2, origin per shadow sample
F_EVAL = 103    # BRDF evaluation per unoccluded front-facing shadow sample
F_LSUM = 6      # 1/4 average + add, per light per vertex
F_DIFF = 76     # diffuse lobe sample + weight + origin offset
F_SPEC = 81     # GGX lobe sample + reflect + weight + origin offset
F_GLASS = 50    # glass vertex
F_MED = 14      # free flight + absorption per in-medium vertex
F_SCAT = 74     # scatter direction
F_ACC = 3       # accumulate per path
F_RES = 33      # resolve per pixel


def algorithmic_flops(c, n_lights, n_pixels=0):
    """c: counter dict (pathtracer_amd._abi.COUNTER_NAMES)."""
    return (
        c["extend_sphere_tests"] * F_SPH + c["extend_plane_tests"] * F_PL + c["extend_hits"] * F_FIN
        + c["paths"] * (F_CAM + F_ACC)
        + c["nee_vertices"] * (F_VTX + F_LSUM * n_lights)
        + c["shadow_rays"] * F_SS
        + c["shadow_sphere_tests"] * (F_SPH + F_ANY) + c["shadow_plane_tests"] * (F_PL + F_ANY)
        + c["lit_evals"] * F_EVAL
        + c["diffuse_lobes"] * F_DIFF + c["spec_lobes"] * F_SPEC + c["glass_vertices"] * F_GLASS
        + c["medium_vertices"] * F_MED + c["scatter_events"] * F_SCAT
        + n_pixels * F_RES
    )


def rays(c):
    return c["extend_rays"] + c["shadow_rays"]
