"""Sample-range partitioning across GPUs (north star "Multi-GPU partitioning").

Every rank renders all pixels for a contiguous range of GLOBAL sample indices; the RNG
stream is keyed (seed, pixel, global sample), so the reduced image does not depend on the
number of ranks. Uneven splits are handled (the reference's wgpu loop over-renders its last
dispatch, gpu_renderer.rs:98-102 — not reproduced)."""


def sample_range(spp, rank, world):
    """(begin, count) of rank's share of [0, spp); counts differ by at most one."""
    base, rem = divmod(int(spp), int(world))
    begin = rank * base + min(rank, rem)
    return begin, base + (1 if rank < rem else 0)
