// pt_bvh_build.h — host entry points of the GPU LBVH build (pt_bvh_build.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pt {

struct BvhBuffers {
    uint4 *nodes = nullptr;        // (n-1) x 2 uint4: 32-byte nodes, child boxes on a 16-bit grid (pt_bvh.h)
    float grid_lo[3] = {0, 0, 0};  // world = grid_lo + u * grid_cell, u = node units of the packed child boxes (pt_bvh.h)
    float grid_cell[3] = {0, 0, 0};
    uint4 *top = nullptr;          // the first kBvhTopLevels levels again, heap order, for the traversal kernels' LDS copy (pt_bvh.h)
    float4 *sph_sorted = nullptr;  // n x {c, r*r}, Morton order
    float4 *aux_sorted = nullptr;  // n x {1/r, material, scene order, r}
    uint32_t n = 0;
    // 4-wide collapse of the same tree for the queue pipeline's traversal kernels (pt_bvh.h: "Wide nodes")
    uint4 *wide = nullptr;         // n_wide x 4 uint4 (64-byte nodes), breadth-first order
    uint32_t n_wide = 0;
    uint32_t wide_levels = 0;      // levels of the wide tree (bounds the traversal stack: 3 entries per level)
};

// Builds the LBVH over n >= 2 spheres given as device arrays {c, r*r} and {1/r, material, order, r}.
// Synchronises `st`. depth_out = height of the tree (bounds the traversal stack).
hipError_t bvh_build(const float4 *d_sph, const float4 *d_aux, uint32_t n, BvhBuffers &out, hipStream_t st, int *depth_out);
void bvh_free(BvhBuffers &b);

}  // namespace pt
