// binning.h -- layout of the per-scene x-binned point copy built by bin_points_x_kernel
// (ballquery_group.hip) and read by the binned ball query and the binned three_nn.
//   [n x float4 {x, y, z, original index}] [BinHeader] [BQS_CELLS + 1 cell start offsets]
// Points are counting-sorted by cell; the order inside a cell is unspecified.
// A second flavour of the same buffer, built by bin_points_xz_kernel (interpolate.hip) for the 3-NN
// search only: cells of a gx x gz grid over (x, z), cell id = cz * gx + cx, gx * gz <= BQS_CELLS.
// It is marked by BinHeader.pad = gx > 0; zmin, 1/cell depth and gz live in the three spare ints
// behind the BQS_CELLS + 1 offsets (the table is padded to 16 bytes).
#pragma once
#include "common.h"

namespace ws3d {

constexpr int SORT_MAX_N = 16384;
constexpr int BQS_CELLS = 2048;

struct BinHeader { float xmin, inv_w; int n, pad; };   // 16 bytes, follows the float4 array

__host__ __device__ inline size_t bin_scene_stride(int n) {
    return (size_t)n * 16 + sizeof(BinHeader) + (((size_t)(BQS_CELLS + 1) * 4 + 15) / 16) * 16;
}

constexpr int GRID_ZMIN = BQS_CELLS + 1, GRID_INV_WZ = BQS_CELLS + 2, GRID_GZ = BQS_CELLS + 3;   // spare table slots

// grid coordinate along one axis: monotone non-decreasing for finite v; NaN -> 0
__device__ __forceinline__ int grid_coord(float v, float vmin, float inv_w, int cells) {
    const float t = (v - vmin) * inv_w;
    return t > 0.f ? (t < (float)(cells - 1) ? (int)t : cells - 1) : 0;
}

// monotone non-decreasing in x for finite x; NaN -> cell 0 (a NaN point can never be a hit)
__device__ __forceinline__ int x_cell(float x, float xmin, float inv_w) {
    const float t = (x - xmin) * inv_w;
    int c = t > 0.f ? (t < (float)(BQS_CELLS - 1) ? (int)t : BQS_CELLS - 1) : 0;
    return c;
}

}  // namespace ws3d
