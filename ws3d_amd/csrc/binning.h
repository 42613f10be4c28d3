// binning.h -- layout of the per-scene x-binned point copy built by bin_points_x_kernel
// (ballquery_group.hip) and read by the binned ball query and the binned three_nn.
//   [n x float4 {x, y, z, original index}] [BinHeader] [BQS_CELLS + 1 cell start offsets]
// Points are counting-sorted by cell; the order inside a cell is unspecified.
// A second flavour of the same buffer, built by bin_points_xz_kernel (interpolate.hip) for the 3-NN
// search only: cells of a gx x gz grid over (x, z), cell id = cz * gx + cx, gx * gz <= BQS_CELLS.
// It is marked by BinHeader.pad = gx > 0; zmin, 1/cell depth and gz live in the three spare ints
// behind the BQS_CELLS + 1 offsets (the table is padded to 16 bytes).
// Third flavour, built by bin_points_grid_kernel (ballquery_group.hip) for the BALL QUERY: a FINE (x, z) grid of up to
// GRID16_CELLS near-square cells (~0.5 points per cell: 0.4 m cells on a KITTI scene) with 16-bit cell-start offsets
// (n <= 16384 < 65536), marked by BinHeader.pad = -gx < 0:  [GRID16_CELLS + 1 uint16 offsets][zmin, 1/cell depth, gz].
// A centre then scans, per grid row overlapping |z - cz| < r, ONE contiguous range of cells overlapping |x - cx| < r.
#pragma once
#include "common.h"

namespace ws3d {

constexpr int SORT_MAX_N = 16384;
constexpr int BQS_CELLS = 2048;
constexpr int GRID16_CELLS = 32768;
constexpr size_t GRID16_TABLE_BYTES = (((size_t)(GRID16_CELLS + 1) * 2 + 3 * 4 + 15) / 16) * 16;   // offsets + zmin, inv_wz, gz

struct BinHeader { float xmin, inv_w; int n, pad; };   // 16 bytes, follows the float4 array

__host__ __device__ inline size_t bin_scene_stride(int n) {
    // the table region holds whichever flavour was built: it is sized for the largest (the 16-bit fine grid)
    return (size_t)n * 16 + sizeof(BinHeader) + GRID16_TABLE_BYTES;
}

constexpr int GRID_ZMIN = BQS_CELLS + 1, GRID_INV_WZ = BQS_CELLS + 2, GRID_GZ = BQS_CELLS + 3;   // spare table slots

// grid coordinate along one axis: monotone non-decreasing for finite v; NaN -> 0
__device__ __forceinline__ int grid_coord(float v, float vmin, float inv_w, int cells) {
    const float t = (v - vmin) * inv_w;
    return t > 0.f ? (t < (float)(cells - 1) ? (int)t : cells - 1) : 0;
}

// byte offset of the three grid parameters behind the 16-bit offsets of the fine-grid flavour
constexpr size_t GRID16_PARAMS = (((size_t)(GRID16_CELLS + 1) * 2 + 3) / 4) * 4;

// monotone non-decreasing in x for finite x; NaN -> cell 0 (a NaN point can never be a hit)
__device__ __forceinline__ int x_cell(float x, float xmin, float inv_w) {
    const float t = (x - xmin) * inv_w;
    int c = t > 0.f ? (t < (float)(BQS_CELLS - 1) ? (int)t : BQS_CELLS - 1) : 0;
    return c;
}

}  // namespace ws3d
