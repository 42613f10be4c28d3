// compact_pool.h -- the pooled epilogue of the SharedMLP kernels that run over COMPACT (centre, sample) rows (gemm_pool.hip,
// sa_mlp.hip): bias + ReLU, the maximum over each centre's rows, an integer atomic max into the centre's (zeroed) output row.
//
// The 32 x 32 accumulator of v_mfma_f32_32x32x2_f32 holds, in register v of a lane of half h (= lane / 32), row
// 8 (v / 4) + 4 h + v % 4 of the lane's column: a lane's 16 rows are four groups of four, so a run of rows of one centre (the
// compact rows of a centre are consecutive) is cut every four rows and costs an atomic per piece -- 4-5 per lane where the tile
// holds 2-3 centres.  Eight v_permlane32_swap_b32 (gfx950) re-deal the two halves' registers so that each half owns 16
// CONSECUTIVE rows (half 0: rows 0-15 of the tile, half 1: rows 16-31): one atomic per centre a half touches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ws3d {

// the centres of the 16 consecutive rows row0 .. row0 + 15 (-1: behind the end)
__device__ __forceinline__ void compact_centres16(const int32_t *__restrict__ rowc, long row0, long T, int (&cen)[16]) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cen[r] = row0 + r < T ? rowc[row0 + r] : -1;
}

// acc: the accumulator of one 32 x 32 tile (lane = column, registers = rows as above); cen: compact_centres16 of THIS half's rows
// (tile row 16 h + r); out_col = the output column of this lane; every value is ReLU'd (>= 0: the float order is the integer order)
template <typename ACC>
__device__ __forceinline__ void compact_pool_atomic(const ACC &acc, float bias, const int (&cen)[16], float *__restrict__ out_col, long out_stride) {
    float y[16];
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        // vdst = acc[v], src = acc[v + 8]: lanes 32-63 of vdst <-> lanes 0-31 of src.  Afterwards, in half h: r[0] = row 16 h + 8 (v / 4) + v % 4,
        // r[1] = row 16 h + 8 (v / 4) + 4 + v % 4
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[v]), __float_as_uint(acc[v + 8]), false, false);
        y[8 * (v / 4) + (v % 4)] = __uint_as_float(r[0]);
        y[8 * (v / 4) + 4 + (v % 4)] = __uint_as_float(r[1]);
    }
    int prev = -1;
    float run = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        if (cen[r] < 0) continue;
        const float v = fmaxf(y[r] + bias, 0.f);          // NaN -> 0 (fmaxf), as in SA1's dense / list kernels (sa_mlp.hip): the integer atomicMax below never sees a NaN pattern.  (The dense last-layer kernels of SA2-4, gemm_pool.hip, propagate a NaN like torch's relu + max_pool2d do; an atomic integer maximum cannot, so on NON-FINITE activations the two sides of the launch gate differ -- finite inputs and weights never produce one)
        if (cen[r] != prev) {
            if (prev >= 0) atomicMax(reinterpret_cast<int *>(out_col + (long)prev * out_stride), __float_as_int(run));
            prev = cen[r]; run = v;
        } else {
            run = fmaxf(run, v);
        }
    }
    if (prev >= 0) atomicMax(reinterpret_cast<int *>(out_col + (long)prev * out_stride), __float_as_int(run));
}

}  // namespace ws3d
