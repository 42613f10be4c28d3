// EXPERIMENT, not part of the library (round 4, profiles/r04_gemm_rows_vs_library.txt): the state that was measured as ws3d_gemm_rows.
// To rebuild it: add the file to ws3d_amd/build.py SOURCES and declare ws3d_gemm_rows / ws3d_gemm_rows_workspace_bytes in include/ws3d_ops.h.
// gemm_rows.hip -- the pointwise layers that are plain matrix products, on our own fp32 matrix-core kernel:
//   out[r, o] = act( sum_k X[r, k] Wt[k, o] + bias[o] )          X (rows, K) row-major, Wt (K, O) = W^T row-major
// These are the per-point products of the Stage-1 network (P = feats @ W_f of SA2..SA4, Q = known_feats @ W_a and the skip products
// of the FP modules, the FP modules' second layers): 14 launches per batch that round 3 left to the library (hipBLASLt through
// torch.mm / addmm / _addmm_activation, 77 TFLOP/s on average over the step's shapes).  Owning them takes the library -- its
// per-process solution choice, its workspace, TunableOp -- out of the inference step: the summation order is fixed by this file.
// Same tile skeleton as gemm_pool_big_kernel (gemm_pool.hip): (64 MB) x (64 NB) output tile per 256-thread workgroup, 2 x 2 waves
// of MB x NB accumulators of 32 x 32 (v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate -- the precision class of the library's
// fp32 GEMM, another summation order), K in steps of 16 through double-buffered LDS (X tile k-major, padded), global loads of
// the next step in flight under the matrix instructions of the current one.  XCD-aware tile order (workgroup g runs on XCD
// g % 8): the column tiles of a row tile, which read the same rows of X, sit next to each other on one XCD's L2.
// SPLIT > 1 (few output tiles, long K -- the deepest FP module: 512 x 1024 -> 512): the K range in SPLIT slices by SPLIT
// workgroups per tile, partial tiles to a workspace, and the workgroup that arrives LAST at the tile's ticket adds the slices in
// slice order (not arrival order), bias and activation on top: deterministic whatever the schedule.
#include "common.h"

namespace ws3d {

typedef float gr_f16 __attribute__((ext_vector_type(16)));
constexpr int GR_KT = 16;

template <int MB, int NB>
__global__ __launch_bounds__(256) void gemm_rows_kernel(int k_dim, int o_dim, const float *__restrict__ x, const float *__restrict__ wt,
                                                        const float *__restrict__ bias, int relu, float *__restrict__ out, int out_stride,
                                                        int split, float *__restrict__ part, int *__restrict__ ticket) {
    constexpr int TM = 64 * MB, TN = 64 * NB, XS = TM + 1;
    __shared__ float xs[2][GR_KT][XS];        // [k][row]
    __shared__ float ws[2][GR_KT][TN];        // [k][col]
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int col_tiles = o_dim / TN;
    // tile order: g -> (slice, row tile, col tile); the 8 workgroups g .. g + 7 run on the 8 XCDs: consecutive j = g >> 3 walk the
    // column tiles (and slices) of ONE row tile on one XCD
    long g = blockIdx.x;
    const long j = g >> 3;
    const int per_row = col_tiles * split;
    const int cs = (int)(j % per_row);
    const int col_tile = cs % col_tiles, slice = cs / col_tiles;
    const long row_tile = (j / per_row) * 8 + (g & 7);
    const long row0 = row_tile * TM;
    const int col0 = col_tile * TN;
    const int k_per = ((k_dim / split + GR_KT - 1) / GR_KT) * GR_KT;         // slice length (a multiple of the k step)
    const int k_lo = slice * k_per, k_hi = min(k_dim, k_lo + k_per);
    float4 xv[MB], wv[NB];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int idx = tid + 256 * i, r = idx >> 2, k = k0 + (idx & 3) * 4;
            xv[i] = k < k_hi ? *reinterpret_cast<const float4 *>(x + (row0 + r) * (long)k_dim + k) : make_float4(0.f, 0.f, 0.f, 0.f);   // k_dim % 4 == 0
        }
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
            const int idx = tid + 256 * jj, k = k0 + idx / (16 * NB), c = (idx % (16 * NB)) * 4;
            wv[jj] = k < k_hi ? *reinterpret_cast<const float4 *>(wt + (long)k * o_dim + col0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int idx = tid + 256 * i, r = idx >> 2, k = (idx & 3) * 4;
            xs[buf][k + 0][r] = xv[i].x; xs[buf][k + 1][r] = xv[i].y; xs[buf][k + 2][r] = xv[i].z; xs[buf][k + 3][r] = xv[i].w;
        }
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
            const int idx = tid + 256 * jj;
            *reinterpret_cast<float4 *>(&ws[buf][idx / (16 * NB)][(idx % (16 * NB)) * 4]) = wv[jj];
        }
    };
    gr_f16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int jj = 0; jj < NB; ++jj)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][jj][v] = 0.f;
    load(k_lo);
    stage(0);
    __syncthreads();
    const int ntiles = (k_hi - k_lo + GR_KT - 1) / GR_KT;
    const int ar = wm * 32 * MB + (lane & 31), bc = wn * 32 * NB + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) load(k_lo + (t + 1) * GR_KT);
#pragma unroll
        for (int k = 0; k < GR_KT; k += 2) {
            float a[MB], bq[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) a[i] = xs[cur][k + kh][ar + 32 * i];
#pragma unroll
            for (int jj = 0; jj < NB; ++jj) bq[jj] = ws[cur][k + kh][bc + 32 * jj];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int jj = 0; jj < NB; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[jj], acc[i][jj], 0, 0, 0);
        }
        if (t + 1 < ntiles) stage(cur ^ 1);
        __syncthreads();
    }
    // accumulator layout (32 x 32 tile): register v of lane l holds row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32
    auto rrow = [&](int i, int v) { return row0 + wm * 32 * MB + 32 * i + 8 * (v >> 2) + 4 * kh + (v & 3); };
    if (split > 1) {
        // this slice's partial tile -> workspace [tile][slice][TM][TN]; the last arrival sums the slices in slice order
        const long tile_id = row_tile * col_tiles + col_tile;
        float *mine = part + ((tile_id * split + slice) * (long)TM) * TN;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int jj = 0; jj < NB; ++jj)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                    mine[(long)(wm * 32 * MB + 32 * i + 8 * (v >> 2) + 4 * kh + (v & 3)) * TN + wn * 32 * NB + 32 * jj + (lane & 31)] = acc[i][jj][v];
        __threadfence();
        __syncthreads();
        if (tid == 0) last_s = atomicAdd(&ticket[tile_id], 1) == split - 1;
        __syncthreads();
        if (!last_s) return;
        __threadfence();
        if (tid == 0) ticket[tile_id] = 0;                                   // (ready for the next launch on the same stream)
        const float *base = part + (tile_id * split * (long)TM) * TN;
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int jj = 0; jj < NB; ++jj)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const long off = (long)(wm * 32 * MB + 32 * i + 8 * (v >> 2) + 4 * kh + (v & 3)) * TN + wn * 32 * NB + 32 * jj + (lane & 31);
                    float s = 0.f;
                    for (int sl = 0; sl < split; ++sl) s += __builtin_nontemporal_load(base + (long)sl * TM * TN + off);
                    acc[i][jj][v] = s;
                }
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int jj = 0; jj < NB; ++jj) {
            const int col = col0 + bc + 32 * jj;
            const float bv = bias ? bias[col] : 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float y = acc[i][jj][v] + bv;
                if (relu) y = y < 0.f ? 0.f : y;          // (NaN stays NaN like the library's ReLU epilogue: the comparison is false)
                out[rrow(i, v) * out_stride + col] = y;
            }
        }
}

}  // namespace ws3d

// workspace of the split-K form for this shape (0: the launch does not split): partial tiles only; the tickets are a separate int32
// array of (rows / 64) * (o / 64) entries that must be ZERO on entry and is left zero
static int gr_split(long rows, int k_dim, int o_dim) {
    const bool big = rows % 1024 == 0 && o_dim % 128 == 0 && (rows / 128) * (o_dim / 128) >= 512;
    if (big || rows % 512 != 0) return 1;
    const long tiles = (rows / 64) * (o_dim / 64);
    int split = 1;
    while (split < 8 && tiles * split < 512 && k_dim / (split * 2) >= 64) split *= 2;
    return split;
}

extern "C" size_t ws3d_gemm_rows_workspace_bytes(long rows, int k_dim, int o_dim) {
    if (rows <= 0 || o_dim <= 0 || k_dim <= 0 || (rows & 63) || (o_dim & 63)) return 0;
    const int split = gr_split(rows, k_dim, o_dim);
    return split > 1 ? (size_t)rows * (size_t)o_dim * 4 * (size_t)split : 0;
}

extern "C" int ws3d_gemm_rows(long rows, int k_dim, int o_dim, const float *x_rows, const float *wt, const float *bias, int relu, float *out,
                              int out_stride, void *workspace, size_t workspace_bytes, int32_t *tickets, ws3d_stream_t stream) {
    using namespace ws3d;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x_rows) | reinterpret_cast<uintptr_t>(wt);
    if (rows < 0 || k_dim <= 0 || (k_dim & 3) || o_dim <= 0 || (o_dim & 63) || (rows & 511) || !x_rows || !wt || !out || (al & 15) || out_stride < o_dim ||
        rows / 64 > (1L << 24)) {
        set_error("ws3d_gemm_rows: unsupported shape (rows=%ld k=%d o=%d; rows %% 512, o %% 64, k %% 4, 16-byte aligned operands)", rows, k_dim, o_dim);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    hipStream_t st = as_stream(stream);
    // tile: 128 x 128 while that leaves >= 2 workgroups per CU, else 64 x 64; few tiles of 64 x 64 and a long K: split K (given a workspace)
    const bool big = rows % 1024 == 0 && o_dim % 128 == 0 && (rows / 128) * (o_dim / 128) >= 512;
    if (big) {
        const unsigned grid = (unsigned)((rows / 128) * (o_dim / 128));
        hipLaunchKernelGGL((gemm_rows_kernel<2, 2>), dim3(grid), dim3(256), 0, st, k_dim, o_dim, x_rows, wt, bias, relu, out, out_stride, 1, nullptr, nullptr);
        return check_launch("ws3d_gemm_rows");
    }
    const long tiles = (rows / 64) * (o_dim / 64);
    int split = gr_split(rows, k_dim, o_dim);
    if (split > 1 && (!workspace || !tickets || workspace_bytes < (size_t)rows * o_dim * 4 * split)) split = 1;
    hipLaunchKernelGGL((gemm_rows_kernel<1, 1>), dim3((unsigned)(tiles * split)), dim3(256), 0, st, k_dim, o_dim, x_rows, wt, bias, relu, out, out_stride,
                       split, static_cast<float *>(workspace), tickets);
    return check_launch("ws3d_gemm_rows");
}
