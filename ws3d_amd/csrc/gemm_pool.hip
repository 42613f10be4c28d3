// gemm_pool.hip -- last SharedMLP layer of a set-abstraction scale fused with the pool over nsample:
//   out[g, o] = relu( max_{s < ns} ( X[g*ns + s, :] . Wt[:, o] ) + bias[o] )
// (bias add and ReLU are monotone, so they commute with the max exactly, like nn_blocks.forward_then_max).
// The separate launches write the (rows, O) activation with the GEMM and read it back for the pool:
// 351 MB each way per batch of 8 scenes at SA2..SA4.  Here a 64 x 64 output tile per workgroup is
// accumulated on the matrix cores (v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate -- same precision
// class as the library GEMM, different summation order) and reduced over its row groups in registers;
// only the pooled rows reach HBM.
#include <cstdlib>

#include "common.h"
#include "compact_pool.h"

namespace ws3d {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int GP_KT = 16;          // K step per LDS tile
constexpr int GP_XS = 65;          // padded row length of the k-major X tile

__device__ __forceinline__ float gp_nanmax(float a, float b) { return (a > b || a != a) ? a : b; }

// grid (O / 64, R / 64), 256 threads = 4 waves as 2 (rows) x 2 (cols) sub-tiles of 32 x 32
template <int NS>
__global__ __launch_bounds__(256) void gemm_pool_kernel(int k_dim, int o_dim, const float *__restrict__ x,
                                                        const float *__restrict__ wt, const float *__restrict__ bias,
                                                        int relu, float *__restrict__ out, int out_stride, int xcd,
                                                        const int32_t *__restrict__ gate, long gate_limit) {
    if (gate && (long)*gate <= gate_limit) return;        // device-side dispatch (ws3d_ops.h "launch gates"): the compact kernels run instead
    __shared__ float xs[2][GP_KT][GP_XS];     // [k][row]
    __shared__ float ws[2][GP_KT][64];        // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    // 1-D grid; the col tiles of a row tile read the SAME activation rows: next to each other on ONE XCD (workgroup g runs on
    // XCD g % 8) the second .. eighth read hits that L2 (with col tiles on blockIdx.x they ran on different XCDs and the
    // activation -- 100 MB at SA2 -- came out of HBM once per col tile)
    const int col_tiles = o_dim / 64;
    long row_tile;
    int col_tile;
    {
        const long g = blockIdx.x;
        if (xcd) {
            const long j = g >> 3;
            col_tile = (int)(j % col_tiles);
            row_tile = (j / col_tiles) * 8 + (g & 7);
        } else {
            col_tile = (int)(g % col_tiles);
            row_tile = g / col_tiles;
        }
    }
    const long row0 = row_tile * 64;
    const int col0 = col_tile * 64;
    // global -> register staging: X tile 64 rows x 16 k (one float4 per thread), W tile 16 k x 64 cols
    const int xr = tid >> 2, xk = (tid & 3) * 4;
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const float *xp = x + (row0 + xr) * (long)k_dim;
    auto load_x = [&](int k0) {
        const int k = k0 + xk;
        return k < k_dim ? *reinterpret_cast<const float4 *>(xp + k) : make_float4(0.f, 0.f, 0.f, 0.f);   // k_dim % 4 == 0
    };
    auto load_w = [&](int k0) {
        const int k = k0 + wk;
        return k < k_dim ? *reinterpret_cast<const float4 *>(wt + (long)k * o_dim + col0 + wc) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf, const float4 xv, const float4 wv) {
        xs[buf][xk + 0][xr] = xv.x; xs[buf][xk + 1][xr] = xv.y; xs[buf][xk + 2][xr] = xv.z; xs[buf][xk + 3][xr] = xv.w;
        *reinterpret_cast<float4 *>(&ws[buf][wk][wc]) = wv;
    };
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 xv = load_x(0), wv = load_w(0);
    stage(0, xv, wv);
    __syncthreads();
    const int ntiles = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) { xv = load_x((t + 1) * GP_KT); wv = load_w((t + 1) * GP_KT); }
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[cur][k + kh][ar], ws[cur][k + kh][bc], acc, 0, 0, 0);
        if (t + 1 < ntiles) stage(cur ^ 1, xv, wv);
        __syncthreads();
    }
    // accumulator layout (32 x 32 tile): register v of lane l holds row 8*(v/4) + 4*(l/32) + v%4, column l%32
    const int col = col0 + bc;
    const float bv = bias ? bias[col] : 0.f;
    if (NS == 16) {
        float m0 = acc[0], m1 = acc[8];
#pragma unroll
        for (int v = 1; v < 8; ++v) { m0 = gp_nanmax(m0, acc[v]); m1 = gp_nanmax(m1, acc[8 + v]); }
        m0 = gp_nanmax(m0, __shfl_xor(m0, 32));
        m1 = gp_nanmax(m1, __shfl_xor(m1, 32));
        if (lane < 32) {
            const long g = (row0 + wm * 32) / 16;
            float r0 = m0 + bv, r1 = m1 + bv;
            if (relu) { r0 = r0 < 0.f ? 0.f : r0; r1 = r1 < 0.f ? 0.f : r1; }
            out[g * out_stride + col] = r0;
            out[(g + 1) * out_stride + col] = r1;
        }
    } else {   // NS == 32: the whole sub-tile is one group
        float m0 = acc[0];
#pragma unroll
        for (int v = 1; v < 16; ++v) m0 = gp_nanmax(m0, acc[v]);
        m0 = gp_nanmax(m0, __shfl_xor(m0, 32));
        if (lane < 32) {
            const long g = (row0 + wm * 32) / 32;
            float r0 = m0 + bv;
            if (relu) r0 = r0 < 0.f ? 0.f : r0;
            out[g * out_stride + col] = r0;
        }
    }
}

// The same product with a (64 MB) x (64 NB) output tile per workgroup: each of the 2 x 2 waves holds MB x NB accumulators of
// 32 x 32, one k-step costs MB + NB LDS reads for MB NB matrix instructions, and the L2 -> LDS traffic per flop falls with the
// tile (64 x 64: 16 flop per byte, 128 x 128: 32).  Measured on the six last-layer shapes (scripts/ubench/gemm_pool_tiles.sh,
// profiles/r02_gemm_pool_tiles.txt): L2 requests and LDS cycles halve at 128 x 128, the kernel time does not move (0.288 vs
// 0.293 ms in total): SQ_VALU_MFMA_BUSY_CYCLES = 64 clk x the instruction count at every tile, i.e. the matrix pipe is busy
// 77 % of the kernel at the 2.0 GHz the chip sustains under this kernel (s_memtime / s_memrealtime hooks, profiles/r02_gemm_pool_wave_timeline.txt; a
// register-only loop of the same instruction holds 2.35 GHz and 154 TFLOP/s, scripts/ubench/mfma_f32_peak.hip).
template <int NS, int MB, int NB>
__global__ __launch_bounds__(256) void gemm_pool_big_kernel(int k_dim, int o_dim, const float *__restrict__ x,
                                                            const float *__restrict__ wt, const float *__restrict__ bias,
                                                            int relu, float *__restrict__ out, int out_stride, int xcd,
                                                            const int32_t *__restrict__ gate, long gate_limit) {
    if (gate && (long)*gate <= gate_limit) return;
    constexpr int TM = 64 * MB, TN = 64 * NB, XS = TM + 1;
    __shared__ float xs[2][GP_KT][XS];        // [k][row]
    __shared__ float ws[2][GP_KT][TN];        // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int col_tiles = o_dim / TN;
    long row_tile;
    int col_tile;
    {
        const long g = blockIdx.x;
        if (xcd) {
            const long j = g >> 3;
            col_tile = (int)(j % col_tiles);
            row_tile = (j / col_tiles) * 8 + (g & 7);
        } else {
            col_tile = (int)(g % col_tiles);
            row_tile = g / col_tiles;
        }
    }
    const long row0 = row_tile * TM;
    const int col0 = col_tile * TN;
    float4 xv[MB], wv[NB];
    auto load = [&](int k0) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int idx = tid + 256 * i, r = idx >> 2, k = k0 + (idx & 3) * 4;
            xv[i] = k < k_dim ? *reinterpret_cast<const float4 *>(x + (row0 + r) * (long)k_dim + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j, k = k0 + idx / (16 * NB), c = (idx % (16 * NB)) * 4;
            wv[j] = k < k_dim ? *reinterpret_cast<const float4 *>(wt + (long)k * o_dim + col0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int idx = tid + 256 * i, r = idx >> 2, k = (idx & 3) * 4;
            xs[buf][k + 0][r] = xv[i].x; xs[buf][k + 1][r] = xv[i].y; xs[buf][k + 2][r] = xv[i].z; xs[buf][k + 3][r] = xv[i].w;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4 *>(&ws[buf][idx / (16 * NB)][(idx % (16 * NB)) * 4]) = wv[j];
        }
    };
    floatx16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    load(0);
    stage(0);
    __syncthreads();
    const int ntiles = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 * MB + (lane & 31), bc = wn * 32 * NB + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) load((t + 1) * GP_KT);
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2) {
            float a[MB], bq[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) a[i] = xs[cur][k + kh][ar + 32 * i];
#pragma unroll
            for (int j = 0; j < NB; ++j) bq[j] = ws[cur][k + kh][bc + 32 * j];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < ntiles) stage(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = col0 + bc + 32 * j;
            const float bv = bias ? bias[col] : 0.f;
            const long rbase = row0 + wm * 32 * MB + 32 * i;
            if (NS == 16) {
                float m0 = acc[i][j][0], m1 = acc[i][j][8];
#pragma unroll
                for (int v = 1; v < 8; ++v) { m0 = gp_nanmax(m0, acc[i][j][v]); m1 = gp_nanmax(m1, acc[i][j][8 + v]); }
                m0 = gp_nanmax(m0, __shfl_xor(m0, 32));
                m1 = gp_nanmax(m1, __shfl_xor(m1, 32));
                if (lane < 32) {
                    const long g = rbase / 16;
                    float r0 = m0 + bv, r1 = m1 + bv;
                    if (relu) { r0 = r0 < 0.f ? 0.f : r0; r1 = r1 < 0.f ? 0.f : r1; }
                    out[g * out_stride + col] = r0;
                    out[(g + 1) * out_stride + col] = r1;
                }
            } else {
                float m0 = acc[i][j][0];
#pragma unroll
                for (int v = 1; v < 16; ++v) m0 = gp_nanmax(m0, acc[i][j][v]);
                m0 = gp_nanmax(m0, __shfl_xor(m0, 32));
                if (lane < 32) {
                    const long g = rbase / 32;
                    float r0 = m0 + bv;
                    if (relu) r0 = r0 < 0.f ? 0.f : r0;
                    out[g * out_stride + col] = r0;
                }
            }
        }
}

// XCD-aware tile order for the gather-GEMMs (1-D grid of col_tiles * row_tiles workgroups): workgroup g runs on XCD g % 8
// (observed dispatch order).  With tps row tiles per scene and a batch that is a multiple of 8,
//   scene = (g / 8 / (col_tiles * tps)) * 8 + g % 8,   row tile = (g / 8 / col_tiles) % tps,   col tile = (g / 8) % col_tiles
// keeps (i) all tiles of a scene on ONE XCD -- the rows they gather (4 MB of features per scene at FP1) stay in that L2 instead
// of being pulled through all eight -- and (ii) the col tiles of a row tile, which gather the SAME rows, next to each other on
// that XCD (with the 2-D grid they ran on col_tiles different XCDs and each fetched the rows again).  tps = 0: plain order.
__device__ __forceinline__ void gg_tile(int col_tiles, int tps, long &row_tile, int &col_tile, long g = -1) {
    if (g < 0) g = blockIdx.x;
    if (tps > 0) {
        const long j = g >> 3;
        col_tile = (int)(j % col_tiles);
        const long jj = j / col_tiles;
        row_tile = ((jj / tps) * 8 + (g & 7)) * tps + jj % tps;
    } else {
        col_tile = (int)(g % col_tiles);
        row_tile = g / col_tiles;
    }
}

// ---- first SharedMLP layer of a set-abstraction scale with the GROUPING fused into its A operand:
//   out[r, o] = relu?( sum_k X[r, k] Wt[k, o] + bias[o] ),   r = (scene b, centre m, sample s),
//   X[r, 0:C] = feats[b, nbr[b, m, s], :]   X[r, C:C+3] = xyz[b, nbr[b, m, s]] - new_xyz[b, m]
// i.e. QueryAndGroup (pointnet2_utils.py:241-264, channels-last, the xyz block behind the features: the caller
// permutes the weight rows once) feeding layer 1 without the (rows, 3+C) grouped tensor ever existing: 156 MB per batch
// written by the grouping kernel and read back by the GEMM at SA2, 102 MB at SA3.  Same 64 x 64 tile / LDS / MFMA
// structure as gemm_pool_kernel; the X tile loader follows the neighbour list instead of a row pointer (a 16-byte load
// of 4 consecutive channels of the neighbour's feature row: C % 4 == 0 keeps it aligned and makes k == C a chunk start).
__global__ __launch_bounds__(256) void gather_gemm_kernel(int c_feat, int o_dim, int n, int m, int ns, const float *__restrict__ feats,
                                                          const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                          const int32_t *__restrict__ nbr, const float *__restrict__ wt,
                                                          const float *__restrict__ bias, int relu, float *__restrict__ out, int tps) {
    __shared__ float xs[2][GP_KT][GP_XS];     // [k][row]
    __shared__ float ws[2][GP_KT][64];        // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    long row_tile;
    int col_tile;
    gg_tile(o_dim / 64, tps, row_tile, col_tile);
    const long row0 = row_tile * 64;
    const int col0 = col_tile * 64;
    const int k_dim = c_feat + 3;
    const int xr = tid >> 2, xk = (tid & 3) * 4;
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    // this thread's row of the tile: (scene, centre, sample) -> neighbour
    const long r = row0 + xr;
    const long cm = r / ns;                                   // scene * m + centre
    const long b = cm / m;
    const int src = nbr[r];
    const float *frow = feats + ((size_t)b * n + (size_t)src) * c_feat;
    const float *prow = xyz + ((size_t)b * n + (size_t)src) * 3;
    const float *crow = new_xyz + (size_t)cm * 3;
    auto load_x = [&](int k0) {
        const int k = k0 + xk;
        if (k < c_feat) return *reinterpret_cast<const float4 *>(frow + k);
        if (k == c_feat) return make_float4(prow[0] - crow[0], prow[1] - crow[1], prow[2] - crow[2], 0.f);   // grouped_xyz -= new_xyz
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto load_w = [&](int k0) {
        const int k = k0 + wk;
        return k < k_dim ? *reinterpret_cast<const float4 *>(wt + (long)k * o_dim + col0 + wc) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf, const float4 xv, const float4 wv) {
        xs[buf][xk + 0][xr] = xv.x; xs[buf][xk + 1][xr] = xv.y; xs[buf][xk + 2][xr] = xv.z; xs[buf][xk + 3][xr] = xv.w;
        *reinterpret_cast<float4 *>(&ws[buf][wk][wc]) = wv;
    };
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 xv = load_x(0), wv = load_w(0);
    stage(0, xv, wv);
    __syncthreads();
    const int ntiles = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) { xv = load_x((t + 1) * GP_KT); wv = load_w((t + 1) * GP_KT); }
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[cur][k + kh][ar], ws[cur][k + kh][bc], acc, 0, 0, 0);
        if (t + 1 < ntiles) stage(cur ^ 1, xv, wv);
        __syncthreads();
    }
    // accumulator layout (32 x 32 tile): register v of lane l holds row 8*(v/4) + 4*(l/32) + v%4, column l%32
    const int col = col0 + bc;
    const float bv = bias ? bias[col] : 0.f;
    float *o = out + (row0 + wm * 32 + 4 * (lane >> 5)) * (long)o_dim + col;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float y = acc[v] + bv;
        if (relu) y = y < 0.f ? 0.f : y;       // NaN stays NaN, like relu_
        o[(long)(8 * (v / 4) + (v % 4)) * o_dim] = y;
    }
}

// ---- the first TWO layers of a set-abstraction SharedMLP in one kernel: layer 1 as in gather_gemm_kernel (grouping fused into
// the A operand), its 64 x O1 activation tile kept in LDS -- [channel][row], the layout the 32 x 32 accumulators write without
// bank conflicts and layer 2's A operand reads -- and layer 2 (O1 -> O2) multiplied straight out of it, 64 output columns at a
// time.  Neither the grouped tensor nor the first activation (rows x O1) reaches HBM, and the rows of a tile are gathered once
// instead of once per 64 output columns.  NB1 = O1 / 64 accumulators per wave in phase 1.
// POOL = 16 | 32 (= ns): the THIRD layer and the pool over nsample follow in the same kernel -- layer 2's 64 x O2 tile goes to LDS
// next to layer 1's instead of to HBM, layer 3 (O2 -> O3) is multiplied out of it 64 columns at a time and reduced over its row
// groups in registers as in gemm_pool_kernel: of a whole SharedMLP only the pooled rows (rows / ns, O3) reach HBM.
template <int NB1, int POOL = 0>
__global__ __launch_bounds__(256) void gather_gemm2_kernel(int c_feat, int o2, int n, int m, int ns, const float *__restrict__ feats,
                                                           const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const int32_t *__restrict__ nbr, const float *__restrict__ w1t,
                                                           const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                           const float *__restrict__ b2, int relu2, float *__restrict__ out,
                                                           int o3 = 0, const float *__restrict__ w3t = nullptr,
                                                           const float *__restrict__ b3 = nullptr, int relu3 = 0, int out_stride = 0) {
    constexpr int O1 = NB1 * 64;
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    // phase 1: xs[2][GP_KT][GP_XS] | w1s[2][GP_KT][O1];   phase 2 (aliases phase 1): act[O1][GP_XS] | w2s[2][GP_KT][64]
    // with POOL: act[O1][GP_XS] | act2[O2P][GP_XS] | w2s[2][GP_KT][64],  O2P = o2 rounded up to the k-tile; layer 3's W tiles
    // ([2][GP_KT][128]) reuse act
    const int o2p = POOL ? (o2 + GP_KT - 1) / GP_KT * GP_KT : 0;
    float *xs = smem2, *w1s = smem2 + 2 * GP_KT * GP_XS;
    float *act = smem2, *act2 = smem2 + O1 * GP_XS, *w2s = smem2 + (O1 + o2p) * GP_XS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const long row0 = (long)blockIdx.x * 64;
    const int k_dim = c_feat + 3;
    const int xr = tid >> 2, xk = (tid & 3) * 4;
    const long r = row0 + xr;
    const long cm = r / ns;                                   // scene * m + centre
    const long b = cm / m;
    const int src = nbr[r];
    const float *frow = feats + ((size_t)b * n + (size_t)src) * c_feat;
    const float *prow = xyz + ((size_t)b * n + (size_t)src) * 3;
    const float *crow = new_xyz + (size_t)cm * 3;
    auto load_x = [&](int k0) {
        const int k = k0 + xk;
        if (k < c_feat) return *reinterpret_cast<const float4 *>(frow + k);
        if (k == c_feat) return make_float4(prow[0] - crow[0], prow[1] - crow[1], prow[2] - crow[2], 0.f);   // grouped_xyz -= new_xyz
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };
    float4 wv[NB1];
    auto load_w1 = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            const int i = tid + 256 * j, k = k0 + i / (O1 / 4), c4 = i % (O1 / 4);
            wv[j] = k < k_dim ? *reinterpret_cast<const float4 *>(w1t + (long)k * O1 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage1 = [&](int buf, const float4 xv) {
        float *x = xs + buf * GP_KT * GP_XS;
        x[(xk + 0) * GP_XS + xr] = xv.x; x[(xk + 1) * GP_XS + xr] = xv.y; x[(xk + 2) * GP_XS + xr] = xv.z; x[(xk + 3) * GP_XS + xr] = xv.w;
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            const int i = tid + 256 * j;
            *reinterpret_cast<float4 *>(w1s + buf * GP_KT * O1 + (i / (O1 / 4)) * O1 + (i % (O1 / 4)) * 4) = wv[j];
        }
    };
    floatx16 acc1[NB1];
#pragma unroll
    for (int j = 0; j < NB1; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc1[j][i] = 0.f;
    float4 xv = load_x(0);
    load_w1(0);
    stage1(0, xv);
    __syncthreads();
    const int nt1 = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < nt1; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt1) { xv = load_x((t + 1) * GP_KT); load_w1((t + 1) * GP_KT); }
        const float *x = xs + cur * GP_KT * GP_XS, *wl = w1s + cur * GP_KT * O1;
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2) {
            const float a = x[(k + kh) * GP_XS + ar];
#pragma unroll
            for (int j = 0; j < NB1; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[(k + kh) * O1 + j * 64 + bc], acc1[j], 0, 0, 0);
        }
        if (t + 1 < nt1) stage1(cur ^ 1, xv);
        __syncthreads();
    }
    // layer-1 epilogue into the activation tile (accumulator layout: register v of lane l = row 8*(v/4) + 4*(l/32) + v%4, column l%32)
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
        const int col = j * 64 + bc;
        const float bv = b1 ? b1[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float y = acc1[j][v] + bv;
            if (relu1) y = y < 0.f ? 0.f : y;
            act[col * GP_XS + wm * 32 + 8 * (v / 4) + 4 * (lane >> 5) + (v % 4)] = y;
        }
    }
    // layer 2, 64 output columns per pass; W2 tiles (16 x 64) double-buffered
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const int nchunk = (o2 + 63) / 64;
    constexpr int nt2 = O1 / GP_KT;
    for (int c = 0; c < nchunk; ++c) {
        const int col0 = c * 64;
        auto load_w2 = [&](int t) {
            const int col = col0 + wc;      // o2 % 4 == 0: a float4 is inside or outside as a whole
            return col < o2 ? *reinterpret_cast<const float4 *>(w2t + (long)(t * GP_KT + wk) * o2 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        floatx16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
        float4 w2v = load_w2(0);
        __syncthreads();                    // the activation tile is complete / the previous pass has left w2s
        *reinterpret_cast<float4 *>(w2s + wk * 64 + wc) = w2v;
        __syncthreads();
        for (int t = 0; t < nt2; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt2) w2v = load_w2(t + 1);
            const float *wl = w2s + cur * GP_KT * 64;
#pragma unroll
            for (int k = 0; k < GP_KT; k += 2)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(act[(t * GP_KT + k + kh) * GP_XS + ar], wl[(k + kh) * 64 + bc], acc2, 0, 0, 0);
            if (t + 1 < nt2) *reinterpret_cast<float4 *>(w2s + (cur ^ 1) * GP_KT * 64 + wk * 64 + wc) = w2v;
            __syncthreads();
        }
        const int col = col0 + bc;
        if (POOL) {
            if (col < o2p) {          // columns o2 .. o2p - 1: the zero padding of layer 3's k dimension
                const float bv = (col < o2 && b2) ? b2[col] : 0.f;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float y = acc2[v] + bv;
                    if (relu2) y = y < 0.f ? 0.f : y;
                    act2[col * GP_XS + wm * 32 + 8 * (v / 4) + 4 * (lane >> 5) + (v % 4)] = col < o2 ? y : 0.f;
                }
            }
        } else if (col < o2) {
            const float bv = b2 ? b2[col] : 0.f;
            float *o = out + (row0 + wm * 32 + 4 * (lane >> 5)) * (long)o2 + col;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float y = acc2[v] + bv;
                if (relu2) y = y < 0.f ? 0.f : y;
                o[(long)(8 * (v / 4) + (v % 4)) * o2] = y;
            }
        }
    }
    if (POOL) {
        // layer 3 out of act2, 128 output columns per pass (o3 % 128 == 0; two accumulators per wave: half the barriers and
        // half the reads of the activation of a 64-column pass), then the max over each group of POOL rows
        const int nt3 = o2p / GP_KT;
        const int wk3 = tid >> 5, wc3 = (tid & 31) * 4;          // W3 tile 16 x 128: two float4 per thread (rows wk3, wk3 + 8)
        float *w3s = smem2;                                      // [2][GP_KT][128] in layer 1's activation tile, which is dead by now (O1 * GP_XS >= 4096)
        for (int c = 0; c < o3 / 128; ++c) {
            const int col0 = c * 128;
            float4 w3v[2];
            auto load_w3 = [&](int t) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int k = t * GP_KT + wk3 + 8 * q;
                    w3v[q] = k < o2 ? *reinterpret_cast<const float4 *>(w3t + (long)k * o3 + col0 + wc3) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            auto stage_w3 = [&](int buf) {
#pragma unroll
                for (int q = 0; q < 2; ++q) *reinterpret_cast<float4 *>(w3s + buf * GP_KT * 128 + (wk3 + 8 * q) * 128 + wc3) = w3v[q];
            };
            floatx16 acc3[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc3[q][i] = 0.f;
            load_w3(0);
            __syncthreads();                // act2 is complete, act is dead / the previous pass has left w3s
            stage_w3(0);
            __syncthreads();
            for (int t = 0; t < nt3; ++t) {
                const int cur = t & 1;
                if (t + 1 < nt3) load_w3(t + 1);
                const float *wl = w3s + cur * GP_KT * 128;
#pragma unroll
                for (int k = 0; k < GP_KT; k += 2) {
                    const float a = act2[(t * GP_KT + k + kh) * GP_XS + ar];
#pragma unroll
                    for (int q = 0; q < 2; ++q) acc3[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[(k + kh) * 128 + q * 64 + bc], acc3[q], 0, 0, 0);
                }
                if (t + 1 < nt3) stage_w3(cur ^ 1);
                __syncthreads();
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int col = col0 + q * 64 + bc;
                const float bv = b3 ? b3[col] : 0.f;
                if (POOL == 16) {
                    float m0 = acc3[q][0], m1 = acc3[q][8];
#pragma unroll
                    for (int v = 1; v < 8; ++v) { m0 = gp_nanmax(m0, acc3[q][v]); m1 = gp_nanmax(m1, acc3[q][8 + v]); }
                    m0 = gp_nanmax(m0, __shfl_xor(m0, 32));
                    m1 = gp_nanmax(m1, __shfl_xor(m1, 32));
                    if (lane < 32) {
                        const long g = (row0 + wm * 32) / 16;
                        float r0 = m0 + bv, r1 = m1 + bv;
                        if (relu3) { r0 = r0 < 0.f ? 0.f : r0; r1 = r1 < 0.f ? 0.f : r1; }
                        out[g * out_stride + col] = r0;
                        out[(g + 1) * out_stride + col] = r1;
                    }
                } else {
                    float m0 = acc3[q][0];
#pragma unroll
                    for (int v = 1; v < 16; ++v) m0 = gp_nanmax(m0, acc3[q][v]);
                    m0 = gp_nanmax(m0, __shfl_xor(m0, 32));
                    if (lane < 32) {
                        const long g = (row0 + wm * 32) / 32;
                        float r0 = m0 + bv;
                        if (relu3) r0 = r0 < 0.f ? 0.f : r0;
                        out[g * out_stride + col] = r0;
                    }
                }
            }
        }
    }
}

// ---- layer 1 of a set-abstraction SharedMLP WITHOUT its per-pair product.  The layer is linear in the grouped row,
//   W [f_j ; x_j - c] = W_f f_j + W_x (x_j - c),
// and W_f f_j depends on the SOURCE point j only: P = feats @ W_f is one small GEMM over the n points of a scene instead of a
// product over its m * ns (centre, sample) pairs -- 12 x fewer rows at SA2..SA4 (ns = 16 + 32 pairs per centre, n = 4 m), i.e.
// 15 % of all the fp32 matrix work of the network.  What is left per pair is a gather of P's row (O1 floats instead of the C
// feature channels: half the bytes), the three-term xyz product -- kept in the centred form, so nothing cancels -- bias and ReLU.
// pgather_gemm2_kernel: that, feeding layer 2 as in gather_gemm2_kernel (the accumulators of layer 1 START at the gathered P
// values and take two matrix steps for [dx dy dz 0]); pgather_rows_kernel: layer 1 alone (rows x O1), for the widest level.
template <int NB1>
__global__ __launch_bounds__(256) void pgather_gemm2_kernel(int o2, int n, int m, int ns, const float *__restrict__ pmat, int p_stride,
                                                            const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                            const int32_t *__restrict__ nbr, const float *__restrict__ w1x,
                                                            const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                            const float *__restrict__ b2, int relu2, float *__restrict__ out,
                                                            const int32_t *__restrict__ gate, long gate_limit) {
    if (gate && (long)*gate <= gate_limit) return;
    constexpr int O1 = NB1 * 64;
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    float *act = smem2, *w2s = smem2 + O1 * GP_XS;           // act[O1][GP_XS] | w2s[2][GP_KT][64]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const long row0 = (long)blockIdx.x * 64;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    const long scene = row0 / ((long)m * ns);               // a 64-row tile lies inside one scene (m * ns % 64 == 0)
    // this lane's row of the A operand: the centred coordinates, k = 0..3 -> (dx, dy | dz, 0) over the two halves of the wave
    float a0, a1;
    {
        const long r = row0 + ar;
        const long cm = r / ns;
        const int src = nbr[r];
        const float *pr = xyz + ((size_t)scene * n + (size_t)src) * 3, *cr = new_xyz + (size_t)cm * 3;
        const float dx = pr[0] - cr[0], dy = pr[1] - cr[1], dz = pr[2] - cr[2];
        a0 = kh ? dy : dx;
        a1 = kh ? 0.f : dz;
    }
    // accumulators start at P[source point of the row][column]: register v of lane l = row 8 (v / 4) + 4 (l / 32) + v % 4
    floatx16 acc1[NB1];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int src = nbr[row0 + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4)];
        const float *prow = pmat + ((size_t)scene * n + (size_t)src) * p_stride + bc;
#pragma unroll
        for (int j = 0; j < NB1; ++j) acc1[j][v] = prow[j * 64];
    }
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
        const int col = j * 64 + bc;
        const float wb0 = w1x[kh * O1 + col];                         // k = 0 | 1: the x | y row of W_x
        const float wb1 = kh ? 0.f : w1x[2 * O1 + col];               // k = 2 | 3: the z row | the zero pad
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wb0, acc1[j], 0, 0, 0);
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wb1, acc1[j], 0, 0, 0);
        const float bv = b1 ? b1[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float y = acc1[j][v] + bv;
            if (relu1) y = y < 0.f ? 0.f : y;
            act[col * GP_XS + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4)] = y;
        }
    }
    // layer 2, 64 output columns per pass; W2 tiles (16 x 64) double-buffered (as in gather_gemm2_kernel)
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const int nchunk = (o2 + 63) / 64;
    constexpr int nt2 = O1 / GP_KT;
    for (int c = 0; c < nchunk; ++c) {
        const int col0 = c * 64;
        auto load_w2 = [&](int t) {
            const int col = col0 + wc;      // o2 % 4 == 0: a float4 is inside or outside as a whole
            return col < o2 ? *reinterpret_cast<const float4 *>(w2t + (long)(t * GP_KT + wk) * o2 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        floatx16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
        float4 w2v = load_w2(0);
        __syncthreads();                    // the activation tile is complete / the previous pass has left w2s
        *reinterpret_cast<float4 *>(w2s + wk * 64 + wc) = w2v;
        __syncthreads();
        for (int t = 0; t < nt2; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt2) w2v = load_w2(t + 1);
            const float *wl = w2s + cur * GP_KT * 64;
#pragma unroll
            for (int k = 0; k < GP_KT; k += 2)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(act[(t * GP_KT + k + kh) * GP_XS + ar], wl[(k + kh) * 64 + bc], acc2, 0, 0, 0);
            if (t + 1 < nt2) *reinterpret_cast<float4 *>(w2s + (cur ^ 1) * GP_KT * 64 + wk * 64 + wc) = w2v;
            __syncthreads();
        }
        const int col = col0 + bc;
        if (col < o2) {
            const float bv = b2 ? b2[col] : 0.f;
            float *o = out + (row0 + wm * 32 + 4 * kh) * (long)o2 + col;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float y = acc2[v] + bv;
                if (relu2) y = y < 0.f ? 0.f : y;
                o[(long)(8 * (v / 4) + (v % 4)) * o2] = y;
            }
        }
    }
}

// layer 1 alone: out[r, :] = relu?( P[source of r, :] + W_x (x - c) + b1 ); 64 threads x float4 per 256 columns, rows walked
__global__ __launch_bounds__(256) void pgather_rows_kernel(long rows, int o1, int n, int m, int ns, const float *__restrict__ pmat, int p_stride,
                                                           const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const int32_t *__restrict__ nbr, const float *__restrict__ w1x,
                                                           const float *__restrict__ b1, int relu1, float *__restrict__ out) {
    const int q4 = o1 >> 2;                                   // float4 per row
    const long total = rows * q4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / q4;
        const int c4 = (int)(e - r * q4) * 4;
        const long cm = r / ns, scene = cm / m;
        const int src = nbr[r];
        const float *pr = xyz + ((size_t)scene * n + (size_t)src) * 3, *cr = new_xyz + (size_t)cm * 3;
        const float dx = pr[0] - cr[0], dy = pr[1] - cr[1], dz = pr[2] - cr[2];
        const float4 pv = *reinterpret_cast<const float4 *>(pmat + ((size_t)scene * n + (size_t)src) * p_stride + c4);
        const float4 wx = *reinterpret_cast<const float4 *>(w1x + c4), wy = *reinterpret_cast<const float4 *>(w1x + o1 + c4),
                     wz = *reinterpret_cast<const float4 *>(w1x + 2 * o1 + c4);
        const float4 bv = b1 ? *reinterpret_cast<const float4 *>(b1 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 y;
        // the order of the matrix path: P, + dx wx, + dy wy, + dz wz, + bias
        y.x = __builtin_fmaf(dz, wz.x, __builtin_fmaf(dy, wy.x, __builtin_fmaf(dx, wx.x, pv.x))) + bv.x;
        y.y = __builtin_fmaf(dz, wz.y, __builtin_fmaf(dy, wy.y, __builtin_fmaf(dx, wx.y, pv.y))) + bv.y;
        y.z = __builtin_fmaf(dz, wz.z, __builtin_fmaf(dy, wy.z, __builtin_fmaf(dx, wx.z, pv.z))) + bv.z;
        y.w = __builtin_fmaf(dz, wz.w, __builtin_fmaf(dy, wy.w, __builtin_fmaf(dx, wx.w, pv.w))) + bv.w;
        if (relu1) { y.x = y.x < 0.f ? 0.f : y.x; y.y = y.y < 0.f ? 0.f : y.y; y.z = y.z < 0.f ? 0.f : y.z; y.w = y.w < 0.f ? 0.f : y.w; }
        *reinterpret_cast<float4 *>(out + r * (long)o1 + c4) = y;
    }
}

// ---- compact (centre, sample) pairs.  A ball-query list holds its hits in ascending order and is padded with the first one
// (ball_query_gpu.cu:29-44): every padded row of the grouped tensor repeats row 0 of its centre, gives the same activations in
// every layer and cannot change the maximum over nsample.  On FPS-sampled lidar most of a list is padding (synthetic KITTI-shaped
// clouds: 1.0-2.1 distinct neighbours of 16 / 32; centres are spread evenly over space, points are not), so the SharedMLP is run
// over the DISTINCT pairs only: cnt[c] = 1 + #{s >= 1: nbr[c][s] > nbr[c][s-1]}, an exclusive prefix sum gives each centre its
// first compact row, pair_rows_kernel writes (centre, source point) per compact row, the layer kernels walk those rows --
// launched for the worst case, tiles beyond the total (a device-side number) return at once -- and the last layer reduces
// with an atomic max into the centre's row (bias and ReLU first: values >= 0 order as integers; the output starts at 0).
// Bit-identical to the dense path: a row's activations do not depend on which rows share its tile.
__global__ __launch_bounds__(256) void pair_count_kernel(long centres, int ns, const int32_t *__restrict__ nbr, int32_t *__restrict__ cnt) {
    const long c = (long)blockIdx.x * 256 + threadIdx.x;
    if (c >= centres) return;
    const int32_t *row = nbr + c * ns;
    int k = 1, prev = row[0];
    for (int s2 = 1; s2 < ns; ++s2) { const int v = row[s2]; k += v > prev; prev = v; }
    cnt[c] = k;
}

__global__ __launch_bounds__(256) void pair_rows_kernel(long centres, int ns, const int32_t *__restrict__ nbr, const int32_t *__restrict__ cnt,
                                                        const int32_t *__restrict__ incl, int32_t *__restrict__ rowc, int32_t *__restrict__ rowsrc,
                                                        int32_t *__restrict__ total) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= centres * ns) return;
    const long c = e / ns;
    const int s2 = (int)(e - c * ns);
    const int k = cnt[c], first = incl[c] - k;              // incl: inclusive prefix sum of cnt
    if (s2 < k) { rowc[first + s2] = (int32_t)c; rowsrc[first + s2] = nbr[e]; }
    if (e == 0) *total = incl[centres - 1];
}

// count + placement in one launch: a workgroup takes 256 centres, scans their counts in LDS and reserves its stretch of compact rows
// with ONE atomic add on *total (zero on entry).  The stretches of different workgroups land in arrival order -- the ORDER of the
// compact rows is not defined, and nothing depends on it: a row's activations are its own and the maximum is order-free.
__global__ __launch_bounds__(256) void pair_compact_kernel(long centres, int ns, const int32_t *__restrict__ nbr, int32_t *__restrict__ rowc,
                                                           int32_t *__restrict__ rowsrc, int32_t *__restrict__ total) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long c = (long)blockIdx.x * 256 + tid;
    int k = 0;
    if (c < centres) {
        const int32_t *row = nbr + c * ns;
        int prev = row[0];
        k = 1;
        for (int s2 = 1; s2 < ns; ++s2) { const int v = row[s2]; k += v > prev; prev = v; }
    }
    int v = k;
    for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(v, o); if (lane >= o) v += t2; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    if (tid == 0) base_s = atomicAdd(total, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    __syncthreads();
    int first = base_s + v - k;
    for (int i = 0; i < w; ++i) first += wsum[i];
    if (c < centres) {
        const int32_t *row = nbr + c * ns;
        for (int s2 = 0; s2 < k; ++s2) { rowc[first + s2] = (int32_t)c; rowsrc[first + s2] = row[s2]; }
    }
}

// pgather_gemm2_kernel over compact rows: (centre, source) per row from rowc / rowsrc, *total rows in all
template <int NB1>
__device__ __forceinline__ void pgather_gemm2_compact_body(const unsigned bx, const unsigned by, const unsigned gy, int o2, int n, int m,
                                                           const float *__restrict__ pmat, int p_stride,
                                                           const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const int32_t *__restrict__ rowc, const int32_t *__restrict__ rowsrc,
                                                           const int32_t *__restrict__ total, const float *__restrict__ w1x,
                                                           const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                           const float *__restrict__ b2, int relu2, float *__restrict__ out, long limit) {
    constexpr int O1 = NB1 * 64;
    const long T = *total;
    const long row0 = (long)bx * 64;
    if (row0 >= T || (limit >= 0 && T > limit)) return;      // workgroup-uniform; beyond the limit the dense kernels run instead
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    float *act = smem2, *w2s = smem2 + O1 * GP_XS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    float a0, a1;
    {
        const long t = min(row0 + ar, T - 1);                // rows behind the end repeat the last one (never stored)
        const long cm = rowc[t];
        const long scene = cm / m;
        const int src = rowsrc[t];
        const float *pr = xyz + ((size_t)scene * n + (size_t)src) * 3, *cr = new_xyz + (size_t)cm * 3;
        const float dx = pr[0] - cr[0], dy = pr[1] - cr[1], dz = pr[2] - cr[2];
        a0 = kh ? dy : dx;
        a1 = kh ? 0.f : dz;
    }
    floatx16 acc1[NB1];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const long t = min(row0 + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4), T - 1);
        const long scene = rowc[t] / m;
        const float *prow = pmat + ((size_t)scene * n + (size_t)rowsrc[t]) * p_stride + bc;
#pragma unroll
        for (int j = 0; j < NB1; ++j) acc1[j][v] = prow[j * 64];
    }
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
        const int col = j * 64 + bc;
        const float wb0 = w1x[kh * O1 + col];
        const float wb1 = kh ? 0.f : w1x[2 * O1 + col];
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wb0, acc1[j], 0, 0, 0);
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wb1, acc1[j], 0, 0, 0);
        const float bv = b1 ? b1[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float y = acc1[j][v] + bv;
            if (relu1) y = y < 0.f ? 0.f : y;
            act[col * GP_XS + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4)] = y;
        }
    }
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const int nchunk = (o2 + 63) / 64;
    constexpr int nt2 = O1 / GP_KT;
    for (int c = by; c < nchunk; c += gy) {       // the 64-column passes of layer 2 are spread over gridDim.y workgroups (each
        const int col0 = c * 64;                                   // rebuilds layer 1's tile: a gather and two matrix steps): compact launches are small
        auto load_w2 = [&](int t) {
            const int col = col0 + wc;
            return col < o2 ? *reinterpret_cast<const float4 *>(w2t + (long)(t * GP_KT + wk) * o2 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        floatx16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
        float4 w2v = load_w2(0);
        __syncthreads();
        *reinterpret_cast<float4 *>(w2s + wk * 64 + wc) = w2v;
        __syncthreads();
        for (int t = 0; t < nt2; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt2) w2v = load_w2(t + 1);
            const float *wl = w2s + cur * GP_KT * 64;
#pragma unroll
            for (int k = 0; k < GP_KT; k += 2)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(act[(t * GP_KT + k + kh) * GP_XS + ar], wl[(k + kh) * 64 + bc], acc2, 0, 0, 0);
            if (t + 1 < nt2) *reinterpret_cast<float4 *>(w2s + (cur ^ 1) * GP_KT * 64 + wk * 64 + wc) = w2v;
            __syncthreads();
        }
        const int col = col0 + bc;
        if (col < o2) {
            const float bv = b2 ? b2[col] : 0.f;
            const long rb = row0 + wm * 32 + 4 * kh;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const long t = rb + 8 * (v / 4) + (v % 4);
                float y = acc2[v] + bv;
                if (relu2) y = y < 0.f ? 0.f : y;
                if (t < T) out[t * (long)o2 + col] = y;
            }
        }
    }
}

template <int NB1>
__global__ __launch_bounds__(256) void pgather_gemm2_compact_kernel(int o2, int n, int m, const float *__restrict__ pmat, int p_stride,
                                                                    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                    const int32_t *__restrict__ rowc, const int32_t *__restrict__ rowsrc,
                                                                    const int32_t *__restrict__ total, const float *__restrict__ w1x,
                                                                    const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                                    const float *__restrict__ b2, int relu2, float *__restrict__ out, long limit) {
    pgather_gemm2_compact_body<NB1>(blockIdx.x, blockIdx.y, gridDim.y, o2, n, m, pmat, p_stride, xyz, new_xyz, rowc, rowsrc, total, w1x, b1, relu1, w2t, b2,
                                    relu2, out, limit);
}

// ---- the two scales of a set-abstraction level in ONE launch (round 5): the kernels of a level's scales differ in their arguments only
// (same first-layer width), neither fills the chip at batch 8, and a launch costs the 20-deep pipeline ~0.2 % of its throughput.
// Workgroups [0, split) along x run the first scale's body, the others the second's (x - split).  One argument block serves the three
// compact kernels: pgather_gemm3 (everything), pgather_gemm2 (rows out -> mid), gemm_pool (mid -> out).
struct CompactMlpArgs {
    const float *pmat, *xyz, *new_xyz;
    const int32_t *rowc, *rowsrc, *total;
    const float *w1x, *b1, *w2t, *b2, *w3t, *b3;
    float *mid, *out;
    long limit;
    int o2, o3, n, m, p_stride, relu1, relu2, out_stride, gy, pad;
};

template <int NB1>
__global__ __launch_bounds__(256) void pgather_gemm2_compact_pair_kernel(const CompactMlpArgs a0, const CompactMlpArgs a1, const unsigned split,
                                                                         const unsigned total_x) {
    // round 6: a workgroup walks the x tiles bx = blockIdx.x, + gridDim.x, .. (the grid is sized for the lists' capacity: three quarters of
    // its tiles lie behind *total on LiDAR clouds and return at once -- with ws3d_tune key 4 they are not launched at all)
    for (unsigned bx = blockIdx.x; bx < total_x; bx += gridDim.x) {
        const bool second = bx >= split;
        const CompactMlpArgs &a = second ? a1 : a0;
        if (blockIdx.y < (unsigned)a.gy)
            pgather_gemm2_compact_body<NB1>(second ? bx - split : bx, blockIdx.y, (unsigned)a.gy, a.o2, a.n, a.m, a.pmat, a.p_stride, a.xyz, a.new_xyz,
                                            a.rowc, a.rowsrc, a.total, a.w1x, a.b1, a.relu1, a.w2t, a.b2, a.relu2, a.mid, a.limit);
        __syncthreads();
    }
}

// last layer over compact rows + max over each centre's rows: out[centre, col] = max(out, relu(x W + b)) by integer atomic max
// (out starts at 0, every candidate is >= 0 after the ReLU: the float order is the integer order)
__device__ __forceinline__ void gemm_pool_compact_body(const unsigned bx, int k_dim, int o_dim, const float *__restrict__ x, const int32_t *__restrict__ rowc,
                                                       const int32_t *__restrict__ total, const float *__restrict__ wt,
                                                       const float *__restrict__ bias, float *__restrict__ out, int out_stride, long limit) {
    const long T = *total;
    const int col_tiles = o_dim / 64;
    const long row_tile = bx / col_tiles;
    const int col_tile = (int)(bx - row_tile * col_tiles);
    const long row0 = row_tile * 64;
    if (row0 >= T || (limit >= 0 && T > limit)) return;
    __shared__ float xs[2][GP_KT][GP_XS];
    __shared__ float ws[2][GP_KT][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int col0 = col_tile * 64;
    const int xr = tid >> 2, xk = (tid & 3) * 4;
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const float *xp = x + min(row0 + xr, T - 1) * (long)k_dim;
    auto load_x = [&](int k0) {
        const int k = k0 + xk;
        return k < k_dim ? *reinterpret_cast<const float4 *>(xp + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto load_w = [&](int k0) {
        const int k = k0 + wk;
        return k < k_dim ? *reinterpret_cast<const float4 *>(wt + (long)k * o_dim + col0 + wc) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf, const float4 xv, const float4 wv) {
        xs[buf][xk + 0][xr] = xv.x; xs[buf][xk + 1][xr] = xv.y; xs[buf][xk + 2][xr] = xv.z; xs[buf][xk + 3][xr] = xv.w;
        *reinterpret_cast<float4 *>(&ws[buf][wk][wc]) = wv;
    };
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 xv = load_x(0), wv = load_w(0);
    stage(0, xv, wv);
    __syncthreads();
    const int ntiles = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) { xv = load_x((t + 1) * GP_KT); wv = load_w((t + 1) * GP_KT); }
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[cur][k + kh][ar], ws[cur][k + kh][bc], acc, 0, 0, 0);
        if (t + 1 < ntiles) stage(cur ^ 1, xv, wv);
        __syncthreads();
    }
    const int col = col0 + bc;
    const float bv = bias ? bias[col] : 0.f;
    // consecutive rows mostly belong to one centre: each half pools 16 consecutive rows in registers, one atomic per centre (compact_pool.h)
    int cen[16];
    compact_centres16(rowc, row0 + wm * 32 + 16 * kh, T, cen);
    compact_pool_atomic(acc, bv, cen, out + col, out_stride);
}

__global__ __launch_bounds__(256) void gemm_pool_compact_kernel(int k_dim, int o_dim, const float *__restrict__ x, const int32_t *__restrict__ rowc,
                                                                const int32_t *__restrict__ total, const float *__restrict__ wt,
                                                                const float *__restrict__ bias, float *__restrict__ out, int out_stride, long limit) {
    gemm_pool_compact_body(blockIdx.x, k_dim, o_dim, x, rowc, total, wt, bias, out, out_stride, limit);
}

__global__ __launch_bounds__(256) void gemm_pool_compact_pair_kernel(const CompactMlpArgs a0, const CompactMlpArgs a1, const unsigned split, const unsigned total_x) {
    for (unsigned bx = blockIdx.x; bx < total_x; bx += gridDim.x) {        // (see pgather_gemm2_compact_pair_kernel)
        const bool second = bx >= split;
        const CompactMlpArgs &a = second ? a1 : a0;
        gemm_pool_compact_body(second ? bx - split : bx, a.o2, a.o3, a.mid, a.rowc, a.total, a.w3t, a.b3, a.out, a.out_stride, a.limit);
        __syncthreads();
    }
}

// ---- the WHOLE SharedMLP of a set-abstraction scale over compact rows in one kernel (round 4): pgather_gemm2_compact_kernel's
// layer 1 (gathered P row + xyz term) and layer 2, layer 2's 64 x O2 tile kept in LDS next to layer 1's (as gather_gemm2_kernel<.., POOL>
// does for dense rows) and gemm_pool_compact_kernel's last layer + atomic max multiplied straight out of it, 128 output columns per
// pass.  Against the two-kernel form: the (rows, O2) activation never reaches HBM (98 MB per batch of 8 at SA2, written and read
// back), the rows of a tile are gathered once instead of once per 64-column pass of layer 2, one launch per scale instead of two.
// The k order of every dot product is the two kernels' (ascending k, two per matrix instruction, zero padding behind o2): the
// pooled rows are bit-identical to theirs.  LDS: (O1 + O2 rounded up to 16) x 65 + 2 x 16 x 64 floats -- 41 / 50 KB at SA2.
template <int NB1>
__device__ __forceinline__ void pgather_gemm3_compact_body(const unsigned bx, int o2, int o3, int n, int m, const float *__restrict__ pmat, int p_stride,
                                                           const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                           const int32_t *__restrict__ rowc, const int32_t *__restrict__ rowsrc,
                                                           const int32_t *__restrict__ total, const float *__restrict__ w1x,
                                                           const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                           const float *__restrict__ b2, int relu2, const float *__restrict__ w3t,
                                                           const float *__restrict__ b3, float *__restrict__ out, int out_stride, long limit) {
    constexpr int O1 = NB1 * 64;
    const long T = *total;
    const long row0 = (long)bx * 64;
    if (row0 >= T || (limit >= 0 && T > limit)) return;      // workgroup-uniform; beyond the limit the dense kernels run instead
    extern __shared__ __attribute__((aligned(16))) float smem2[];
    const int o2p = (o2 + GP_KT - 1) / GP_KT * GP_KT;
    float *act = smem2, *act2 = smem2 + O1 * GP_XS, *w2s = smem2 + (O1 + o2p) * GP_XS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    float a0, a1;
    {
        const long t = min(row0 + ar, T - 1);                // rows behind the end repeat the last one (never reduced)
        const long cm = rowc[t];
        const long scene = cm / m;
        const int src = rowsrc[t];
        const float *pr = xyz + ((size_t)scene * n + (size_t)src) * 3, *cr = new_xyz + (size_t)cm * 3;
        const float dx = pr[0] - cr[0], dy = pr[1] - cr[1], dz = pr[2] - cr[2];
        a0 = kh ? dy : dx;
        a1 = kh ? 0.f : dz;
    }
    int cidx[16];                                            // the centres of the 16 consecutive rows this half pools (compact_pool.h)
    compact_centres16(rowc, row0 + wm * 32 + 16 * kh, T, cidx);
    floatx16 acc1[NB1];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const long tr = row0 + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4);
        const long t = min(tr, T - 1);
        const int cm = rowc[t];
        const long scene = cm / m;
        const float *prow = pmat + ((size_t)scene * n + (size_t)rowsrc[t]) * p_stride + bc;
#pragma unroll
        for (int j = 0; j < NB1; ++j) acc1[j][v] = prow[j * 64];
    }
#pragma unroll
    for (int j = 0; j < NB1; ++j) {
        const int col = j * 64 + bc;
        const float wb0 = w1x[kh * O1 + col];
        const float wb1 = kh ? 0.f : w1x[2 * O1 + col];
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wb0, acc1[j], 0, 0, 0);
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wb1, acc1[j], 0, 0, 0);
        const float bv = b1 ? b1[col] : 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            float y = acc1[j][v] + bv;
            if (relu1) y = y < 0.f ? 0.f : y;
            act[col * GP_XS + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4)] = y;
        }
    }
    // layer 2 into act2, 64 output columns per pass
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const int nchunk = (o2 + 63) / 64;
    constexpr int nt2 = O1 / GP_KT;
    for (int c = 0; c < nchunk; ++c) {
        const int col0 = c * 64;
        auto load_w2 = [&](int t) {
            const int col = col0 + wc;
            return col < o2 ? *reinterpret_cast<const float4 *>(w2t + (long)(t * GP_KT + wk) * o2 + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        floatx16 acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
        float4 w2v = load_w2(0);
        __syncthreads();                    // the activation tile is complete / the previous pass has left w2s
        *reinterpret_cast<float4 *>(w2s + wk * 64 + wc) = w2v;
        __syncthreads();
        for (int t = 0; t < nt2; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt2) w2v = load_w2(t + 1);
            const float *wl = w2s + cur * GP_KT * 64;
#pragma unroll
            for (int k = 0; k < GP_KT; k += 2)
                acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(act[(t * GP_KT + k + kh) * GP_XS + ar], wl[(k + kh) * 64 + bc], acc2, 0, 0, 0);
            if (t + 1 < nt2) *reinterpret_cast<float4 *>(w2s + (cur ^ 1) * GP_KT * 64 + wk * 64 + wc) = w2v;
            __syncthreads();
        }
        const int col = col0 + bc;
        if (col < o2p) {                    // columns o2 .. o2p - 1: the zero padding of layer 3's k dimension
            const float bv = (col < o2 && b2) ? b2[col] : 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float y = acc2[v] + bv;
                if (relu2) y = y < 0.f ? 0.f : y;
                act2[col * GP_XS + wm * 32 + 8 * (v / 4) + 4 * kh + (v % 4)] = col < o2 ? y : 0.f;
            }
        }
    }
    // layer 3 out of act2, 128 output columns per pass (o3 % 128 == 0), W3 tiles ([2][GP_KT][128]) in layer 1's tile, which is dead by now
    const int nt3 = o2p / GP_KT;
    const int wk3 = tid >> 5, wc3 = (tid & 31) * 4;
    float *w3s = smem2;
    for (int c = 0; c < o3 / 128; ++c) {
        const int col0 = c * 128;
        float4 w3v[2];
        auto load_w3 = [&](int t) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = t * GP_KT + wk3 + 8 * q;
                w3v[q] = k < o2 ? *reinterpret_cast<const float4 *>(w3t + (long)k * o3 + col0 + wc3) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        auto stage_w3 = [&](int buf) {
#pragma unroll
            for (int q = 0; q < 2; ++q) *reinterpret_cast<float4 *>(w3s + buf * GP_KT * 128 + (wk3 + 8 * q) * 128 + wc3) = w3v[q];
        };
        floatx16 acc3[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc3[q][i] = 0.f;
        load_w3(0);
        __syncthreads();                    // act2 is complete, act is dead / the previous pass has left w3s
        stage_w3(0);
        __syncthreads();
        for (int t = 0; t < nt3; ++t) {
            const int cur = t & 1;
            if (t + 1 < nt3) load_w3(t + 1);
            const float *wl = w3s + cur * GP_KT * 128;
#pragma unroll
            for (int k = 0; k < GP_KT; k += 2) {
                const float a = act2[(t * GP_KT + k + kh) * GP_XS + ar];
#pragma unroll
                for (int q = 0; q < 2; ++q) acc3[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wl[(k + kh) * 128 + q * 64 + bc], acc3[q], 0, 0, 0);
            }
            if (t + 1 < nt3) stage_w3(cur ^ 1);
            __syncthreads();
        }
        // bias + ReLU, each half pools its 16 consecutive rows, one integer atomic max per centre (compact_pool.h)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int col = col0 + q * 64 + bc;
            compact_pool_atomic(acc3[q], b3 ? b3[col] : 0.f, cidx, out + col, out_stride);
        }
    }
}

template <int NB1>
__global__ __launch_bounds__(256) void pgather_gemm3_compact_kernel(int o2, int o3, int n, int m, const float *__restrict__ pmat, int p_stride,
                                                                    const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                    const int32_t *__restrict__ rowc, const int32_t *__restrict__ rowsrc,
                                                                    const int32_t *__restrict__ total, const float *__restrict__ w1x,
                                                                    const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                                    const float *__restrict__ b2, int relu2, const float *__restrict__ w3t,
                                                                    const float *__restrict__ b3, float *__restrict__ out, int out_stride, long limit) {
    pgather_gemm3_compact_body<NB1>(blockIdx.x, o2, o3, n, m, pmat, p_stride, xyz, new_xyz, rowc, rowsrc, total, w1x, b1, relu1, w2t, b2, relu2, w3t, b3, out,
                                    out_stride, limit);
}

template <int NB1>
__global__ __launch_bounds__(256) void pgather_gemm3_compact_pair_kernel(const CompactMlpArgs a0, const CompactMlpArgs a1, const unsigned split, const unsigned total_x) {
    const bool second = blockIdx.x >= split;
    const CompactMlpArgs &a = second ? a1 : a0;
    pgather_gemm3_compact_body<NB1>(second ? blockIdx.x - split : blockIdx.x, a.o2, a.o3, a.n, a.m, a.pmat, a.p_stride, a.xyz, a.new_xyz, a.rowc, a.rowsrc, a.total,
                                    a.w1x, a.b1, a.relu1, a.w2t, a.b2, a.relu2, a.w3t, a.b3, a.out, a.out_stride, a.limit);
}

// ---- first layer of a feature-propagation module WITHOUT its per-point product over the interpolated channels.  Interpolation
// is linear, so  W_a (w0 f[i0] + w1 f[i1] + w2 f[i2]) = w0 (W_a f)[i0] + w1 (W_a f)[i1] + w2 (W_a f)[i2]:  Q = known_feats @ W_a
// is one product over the m KNOWN points (a quarter of the unknown ones), and the layer is
//   out[r, :] = relu?( w0 Q[i0, :] + w1 Q[i1, :] + w2 Q[i2, :] + lin[r, :] ),   lin = skip @ W_b + bias
// where lin comes from a GEMM over the skip channels only (lin != NULL), or -- c1 <= 4 skip channels (FP1: one) -- is
// evaluated here as fmaf chains over skip[r, 0:c1] and wb (c1, o).  The interpolation uses three_interpolate's expression
// (interpolate_gpu.cu:77-97) on Q's rows.  One float4 of a row per thread.
__global__ __launch_bounds__(256) void qinterp_rows_kernel(long rows, int o, int n, int m, const float *__restrict__ q, const int32_t *__restrict__ idx3,
                                                           const float *__restrict__ w3, const float *__restrict__ lin,
                                                           const float *__restrict__ skip, int c1, const float *__restrict__ wb,
                                                           const float *__restrict__ bias, int relu, float *__restrict__ out) {
    const int q4 = o >> 2;
    const long total = rows * q4;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
        const long r = e / q4;
        const int c4 = (int)(e - r * q4) * 4;
        const long scene = r / n;
        const int i0 = idx3[r * 3 + 0], i1 = idx3[r * 3 + 1], i2 = idx3[r * 3 + 2];
        const float w0 = w3[r * 3 + 0], w1 = w3[r * 3 + 1], w2 = w3[r * 3 + 2];
        const float *qb = q + (size_t)scene * m * o + c4;
        const float4 p0 = *reinterpret_cast<const float4 *>(qb + (size_t)i0 * o), p1 = *reinterpret_cast<const float4 *>(qb + (size_t)i1 * o),
                     p2 = *reinterpret_cast<const float4 *>(qb + (size_t)i2 * o);
        float4 y = make_float4(__builtin_fmaf(w2, p2.x, __builtin_fmaf(w0, p0.x, w1 * p1.x)), __builtin_fmaf(w2, p2.y, __builtin_fmaf(w0, p0.y, w1 * p1.y)),
                               __builtin_fmaf(w2, p2.z, __builtin_fmaf(w0, p0.z, w1 * p1.z)), __builtin_fmaf(w2, p2.w, __builtin_fmaf(w0, p0.w, w1 * p1.w)));
        if (lin) {
            const float4 l4 = *reinterpret_cast<const float4 *>(lin + r * (long)o + c4);
            y.x += l4.x; y.y += l4.y; y.z += l4.z; y.w += l4.w;
        } else {
            for (int k = 0; k < c1; ++k) {
                const float sv = skip[r * (long)c1 + k];
                const float4 wv = *reinterpret_cast<const float4 *>(wb + (long)k * o + c4);
                y.x = __builtin_fmaf(sv, wv.x, y.x); y.y = __builtin_fmaf(sv, wv.y, y.y); y.z = __builtin_fmaf(sv, wv.z, y.z); y.w = __builtin_fmaf(sv, wv.w, y.w);
            }
            if (bias) {
                const float4 bv = *reinterpret_cast<const float4 *>(bias + c4);
                y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
            }
        }
        if (relu) { y.x = y.x < 0.f ? 0.f : y.x; y.y = y.y < 0.f ? 0.f : y.y; y.z = y.z < 0.f ? 0.f : y.z; y.w = y.w < 0.f ? 0.f : y.w; }
        *reinterpret_cast<float4 *>(out + r * (long)o + c4) = y;
    }
}

// ---- first layer of a feature-propagation module with the interpolation and the concatenation fused into its A operand:
//   out[r, o] = relu?( sum_k X[r, k] Wt[k, o] + bias[o] ),   r = (scene b, unknown point p),
//   X[r, 0:c2]      = w0 f[i0, :] + w1 f[i1, :] + w2 f[i2, :]   (three_interpolate, interpolate_gpu.cu:77-97, the same fmaf
//                     expression as the stand-alone kernels)     f = known_feats (b, m, c2) channels-last
//   X[r, c2:c2+c1]  = unknown_feats[b, p, :]                     (the skip connection, pointnet2_modules.py:147-150)
// so neither the interpolated tensor nor the (rows, c2 + c1) concat buffer exists (135 MB per batch at FP1).
__global__ __launch_bounds__(256) void interp_gemm_kernel(int c2, int c1, int o_dim, int n, int m, const float *__restrict__ known_feats,
                                                          const float *__restrict__ unknown_feats, const int32_t *__restrict__ idx3,
                                                          const float *__restrict__ w3, const float *__restrict__ wt,
                                                          const float *__restrict__ bias, int relu, float *__restrict__ out, int tps) {
    __shared__ float xs[2][GP_KT][GP_XS];     // [k][row]
    __shared__ float ws[2][GP_KT][64];        // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    long row_tile;
    int col_tile;
    gg_tile(o_dim / 64, tps, row_tile, col_tile);
    const long row0 = row_tile * 64;
    const int col0 = col_tile * 64;
    const int k_dim = c2 + c1;
    const int xr = tid >> 2, xk = (tid & 3) * 4;
    const int wk = tid >> 4, wc = (tid & 15) * 4;
    const long r = row0 + xr;
    const long b = r / n;
    const int i0 = idx3[r * 3 + 0], i1 = idx3[r * 3 + 1], i2 = idx3[r * 3 + 2];
    const float w0 = w3[r * 3 + 0], w1 = w3[r * 3 + 1], w2 = w3[r * 3 + 2];
    const float *f0 = known_feats + ((size_t)b * m + (size_t)i0) * c2, *f1 = known_feats + ((size_t)b * m + (size_t)i1) * c2,
                *f2 = known_feats + ((size_t)b * m + (size_t)i2) * c2;
    const float *urow = unknown_feats ? unknown_feats + (size_t)r * c1 : nullptr;
    auto load_x = [&](int k0) {
        const int k = k0 + xk;
        if (k < c2) {
            const float4 p0 = *reinterpret_cast<const float4 *>(f0 + k), p1 = *reinterpret_cast<const float4 *>(f1 + k),
                         p2 = *reinterpret_cast<const float4 *>(f2 + k);
            return make_float4(__builtin_fmaf(w2, p2.x, __builtin_fmaf(w0, p0.x, w1 * p1.x)), __builtin_fmaf(w2, p2.y, __builtin_fmaf(w0, p0.y, w1 * p1.y)),
                               __builtin_fmaf(w2, p2.z, __builtin_fmaf(w0, p0.z, w1 * p1.z)), __builtin_fmaf(w2, p2.w, __builtin_fmaf(w0, p0.w, w1 * p1.w)));
        }
        const int ku = k - c2;
        if (ku + 3 < c1) return *reinterpret_cast<const float4 *>(urow + ku);          // c1 % 4 == 0: aligned
        float v[4] = {0.f, 0.f, 0.f, 0.f};                                            // the ragged tail (c1 = 1 at FP1)
#pragma unroll
        for (int q = 0; q < 4; ++q) if (ku + q < c1) v[q] = urow[ku + q];
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto load_w = [&](int k0) {
        const int k = k0 + wk;
        return k < k_dim ? *reinterpret_cast<const float4 *>(wt + (long)k * o_dim + col0 + wc) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage = [&](int buf, const float4 xv, const float4 wv) {
        xs[buf][xk + 0][xr] = xv.x; xs[buf][xk + 1][xr] = xv.y; xs[buf][xk + 2][xr] = xv.z; xs[buf][xk + 3][xr] = xv.w;
        *reinterpret_cast<float4 *>(&ws[buf][wk][wc]) = wv;
    };
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float4 xv = load_x(0), wv = load_w(0);
    stage(0, xv, wv);
    __syncthreads();
    const int ntiles = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 + (lane & 31), bc = wn * 32 + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) { xv = load_x((t + 1) * GP_KT); wv = load_w((t + 1) * GP_KT); }
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xs[cur][k + kh][ar], ws[cur][k + kh][bc], acc, 0, 0, 0);
        if (t + 1 < ntiles) stage(cur ^ 1, xv, wv);
        __syncthreads();
    }
    const int col = col0 + bc;
    const float bv = bias ? bias[col] : 0.f;
    float *o = out + (row0 + wm * 32 + 4 * (lane >> 5)) * (long)o_dim + col;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        float y = acc[v] + bv;
        if (relu) y = y < 0.f ? 0.f : y;
        o[(long)(8 * (v / 4) + (v % 4)) * o_dim] = y;
    }
}

// interp_gemm_kernel with a (64 MB) x (64 NB) output tile (see gemm_pool_big_kernel): besides the lighter L2 / LDS traffic the
// interpolated A tile -- three gathered rows and three fmaf per element -- is built once per 64 NB output columns instead of
// once per 64 (at FP1, 128 output channels: exactly once).
template <int MB, int NB, bool PRE = false>
__global__ __launch_bounds__(256) void interp_gemm_big_kernel(int c2, int c1, int o_dim, int n, int m, const float *__restrict__ known_feats,
                                                              const float *__restrict__ unknown_feats, const int32_t *__restrict__ idx3,
                                                              const float *__restrict__ w3, const float *__restrict__ wt,
                                                              const float *__restrict__ bias, int relu, float *__restrict__ out, int tps,
                                                              const float *__restrict__ lin = nullptr, const float *__restrict__ wb = nullptr,
                                                              const float *__restrict__ b1 = nullptr, int relu_a = 0, long total_tiles = 0) {
    // PRE (ws3d_qinterp_gemm): the A operand is the FIRST layer of the module, built on the fly as ws3d_qinterp_rows builds it --
    //   x[r, k] = relu_a?( w0 Q[i0, k] + w1 Q[i1, k] + w2 Q[i2, k] + (lin[r, k]  |  sum_j skip[r, j] wb[j, k] + b1[k]) )
    // with known_feats = Q (c2 = the layer's width), unknown_feats = the c1 <= 4 skip channels of the second form -- and this kernel's
    // product is the SECOND layer: K = c2, no concatenated columns.
    constexpr int TM = 64 * MB, TN = 64 * NB, XS = TM + 1;
    __shared__ float xs[2][GP_KT][XS];        // [k][row]
    __shared__ float ws[2][GP_KT][TN];        // [k][col]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    // round 6: a workgroup walks tiles vb = blockIdx.x, + gridDim.x, .. (total_tiles = 0: one tile per workgroup, the launch of rounds
    // 2-5); gridDim.x is a multiple of 8 then, so a workgroup's tiles stay on its XCD's scenes (gg_tile)
    const long n_tiles = total_tiles > 0 ? total_tiles : (long)gridDim.x;
    for (long vb = blockIdx.x; vb < n_tiles; vb += gridDim.x) {
    long row_tile;
    int col_tile;
    gg_tile(o_dim / TN, tps, row_tile, col_tile, vb);
    const long row0 = row_tile * TM;
    const int col0 = col_tile * TN;
    const int k_dim = PRE ? c2 : c2 + c1;
    const int xk = (tid & 3) * 4;
    const float *f0[MB], *f1[MB], *f2[MB], *urow[MB];
    const float *lrow[MB];
    float w0[MB], w1[MB], w2[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const long r = row0 + (tid >> 2) + 64 * i;
        const long b = r / n;
        const float *base = known_feats + (size_t)b * m * c2;
        f0[i] = base + (size_t)idx3[r * 3 + 0] * c2; f1[i] = base + (size_t)idx3[r * 3 + 1] * c2; f2[i] = base + (size_t)idx3[r * 3 + 2] * c2;
        w0[i] = w3[r * 3 + 0]; w1[i] = w3[r * 3 + 1]; w2[i] = w3[r * 3 + 2];
        urow[i] = unknown_feats ? unknown_feats + (size_t)r * c1 : nullptr;
        lrow[i] = (PRE && lin) ? lin + (size_t)r * c2 : nullptr;
    }
    float4 xv[MB], wv[NB];
    auto load = [&](int k0) {
        const int k = k0 + xk;
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            if (k < c2) {
                const float4 p0 = *reinterpret_cast<const float4 *>(f0[i] + k), p1 = *reinterpret_cast<const float4 *>(f1[i] + k),
                             p2 = *reinterpret_cast<const float4 *>(f2[i] + k);
                xv[i] = make_float4(__builtin_fmaf(w2[i], p2.x, __builtin_fmaf(w0[i], p0.x, w1[i] * p1.x)),
                                    __builtin_fmaf(w2[i], p2.y, __builtin_fmaf(w0[i], p0.y, w1[i] * p1.y)),
                                    __builtin_fmaf(w2[i], p2.z, __builtin_fmaf(w0[i], p0.z, w1[i] * p1.z)),
                                    __builtin_fmaf(w2[i], p2.w, __builtin_fmaf(w0[i], p0.w, w1[i] * p1.w)));
                if (PRE) {
                    float4 y = xv[i];
                    if (lrow[i]) {
                        const float4 l4 = *reinterpret_cast<const float4 *>(lrow[i] + k);
                        y.x += l4.x; y.y += l4.y; y.z += l4.z; y.w += l4.w;
                    } else {
                        for (int j = 0; j < c1; ++j) {
                            const float sv = urow[i][j];
                            const float4 wv4 = *reinterpret_cast<const float4 *>(wb + (long)j * c2 + k);
                            y.x = __builtin_fmaf(sv, wv4.x, y.x); y.y = __builtin_fmaf(sv, wv4.y, y.y); y.z = __builtin_fmaf(sv, wv4.z, y.z); y.w = __builtin_fmaf(sv, wv4.w, y.w);
                        }
                        if (b1) {
                            const float4 bv = *reinterpret_cast<const float4 *>(b1 + k);
                            y.x += bv.x; y.y += bv.y; y.z += bv.z; y.w += bv.w;
                        }
                    }
                    if (relu_a) { y.x = y.x < 0.f ? 0.f : y.x; y.y = y.y < 0.f ? 0.f : y.y; y.z = y.z < 0.f ? 0.f : y.z; y.w = y.w < 0.f ? 0.f : y.w; }
                    xv[i] = y;
                }
            } else if (PRE) {
                xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int ku = k - c2;
                if (ku + 3 < c1) {
                    xv[i] = *reinterpret_cast<const float4 *>(urow[i] + ku);            // c1 % 4 == 0: aligned
                } else {
                    float v[4] = {0.f, 0.f, 0.f, 0.f};                                  // the ragged tail (c1 = 1 at FP1)
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (ku + q < c1) v[q] = urow[i][ku + q];
                    xv[i] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j, kk = k0 + idx / (16 * NB), c = (idx % (16 * NB)) * 4;
            wv[j] = kk < k_dim ? *reinterpret_cast<const float4 *>(wt + (long)kk * o_dim + col0 + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int r = (tid >> 2) + 64 * i;
            xs[buf][xk + 0][r] = xv[i].x; xs[buf][xk + 1][r] = xv[i].y; xs[buf][xk + 2][r] = xv[i].z; xs[buf][xk + 3][r] = xv[i].w;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int idx = tid + 256 * j;
            *reinterpret_cast<float4 *>(&ws[buf][idx / (16 * NB)][(idx % (16 * NB)) * 4]) = wv[j];
        }
    };
    floatx16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    load(0);
    stage(0);
    __syncthreads();
    const int ntiles = (k_dim + GP_KT - 1) / GP_KT;
    const int ar = wm * 32 * MB + (lane & 31), bc = wn * 32 * NB + (lane & 31), kh = lane >> 5;
    for (int t = 0; t < ntiles; ++t) {
        const int cur = t & 1;
        if (t + 1 < ntiles) load((t + 1) * GP_KT);
#pragma unroll
        for (int k = 0; k < GP_KT; k += 2) {
            float a[MB], bq[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) a[i] = xs[cur][k + kh][ar + 32 * i];
#pragma unroll
            for (int j = 0; j < NB; ++j) bq[j] = ws[cur][k + kh][bc + 32 * j];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < ntiles) stage(cur ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = col0 + bc + 32 * j;
            const float bv = bias ? bias[col] : 0.f;
            float *o = out + (row0 + wm * 32 * MB + 32 * i + 4 * (lane >> 5)) * (long)o_dim + col;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float y = acc[i][j][v] + bv;
                if (relu) y = y < 0.f ? 0.f : y;
                o[(long)(8 * (v / 4) + (v % 4)) * o_dim] = y;
            }
        }
    }
}

}  // namespace ws3d

extern "C" int ws3d_gemm_pool(long rows, int nsample, int k_dim, int o_dim, const float *x_rows, const float *wt,
                              const float *bias, int relu, float *out, int out_stride, const int32_t *gate, long gate_limit, ws3d_stream_t stream) {
    using namespace ws3d;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x_rows) | reinterpret_cast<uintptr_t>(wt);
    if (rows < 0 || (nsample != 16 && nsample != 32) || k_dim <= 0 || (k_dim & 3) || o_dim <= 0 || (o_dim & 63) || (rows & 63) ||
        !x_rows || !wt || !out || out_stride < o_dim || (al & 15)) {
        set_error("ws3d_gemm_pool: unsupported shape (rows=%ld ns=%d k=%d o=%d; rows, o %% 64, k %% 4, ns 16|32)", rows, nsample,
                  k_dim, o_dim);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    constexpr int xcd_env = 1;        // XCD-aware tile order (the plain order was an A/B switch until round 4)
    // tile: 128 x 128 when that still gives two workgroups per CU (same kernel time as 64 x 64 -- the matrix pipe is the bound at
    // either size -- at half the L2 and LDS traffic: +1.3 % in the 20-deep pipeline), else 64 x 64 (128 x 64 and 64 x 128 were
    // measured too: profiles/r02_gemm_pool_tiles.txt)
    const int tile_env = (rows % 128 == 0 && o_dim % 128 == 0 && (rows / 128) * (o_dim / 128) >= 512) ? 22 : 11;
    {
        const int mb = tile_env / 10, nb = tile_env % 10;
        if (tile_env != 11) {
            const long rt = rows / (64 * mb);
            const int xc = (xcd_env && rt % 8 == 0) ? 1 : 0;
            const dim3 grid((unsigned)((o_dim / (64 * nb)) * rt)), block(256);
#define GP_BIG(M_, N_)                                                                                                              \
    if (mb == M_ && nb == N_) {                                                                                                     \
        if (nsample == 16)                                                                                                          \
            hipLaunchKernelGGL((gemm_pool_big_kernel<16, M_, N_>), grid, block, 0, as_stream(stream), k_dim, o_dim, x_rows, wt, bias, relu, out, out_stride, xc, gate, gate_limit); \
        else                                                                                                                        \
            hipLaunchKernelGGL((gemm_pool_big_kernel<32, M_, N_>), grid, block, 0, as_stream(stream), k_dim, o_dim, x_rows, wt, bias, relu, out, out_stride, xc, gate, gate_limit); \
        return check_launch("ws3d_gemm_pool");                                                                                      \
    }
            GP_BIG(2, 2)
#undef GP_BIG
        }
    }
    const long row_tiles = rows / 64;
    const int xcd = (xcd_env && row_tiles % 8 == 0) ? 1 : 0;
    const dim3 grid((unsigned)((o_dim / 64) * row_tiles)), block(256);
    if (nsample == 16)
        hipLaunchKernelGGL(gemm_pool_kernel<16>, grid, block, 0, as_stream(stream), k_dim, o_dim, x_rows, wt, bias, relu, out, out_stride, xcd, gate, gate_limit);
    else
        hipLaunchKernelGGL(gemm_pool_kernel<32>, grid, block, 0, as_stream(stream), k_dim, o_dim, x_rows, wt, bias, relu, out, out_stride, xcd, gate, gate_limit);
    return check_launch("ws3d_gemm_pool");
}

extern "C" int ws3d_gather_gemm(int b, int n, int m, int nsample, int c_feat, int o_dim, const float *feats, const float *xyz,
                                const float *new_xyz, const int32_t *nbr, const float *wt, const float *bias, int relu, float *out,
                                ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * m * nsample;
    const uintptr_t al = reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(wt);
    if (b < 0 || n <= 0 || m <= 0 || nsample <= 0 || c_feat <= 0 || (c_feat & 3) || o_dim <= 0 || (o_dim & 63) || (rows & 63) || !feats || !xyz ||
        !new_xyz || !nbr || !wt || !out || (al & 15)) {
        set_error("ws3d_gather_gemm: unsupported shape (b=%d n=%d m=%d ns=%d c=%d o=%d; rows, o %% 64, c %% 4)", b, n, m, nsample, c_feat, o_dim);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    constexpr int xcd_env = 1;
    const long per_scene = (long)m * nsample;
    const int tps = (xcd_env && (b & 7) == 0 && per_scene % 64 == 0) ? (int)(per_scene / 64) : 0;
    hipLaunchKernelGGL(gather_gemm_kernel, dim3((unsigned)((o_dim / 64) * (rows / 64))), dim3(256), 0, as_stream(stream), c_feat, o_dim, n, m,
                       nsample, feats, xyz, new_xyz, nbr, wt, bias, relu, out, tps);
    return check_launch("ws3d_gather_gemm");
}

extern "C" int ws3d_interp_gemm(int b, int n, int m, int c2, int c1, int o_dim, const float *known_feats, const float *unknown_feats,
                                const int32_t *idx, const float *weight, const float *wt, const float *bias, int relu, float *out,
                                ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * n;
    const uintptr_t al = reinterpret_cast<uintptr_t>(known_feats) | reinterpret_cast<uintptr_t>(wt) |
                         ((c1 & 3) == 0 ? reinterpret_cast<uintptr_t>(unknown_feats) : 0);
    if (b < 0 || n <= 0 || m <= 0 || c2 <= 0 || (c2 & 3) || c1 < 0 || o_dim <= 0 || (o_dim & 63) || (rows & 63) || !known_feats ||
        (c1 > 0 && !unknown_feats) || !idx || !weight || !wt || !out || (al & 15)) {
        set_error("ws3d_interp_gemm: unsupported shape (b=%d n=%d m=%d c2=%d c1=%d o=%d; rows, o %% 64, c2 %% 4)", b, n, m, c2, c1, o_dim);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    constexpr int xcd_env = 1;
    // 64 x 128 output tiles where they leave two workgroups per CU (FP1..FP3 of the c3 network): the interpolated A tile is built
    // half as often; measured 120 / 119 / 81 us against 128 / 131 / 82 at 64 x 64 and 119 / 123 / 90 at 128 x 128
    // (profiles/r02_interp_gemm_tiles.txt)
    const int tile = (o_dim % 128 == 0 && (rows / 64) * (o_dim / 128) >= 512) ? 12 : 11;
    const int mb = tile / 10, nb = tile % 10;
    if (tile != 11 && rows % (64 * mb) == 0 && o_dim % (64 * nb) == 0) {
        const int tpsb = (xcd_env && (b & 7) == 0 && n % (64 * mb) == 0) ? n / (64 * mb) : 0;
        const dim3 grid((unsigned)((o_dim / (64 * nb)) * (rows / (64 * mb))));
#define IG_BIG(M_, N_)                                                                                                                  \
    if (mb == M_ && nb == N_) {                                                                                                         \
        hipLaunchKernelGGL((interp_gemm_big_kernel<M_, N_>), grid, dim3(256), 0, as_stream(stream), c2, c1, o_dim, n, m, known_feats,   \
                           unknown_feats, idx, weight, wt, bias, relu, out, tpsb);                                                      \
        return check_launch("ws3d_interp_gemm");                                                                                        \
    }
        IG_BIG(1, 2)
#undef IG_BIG
    }
    const int tps = (xcd_env && (b & 7) == 0 && n % 64 == 0) ? n / 64 : 0;
    hipLaunchKernelGGL(interp_gemm_kernel, dim3((unsigned)((o_dim / 64) * (rows / 64))), dim3(256), 0, as_stream(stream), c2, c1, o_dim, n, m,
                       known_feats, unknown_feats, idx, weight, wt, bias, relu, out, tps);
    return check_launch("ws3d_interp_gemm");
}

extern "C" int ws3d_qinterp_gemm(int b, int n, int m, int c, int o_dim, const float *q, const int32_t *idx, const float *weight, const float *lin,
                                const float *skip, int c1, const float *wb, const float *b1, int relu1, const float *w2t, const float *b2, int relu2,
                                float *out, ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * n;
    const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(w2t) | reinterpret_cast<uintptr_t>(lin) | reinterpret_cast<uintptr_t>(wb) |
                         reinterpret_cast<uintptr_t>(b1);
    if (b < 0 || n <= 0 || m <= 0 || c <= 0 || (c & 15) || o_dim <= 0 || (o_dim & 127) || (rows & 63) || !q || !idx || !weight || !w2t || !out || (al & 15) ||
        (!lin && (c1 < 0 || c1 > 4 || (c1 > 0 && (!skip || !wb))))) {
        set_error("ws3d_qinterp_gemm: unsupported shape (b=%d n=%d m=%d c=%d o=%d c1=%d; rows %% 64, c %% 16, o %% 128, lin or c1 <= 4)", b, n, m, c, o_dim, c1);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    const int tpsb = ((b & 7) == 0 && n % 64 == 0) ? n / 64 : 0;
    const long tiles = (long)(o_dim / 128) * (rows / 64);
    // ws3d_tune key 3: at most that many workgroups (rounded to a multiple of 8), each walking its share of the tiles
    long wgs = g_tune[TUNE_FP_WGS] > 0 ? std::min((long)g_tune[TUNE_FP_WGS], tiles) : tiles;
    if (wgs < tiles) wgs = std::max(8L, wgs / 8 * 8);
    const dim3 grid((unsigned)wgs);
    hipLaunchKernelGGL((interp_gemm_big_kernel<1, 2, true>), grid, dim3(256), 0, as_stream(stream), c, lin ? 0 : c1, o_dim, n, m, q, lin ? nullptr : skip, idx,
                       weight, w2t, b2, relu2, out, tpsb, lin, wb, b1, relu1, wgs < tiles ? tiles : 0L);
    return check_launch("ws3d_qinterp_gemm");
}

extern "C" int ws3d_gather_gemm2(int b, int n, int m, int nsample, int c_feat, int o1, int o2, const float *feats, const float *xyz,
                                 const float *new_xyz, const int32_t *nbr, const float *w1t, const float *b1, int relu1,
                                 const float *w2t, const float *b2, int relu2, float *out, ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * m * nsample;
    const uintptr_t al = reinterpret_cast<uintptr_t>(feats) | reinterpret_cast<uintptr_t>(w1t) | reinterpret_cast<uintptr_t>(w2t);
    if (b < 0 || n <= 0 || m <= 0 || nsample <= 0 || c_feat <= 0 || (c_feat & 3) || (o1 != 64 && o1 != 128 && o1 != 256) || o2 <= 0 || (o2 & 3) ||
        (rows & 63) || !feats || !xyz || !new_xyz || !nbr || !w1t || !w2t || !out || (al & 15)) {
        set_error("ws3d_gather_gemm2: unsupported shape (b=%d n=%d m=%d ns=%d c=%d o1=%d o2=%d; rows %% 64, c %% 4, o1 in {64,128,256}, o2 %% 4)",
                  b, n, m, nsample, c_feat, o1, o2);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    const size_t p1 = (size_t)2 * GP_KT * GP_XS + (size_t)2 * GP_KT * o1, p2 = (size_t)o1 * GP_XS + (size_t)2 * GP_KT * 64;
    const size_t lds = sizeof(float) * (p1 > p2 ? p1 : p2);
#define WS3D_GG2(NB)                                                                                                                   \
    {                                                                                                                                  \
        if (int rc = raise_lds_cap((const void *)gather_gemm2_kernel<NB>, lds, "ws3d_gather_gemm2")) return rc;                         \
        hipLaunchKernelGGL((gather_gemm2_kernel<NB>), dim3((unsigned)(rows / 64)), dim3(256), lds, as_stream(stream), c_feat, o2, n, m, nsample, \
                           feats, xyz, new_xyz, nbr, w1t, b1, relu1, w2t, b2, relu2, out);                                             \
    }
    if (o1 == 64) WS3D_GG2(1) else if (o1 == 128) WS3D_GG2(2) else WS3D_GG2(4)
#undef WS3D_GG2
    return check_launch("ws3d_gather_gemm2");
}

extern "C" int ws3d_pgather_gemm2(int b, int n, int m, int nsample, int o1, int o2, const float *pmat, int p_stride, const float *xyz,
                                  const float *new_xyz, const int32_t *nbr, const float *w1x, const float *b1, int relu1,
                                  const float *w2t, const float *b2, int relu2, float *out, const int32_t *gate, long gate_limit, ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * m * nsample;
    if (b < 0 || n <= 0 || m <= 0 || nsample <= 0 || (o1 != 64 && o1 != 128) || o2 <= 0 || (o2 & 3) || ((long)m * nsample) % 64 || p_stride < o1 ||
        !pmat || !xyz || !new_xyz || !nbr || !w1x || !w2t || !out || (reinterpret_cast<uintptr_t>(w2t) & 15)) {
        set_error("ws3d_pgather_gemm2: unsupported shape (b=%d n=%d m=%d ns=%d o1=%d o2=%d; m * ns %% 64, o1 in {64,128}, o2 %% 4)", b, n, m, nsample, o1, o2);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    const size_t lds = sizeof(float) * ((size_t)o1 * GP_XS + (size_t)2 * GP_KT * 64);
    if (o1 == 64)
        hipLaunchKernelGGL((pgather_gemm2_kernel<1>), dim3((unsigned)(rows / 64)), dim3(256), lds, as_stream(stream), o2, n, m, nsample, pmat, p_stride,
                           xyz, new_xyz, nbr, w1x, b1, relu1, w2t, b2, relu2, out, gate, gate_limit);
    else
        hipLaunchKernelGGL((pgather_gemm2_kernel<2>), dim3((unsigned)(rows / 64)), dim3(256), lds, as_stream(stream), o2, n, m, nsample, pmat, p_stride,
                           xyz, new_xyz, nbr, w1x, b1, relu1, w2t, b2, relu2, out, gate, gate_limit);
    return check_launch("ws3d_pgather_gemm2");
}

extern "C" int ws3d_pgather_rows(int b, int n, int m, int nsample, int o1, const float *pmat, int p_stride, const float *xyz, const float *new_xyz,
                                 const int32_t *nbr, const float *w1x, const float *b1, int relu1, float *out, ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * m * nsample;
    const uintptr_t al = reinterpret_cast<uintptr_t>(pmat) | reinterpret_cast<uintptr_t>(w1x) | reinterpret_cast<uintptr_t>(b1) | reinterpret_cast<uintptr_t>(out);
    if (b < 0 || n <= 0 || m <= 0 || nsample <= 0 || o1 <= 0 || (o1 & 3) || p_stride < o1 || (p_stride & 3) || !pmat || !xyz || !new_xyz || !nbr ||
        !w1x || !out || (al & 15)) {
        set_error("ws3d_pgather_rows: unsupported shape (b=%d n=%d m=%d ns=%d o1=%d stride=%d; o1, stride %% 4, 16-byte aligned)", b, n, m, nsample, o1, p_stride);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    const long units = rows * (o1 >> 2);
    const unsigned grid = (unsigned)((units + 255) / 256 < 8192 ? (units + 255) / 256 : 8192);
    hipLaunchKernelGGL(pgather_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), rows, o1, n, m, nsample, pmat, p_stride, xyz, new_xyz, nbr, w1x, b1,
                       relu1, out);
    return check_launch("ws3d_pgather_rows");
}

extern "C" int ws3d_qinterp_rows(int b, int n, int m, int o, const float *q, const int32_t *idx, const float *weight, const float *lin,
                                 const float *skip, int c1, const float *wb, const float *bias, int relu, float *out, ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * n;
    const uintptr_t al = reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(lin) | reinterpret_cast<uintptr_t>(wb) |
                         reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(out);
    if (b < 0 || n <= 0 || m <= 0 || o <= 0 || (o & 3) || !q || !idx || !weight || !out || (al & 15) || c1 < 0 ||
        (!lin && c1 > 0 && (!skip || !wb)) || (!lin && c1 > 4)) {
        set_error("ws3d_qinterp_rows: unsupported shape (b=%d n=%d m=%d o=%d c1=%d; o %% 4, 16-byte aligned, c1 <= 4 without lin)", b, n, m, o, c1);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    const long units = rows * (o >> 2);
    const unsigned grid = (unsigned)((units + 255) / 256 < 16384 ? (units + 255) / 256 : 16384);
    hipLaunchKernelGGL(qinterp_rows_kernel, dim3(grid), dim3(256), 0, as_stream(stream), rows, o, n, m, q, idx, weight, lin, skip, lin ? 0 : c1, wb, bias,
                       relu, out);
    return check_launch("ws3d_qinterp_rows");
}

extern "C" int ws3d_compact_pairs_count(long centres, int nsample, const int32_t *nbr, int32_t *cnt, ws3d_stream_t stream) {
    using namespace ws3d;
    if (centres < 0 || nsample <= 0 || !nbr || !cnt) { set_error("ws3d_compact_pairs_count: invalid argument"); return WS3D_E_INVALID; }
    if (centres == 0) return WS3D_OK;
    hipLaunchKernelGGL(pair_count_kernel, dim3((unsigned)((centres + 255) / 256)), dim3(256), 0, as_stream(stream), centres, nsample, nbr, cnt);
    return check_launch("ws3d_compact_pairs_count");
}

extern "C" int ws3d_compact_pairs_rows(long centres, int nsample, const int32_t *nbr, const int32_t *cnt, const int32_t *incl, int32_t *rowc,
                                       int32_t *rowsrc, int32_t *total, ws3d_stream_t stream) {
    using namespace ws3d;
    if (centres <= 0 || nsample <= 0 || !nbr || !cnt || !incl || !rowc || !rowsrc || !total) {
        set_error("ws3d_compact_pairs_rows: invalid argument");
        return WS3D_E_INVALID;
    }
    hipLaunchKernelGGL(pair_rows_kernel, dim3((unsigned)((centres * nsample + 255) / 256)), dim3(256), 0, as_stream(stream), centres, nsample, nbr, cnt,
                       incl, rowc, rowsrc, total);
    return check_launch("ws3d_compact_pairs_rows");
}

extern "C" int ws3d_compact_pairs(long centres, int nsample, const int32_t *nbr, int32_t *rowc, int32_t *rowsrc, int32_t *total,
                                  ws3d_stream_t stream) {
    using namespace ws3d;
    if (centres < 0 || nsample <= 0 || !nbr || !rowc || !rowsrc || !total) { set_error("ws3d_compact_pairs: invalid argument"); return WS3D_E_INVALID; }
    if (centres == 0) return WS3D_OK;
    hipLaunchKernelGGL(pair_compact_kernel, dim3((unsigned)((centres + 255) / 256)), dim3(256), 0, as_stream(stream), centres, nsample, nbr, rowc, rowsrc, total);
    return check_launch("ws3d_compact_pairs");
}

extern "C" int ws3d_pgather_gemm2_compact(int b, int n, int m, long max_rows, int o1, int o2, const float *pmat, int p_stride, const float *xyz,
                                          const float *new_xyz, const int32_t *rowc, const int32_t *rowsrc, const int32_t *total, const float *w1x,
                                          const float *b1, int relu1, const float *w2t, const float *b2, int relu2, float *out, long limit, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b <= 0 || n <= 0 || m <= 0 || max_rows <= 0 || (o1 != 64 && o1 != 128 && o1 != 256) || o2 <= 0 || (o2 & 3) || p_stride < o1 || !pmat || !xyz || !new_xyz ||
        !rowc || !rowsrc || !total || !w1x || !w2t || !out || (reinterpret_cast<uintptr_t>(w2t) & 15)) {
        set_error("ws3d_pgather_gemm2_compact: unsupported shape (b=%d n=%d m=%d rows<=%ld o1=%d o2=%d; o1 in {64,128,256}, o2 %% 4)", b, n, m, max_rows, o1, o2);
        return WS3D_E_UNSUPPORTED;
    }
    const size_t lds = sizeof(float) * ((size_t)o1 * GP_XS + (size_t)2 * GP_KT * 64);
    const dim3 grid((unsigned)((max_rows + 63) / 64), (unsigned)((o2 + 63) / 64));
    if (o1 == 64)
        hipLaunchKernelGGL((pgather_gemm2_compact_kernel<1>), grid, dim3(256), lds, as_stream(stream), o2, n, m, pmat, p_stride, xyz, new_xyz, rowc,
                           rowsrc, total, w1x, b1, relu1, w2t, b2, relu2, out, limit);
    else if (o1 == 128)
        hipLaunchKernelGGL((pgather_gemm2_compact_kernel<2>), grid, dim3(256), lds, as_stream(stream), o2, n, m, pmat, p_stride, xyz, new_xyz, rowc,
                           rowsrc, total, w1x, b1, relu1, w2t, b2, relu2, out, limit);
    else {
        if (int rc = raise_lds_cap((const void *)pgather_gemm2_compact_kernel<4>, lds, "ws3d_pgather_gemm2_compact")) return rc;          // 74 KB of LDS: above the default limit
        hipLaunchKernelGGL((pgather_gemm2_compact_kernel<4>), grid, dim3(256), lds, as_stream(stream), o2, n, m, pmat, p_stride, xyz, new_xyz, rowc,
                           rowsrc, total, w1x, b1, relu1, w2t, b2, relu2, out, limit);
    }
    return check_launch("ws3d_pgather_gemm2_compact");
}

extern "C" int ws3d_pgather_gemm3_compact(int b, int n, int m, long max_rows, int o1, int o2, int o3, const float *pmat, int p_stride, const float *xyz,
                                          const float *new_xyz, const int32_t *rowc, const int32_t *rowsrc, const int32_t *total, const float *w1x,
                                          const float *b1, int relu1, const float *w2t, const float *b2, int relu2, const float *w3t, const float *b3,
                                          float *out, int out_stride, long limit, ws3d_stream_t stream) {
    using namespace ws3d;
    const size_t o2p = ((size_t)o2 + GP_KT - 1) / GP_KT * GP_KT;
    const size_t lds = sizeof(float) * (((size_t)o1 + o2p) * GP_XS + (size_t)2 * GP_KT * 64);
    const uintptr_t al = reinterpret_cast<uintptr_t>(w2t) | reinterpret_cast<uintptr_t>(w3t);
    if (b <= 0 || n <= 0 || m <= 0 || max_rows <= 0 || (o1 != 64 && o1 != 128) || o2 <= 0 || (o2 & 3) || o3 <= 0 || (o3 & 127) || p_stride < o1 || !pmat || !xyz ||
        !new_xyz || !rowc || !rowsrc || !total || !w1x || !w2t || !w3t || !out || out_stride < o3 || (al & 15) || lds > 160 * 1024 - 1024) {
        set_error("ws3d_pgather_gemm3_compact: unsupported shape (b=%d n=%d m=%d rows<=%ld o1=%d o2=%d o3=%d; o1 in {64,128}, o2 %% 4, o3 %% 128, %zu B of LDS)",
                  b, n, m, max_rows, o1, o2, o3, lds);
        return WS3D_E_UNSUPPORTED;
    }
    const dim3 grid((unsigned)((max_rows + 63) / 64));
    if (o1 == 64) {
        if (int rc = raise_lds_cap((const void *)pgather_gemm3_compact_kernel<1>, lds, "ws3d_pgather_gemm3_compact")) return rc;
        hipLaunchKernelGGL((pgather_gemm3_compact_kernel<1>), grid, dim3(256), lds, as_stream(stream), o2, o3, n, m, pmat, p_stride, xyz, new_xyz, rowc,
                           rowsrc, total, w1x, b1, relu1, w2t, b2, relu2, w3t, b3, out, out_stride, limit);
    } else {
        if (int rc = raise_lds_cap((const void *)pgather_gemm3_compact_kernel<2>, lds, "ws3d_pgather_gemm3_compact")) return rc;
        hipLaunchKernelGGL((pgather_gemm3_compact_kernel<2>), grid, dim3(256), lds, as_stream(stream), o2, o3, n, m, pmat, p_stride, xyz, new_xyz, rowc,
                           rowsrc, total, w1x, b1, relu1, w2t, b2, relu2, w3t, b3, out, out_stride, limit);
    }
    return check_launch("ws3d_pgather_gemm3_compact");
}

// kind 3: ws3d_pgather_gemm3_compact of both argument blocks, 2: ws3d_pgather_gemm2_compact (rows -> mid), 1: ws3d_gemm_pool_compact (mid -> out)
extern "C" int ws3d_compact_mlp_pair(int kind, const ws3d_compact_mlp_args *p0, const ws3d_compact_mlp_args *p1, ws3d_stream_t stream) {
    using namespace ws3d;
    if (!p0 || !p1 || kind < 1 || kind > 3) { set_error("ws3d_compact_mlp_pair: invalid argument (kind=%d)", kind); return WS3D_E_INVALID; }
    const ws3d_compact_mlp_args *ps[2] = {p0, p1};
    CompactMlpArgs a[2];
    long tiles[2];
    size_t lds = 0;
    unsigned gy = 1;
    for (int i = 0; i < 2; ++i) {
        const ws3d_compact_mlp_args &q = *ps[i];
        bool ok = q.max_rows > 0 && q.rowc && q.total && q.o2 > 0 && !(q.o2 & 3);
        if (kind >= 2) {
            ok = ok && q.b > 0 && q.n > 0 && q.m > 0 && q.o1 == p0->o1 && (q.o1 == 64 || q.o1 == 128 || (kind == 2 && q.o1 == 256)) && q.p_stride >= q.o1 && q.pmat &&
                 q.xyz && q.new_xyz && q.rowsrc && q.w1x && q.w2t && !(reinterpret_cast<uintptr_t>(q.w2t) & 15);
        }
        if (kind == 3) {
            const size_t o2p = ((size_t)q.o2 + GP_KT - 1) / GP_KT * GP_KT;
            lds = std::max(lds, sizeof(float) * (((size_t)q.o1 + o2p) * GP_XS + (size_t)2 * GP_KT * 64));
            ok = ok && q.o3 > 0 && !(q.o3 & 127) && q.w3t && !(reinterpret_cast<uintptr_t>(q.w3t) & 15) && q.out && q.out_stride >= q.o3 && lds <= 160 * 1024 - 1024;
        } else if (kind == 2) {
            lds = sizeof(float) * ((size_t)q.o1 * GP_XS + (size_t)2 * GP_KT * 64);
            ok = ok && q.mid;
            gy = std::max(gy, (unsigned)((q.o2 + 63) / 64));
        } else {
            ok = ok && q.o3 > 0 && !(q.o3 & 63) && q.mid && q.w3t && q.out && q.out_stride >= q.o3 &&
                 !((reinterpret_cast<uintptr_t>(q.mid) | reinterpret_cast<uintptr_t>(q.w3t)) & 15);
        }
        if (!ok) {
            set_error("ws3d_compact_mlp_pair(kind %d): block %d not covered (rows<=%ld o1=%d o2=%d o3=%d; the shapes of the single-scale entry, equal o1)", kind, i,
                      q.max_rows, q.o1, q.o2, q.o3);
            return WS3D_E_UNSUPPORTED;
        }
        tiles[i] = ((q.max_rows + 63) / 64) * (kind == 1 ? q.o3 / 64 : 1);
        a[i] = CompactMlpArgs{q.pmat, q.xyz, q.new_xyz, q.rowc, q.rowsrc, q.total, q.w1x, q.b1, q.w2t, q.b2, q.w3t, q.b3, q.mid, q.out, q.limit,
                              q.o2, q.o3, q.n, q.m, q.p_stride, q.relu1, q.relu2, q.out_stride, (q.o2 + 63) / 64, 0};
    }
    if (tiles[0] + tiles[1] > 0x7fffffffL) { set_error("ws3d_compact_mlp_pair: too many rows"); return WS3D_E_UNSUPPORTED; }
    const unsigned total_x = (unsigned)(tiles[0] + tiles[1]);
    // kinds 2 and 1 walk their tiles: ws3d_tune key 4 caps the workgroups along x (0: one per tile, the launch of round 5)
    const unsigned cap = (kind != 3 && g_tune[TUNE_PAIR_WGS] > 0) ? (unsigned)g_tune[TUNE_PAIR_WGS] : total_x;
    const dim3 grid(std::max(1u, std::min(total_x, cap)), kind == 2 ? gy : 1u);
    const unsigned split = (unsigned)tiles[0];
    hipStream_t st = as_stream(stream);
#define WS3D_PAIR_GO(KERN)                                                                            \
    do {                                                                                              \
        if (int rc = raise_lds_cap((const void *)KERN, lds, "ws3d_compact_mlp_pair")) return rc;      \
        hipLaunchKernelGGL(KERN, grid, dim3(256), lds, st, a[0], a[1], split, total_x);               \
    } while (0)
    if (kind == 3) {
        if (p0->o1 == 64) WS3D_PAIR_GO(pgather_gemm3_compact_pair_kernel<1>); else WS3D_PAIR_GO(pgather_gemm3_compact_pair_kernel<2>);
    } else if (kind == 2) {
        if (p0->o1 == 64) WS3D_PAIR_GO(pgather_gemm2_compact_pair_kernel<1>);
        else if (p0->o1 == 128) WS3D_PAIR_GO(pgather_gemm2_compact_pair_kernel<2>);
        else WS3D_PAIR_GO(pgather_gemm2_compact_pair_kernel<4>);
    } else {
        hipLaunchKernelGGL(gemm_pool_compact_pair_kernel, grid, dim3(256), 0, st, a[0], a[1], split, total_x);
    }
#undef WS3D_PAIR_GO
    return check_launch("ws3d_compact_mlp_pair");
}

extern "C" int ws3d_gemm_pool_compact(long max_rows, int k_dim, int o_dim, const float *x_rows, const int32_t *rowc, const int32_t *total,
                                      const float *wt, const float *bias, float *out, int out_stride, long limit, ws3d_stream_t stream) {
    using namespace ws3d;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x_rows) | reinterpret_cast<uintptr_t>(wt);
    if (max_rows <= 0 || k_dim <= 0 || (k_dim & 3) || o_dim <= 0 || (o_dim & 63) || !x_rows || !rowc || !total || !wt || !out || out_stride < o_dim ||
        (al & 15)) {
        set_error("ws3d_gemm_pool_compact: unsupported shape (rows<=%ld k=%d o=%d; o %% 64, k %% 4)", max_rows, k_dim, o_dim);
        return WS3D_E_UNSUPPORTED;
    }
    const unsigned grid = (unsigned)(((max_rows + 63) / 64) * (o_dim / 64));
    hipLaunchKernelGGL(gemm_pool_compact_kernel, dim3(grid), dim3(256), 0, as_stream(stream), k_dim, o_dim, x_rows, rowc, total, wt, bias, out, out_stride, limit);
    return check_launch("ws3d_gemm_pool_compact");
}

