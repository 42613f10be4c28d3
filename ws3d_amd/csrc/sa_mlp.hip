// sa_mlp.hip -- the first set-abstraction level's SharedMLP + pool as ONE kernel.
//
// At SA level 1 the grouped tensor has 4 channels (dx, dy, dz, intensity) and 0.5-1 M columns per
// batch, the MLP widths are 16-64.  As three GEMMs + pool that is ~2 GB of intermediate traffic
// for ~8 GFLOP (0.55 ms per 8 scenes); here one lane carries one (centre, sample) column through
// all three layers in registers (weights broadcast from LDS, W^T layout so that one 16-byte LDS
// read feeds four outputs), the nsample lanes of a centre are max-reduced with DPP, and only the
// pooled (centre, C3) rows leave the chip.  bias + ReLU of the last layer are applied after the
// pool (they commute with max exactly, nn_blocks.forward_then_max).  fp32 FMA chains in a fixed
// order: same math as the GEMM path up to summation order (both are within ~1e-6 relative of the
// exact value; the network-level tests hold the 1e-4 tolerance of the GEMM path).
//
// Not a reference entry point: ws3d_amd/fastpath.py uses it when the shapes match
// (C0 = 4, C1, C2 <= 32, C3 <= 64, nsample in {16, 32}); anything else takes the GEMM chain.
#include <cstdlib>

#include "common.h"
#include "compact_pool.h"

namespace ws3d {

template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(256) void sa_mlp3_pool_kernel(long rows, const float *__restrict__ x,
                                                           const float *__restrict__ w1t, const float *__restrict__ b1,
                                                           const float *__restrict__ w2t, const float *__restrict__ b2,
                                                           const float *__restrict__ w3t, const float *__restrict__ b3,
                                                           int relu3, float *__restrict__ out, int out_stride) {
    static_assert(C1 % 4 == 0 && C2 % 4 == 0 && C3 % 16 == 0 && (NS == 16 || NS == 32), "shape");
    __shared__ __attribute__((aligned(16))) float s_w1[4 * C1], s_w2[C1 * C2], s_w3[C2 * C3], s_b1[C1], s_b2[C2], s_b3[C3];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4 * C1; i += 256) s_w1[i] = w1t[i];
    for (int i = tid; i < C1 * C2; i += 256) s_w2[i] = w2t[i];
    for (int i = tid; i < C2 * C3; i += 256) s_w3[i] = w3t[i];
    if (tid < C1) s_b1[tid] = b1[tid];
    if (tid < C2) s_b2[tid] = b2[tid];
    if (tid < C3) s_b3[tid] = b3[tid];
    __syncthreads();

    const long r = (long)blockIdx.x * 256 + tid;          // rows is a multiple of NS; a group never straddles the end
    const bool live = r < rows;
    const float4 xin = live ? reinterpret_cast<const float4 *>(x)[r] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float xi[4] = {xin.x, xin.y, xin.z, xin.w};

    float h1[C1];
#pragma unroll
    for (int o = 0; o < C1; o += 4) {
        float4 a = *reinterpret_cast<const float4 *>(s_b1 + o);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 w = *reinterpret_cast<const float4 *>(s_w1 + i * C1 + o);
            a.x = __builtin_fmaf(xi[i], w.x, a.x); a.y = __builtin_fmaf(xi[i], w.y, a.y);
            a.z = __builtin_fmaf(xi[i], w.z, a.z); a.w = __builtin_fmaf(xi[i], w.w, a.w);
        }
        h1[o] = fmaxf(a.x, 0.f); h1[o + 1] = fmaxf(a.y, 0.f); h1[o + 2] = fmaxf(a.z, 0.f); h1[o + 3] = fmaxf(a.w, 0.f);
    }
    float h2[C2];
#pragma unroll
    for (int o = 0; o < C2; o += 4) {
        float4 a = *reinterpret_cast<const float4 *>(s_b2 + o);
#pragma unroll
        for (int i = 0; i < C1; ++i) {
            const float4 w = *reinterpret_cast<const float4 *>(s_w2 + i * C2 + o);
            a.x = __builtin_fmaf(h1[i], w.x, a.x); a.y = __builtin_fmaf(h1[i], w.y, a.y);
            a.z = __builtin_fmaf(h1[i], w.z, a.z); a.w = __builtin_fmaf(h1[i], w.w, a.w);
        }
        h2[o] = fmaxf(a.x, 0.f); h2[o + 1] = fmaxf(a.y, 0.f); h2[o + 2] = fmaxf(a.z, 0.f); h2[o + 3] = fmaxf(a.w, 0.f);
    }
    const int lane = tid & 63;
    const bool writer = live && (NS == 16 ? (lane & 15) == 0 : (lane & 31) == 16);
    float *orow = out + (r / NS) * (long)out_stride;
#pragma unroll 1
    for (int o0 = 0; o0 < C3; o0 += 16) {
        float acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int i = 0; i < C2; ++i) {
#pragma unroll
            for (int q = 0; q < 16; q += 4) {
                const float4 w = *reinterpret_cast<const float4 *>(s_w3 + i * C3 + o0 + q);
                acc[q] = __builtin_fmaf(h2[i], w.x, acc[q]); acc[q + 1] = __builtin_fmaf(h2[i], w.y, acc[q + 1]);
                acc[q + 2] = __builtin_fmaf(h2[i], w.z, acc[q + 2]); acc[q + 3] = __builtin_fmaf(h2[i], w.w, acc[q + 3]);
            }
        }
        // max over the NS lanes of the centre (a dead lane contributes its own finite garbage only
        // to dead groups: rows % NS == 0)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float v = row16_max(acc[q]);
            if (NS == 32) asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v));
            v += s_b3[o0 + q];
            acc[q] = relu3 ? fmaxf(v, 0.f) : v;
        }
        if (writer) {
#pragma unroll
            for (int q = 0; q < 16; q += 4)
                *reinterpret_cast<float4 *>(orow + o0 + q) = make_float4(acc[q], acc[q + 1], acc[q + 2], acc[q + 3]);
        }
    }
}

// ---- the (32, 32, 64) scale on the fp32 matrix cores, all three layers chained IN REGISTERS ----------------------------------
// Every layer is computed transposed, out^T = W^T X^T, with v_mfma_f32_32x32x2_f32: the A operand is W^T (lane l holds
// W^T[o = l % 32][k]), the B operand X^T (lane l holds X[row = l % 32][k]; lanes 0-31 supply the first K element of a step,
// lanes 32-63 the second), and the accumulator comes out as lane = row, register v (with the lane's half h = l / 32) = channel
// kp(v, h) = 8 (v / 4) + 4 h + v % 4.  That is exactly the shape of the NEXT layer's B operand if step v of the next layer pairs
// the K elements (kp(v, 0), kp(v, 1)) -- the order of the K steps is free as long as the A operand follows it -- so
// bias + ReLU are applied to the accumulator registers and they are fed straight back in: no shuffle, no LDS, no barrier
// between the layers.  The weights live in registers (2 + 16 + 32 per lane, loaded once per wave), a wave walks over tiles of
// 32 grouped rows (one centre at nsample 32, two at 16); the last layer is multiplied the other way round (same registers,
// operands swapped) so that the pool runs over registers.
// 50 MFMAs per tile = 3,200 matrix-core cycles against ~6,500 VALU cycles of the kernel above (wide scale; the narrow one,
// 16-16-32, uses half of the accumulator rows in layers 1 and 2: 18 MFMAs per tile).
typedef float sa_f16 __attribute__((ext_vector_type(16)));

template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(256) void sa_mlp3_pool_mfma_kernel(long tiles, const float *__restrict__ x, const float *__restrict__ w1t,
                                                                const float *__restrict__ b1, const float *__restrict__ w2t,
                                                                const float *__restrict__ b2, const float *__restrict__ w3t,
                                                                const float *__restrict__ b3, int relu3, float *__restrict__ out,
                                                                int out_stride) {
    static_assert((NS == 16 || NS == 32) && (C1 == 16 || C1 == 32) && (C2 == 16 || C2 == 32) && (C3 == 32 || C3 == 64), "shape");
    constexpr int V1 = C1 / 2, V2 = C2 / 2, NB3 = C3 / 32;      // K steps of layers 2 / 3 (channel pairs), 32-channel blocks of layer 3
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    auto kp = [&](int v) { return 8 * (v / 4) + 4 * h + (v % 4); };
    // a 16-channel layer fills half of the 32 accumulator rows: its weights beyond the width are zero, its K steps half as many
    float a1[2], a2[V1], a3[NB3][V2], bb1[V1], bb2[V2], b3v[NB3];
#pragma unroll
    for (int j = 0; j < 2; ++j) a1[j] = c < C1 ? w1t[(2 * j + h) * C1 + c] : 0.f;
#pragma unroll
    for (int v = 0; v < V1; ++v) {
        a2[v] = c < C2 ? w2t[kp(v) * C2 + c] : 0.f;
        bb1[v] = b1[kp(v)];
    }
#pragma unroll
    for (int v = 0; v < V2; ++v) {
#pragma unroll
        for (int blk = 0; blk < NB3; ++blk) a3[blk][v] = w3t[kp(v) * C3 + blk * 32 + c];
        bb2[v] = b2[kp(v)];
    }
#pragma unroll
    for (int blk = 0; blk < NB3; ++blk) b3v[blk] = b3[blk * 32 + c];
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long tile = wave; tile < tiles; tile += nwaves) {
        const float4 xr = reinterpret_cast<const float4 *>(x)[tile * 32 + c];          // row c of the tile (both halves)
        sa_f16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], h ? xr.y : xr.x, acc, 0, 0, 0);   // k = h
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], h ? xr.w : xr.z, acc, 0, 0, 0);   // k = 2 + h
        float act[16];
#pragma unroll
        for (int v = 0; v < V1; ++v) act[v] = fmaxf(acc[v] + bb1[v], 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int v = 0; v < V1; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[v], act[v], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < V2; ++v) act[v] = fmaxf(acc[v] + bb2[v], 0.f);
#pragma unroll
        for (int blk = 0; blk < NB3; ++blk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            // the LAST layer the other way round, out = X W (the very same registers, operands swapped): lane = channel,
            // register v (+ half) = row kp(v, h) -- so the pool over the rows of a centre is a maximum over REGISTERS (15
            // v_max + one exchange between the halves) instead of five DPP stages for each of 16 registers
#pragma unroll
            for (int v = 0; v < V2; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(act[v], a3[blk][v], acc, 0, 0, 0);
            const float bias = b3v[blk];
            if (NS == 32) {
                float m = acc[0];
#pragma unroll
                for (int v = 1; v < 16; ++v) m = fmaxf(m, acc[v]);
                m = fmaxf(m, __shfl_xor(m, 32));                     // rows 4h .. of the other half
                m += bias;
                if (relu3) m = fmaxf(m, 0.f);
                if (h == 0) out[tile * (long)out_stride + blk * 32 + c] = m;
            } else {
                float m0 = acc[0], m1 = acc[8];                      // rows 0-15 (registers 0-7) / rows 16-31 (registers 8-15)
#pragma unroll
                for (int v = 1; v < 8; ++v) { m0 = fmaxf(m0, acc[v]); m1 = fmaxf(m1, acc[8 + v]); }
                m0 = fmaxf(m0, __shfl_xor(m0, 32));
                m1 = fmaxf(m1, __shfl_xor(m1, 32));
                float m = (h ? m1 : m0) + bias;                      // half 0 writes the first centre, half 1 the second
                if (relu3) m = fmaxf(m, 0.f);
                out[(tile * 2 + h) * (long)out_stride + blk * 32 + c] = m;
            }
        }
    }
}

// ---- the same chain over COMPACT (centre, sample) pairs (gemm_pool.hip: a ball-query list is mostly padding, a padded row
// repeats row 0 of its centre and cannot change the maximum).  Row t = (centre rowc[t], source point rowsrc[t]); the 4-channel
// input row [dx dy dz f] is built here from xyz / new_xyz / the one feature channel -- no grouped tensor -- and the pool is an
// integer atomic max of the ReLU'd values (>= 0) into the centre's row, which the caller zeroes.  *total rows; the grid is
// sized for the worst case and waves walk the tiles that exist.
#ifndef SA1_ABL
#define SA1_ABL 0
#endif
template <int C1, int C2, int C3>
__global__ __launch_bounds__(256) void sa_mlp3_compact_mfma_kernel(int n, int m, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                   const float *__restrict__ feat, const int32_t *__restrict__ rowc,
                                                                   const int32_t *__restrict__ rowsrc, const int32_t *__restrict__ total,
                                                                   const float *__restrict__ w1t, const float *__restrict__ b1,
                                                                   const float *__restrict__ w2t, const float *__restrict__ b2,
                                                                   const float *__restrict__ w3t, const float *__restrict__ b3,
                                                                   float *__restrict__ out, int out_stride, long limit) {
    static_assert((C1 == 16 || C1 == 32) && (C2 == 16 || C2 == 32) && (C3 == 32 || C3 == 64), "shape");
    constexpr int V1 = C1 / 2, V2 = C2 / 2, NB3 = C3 / 32;
    const long T = *total;
    const long tiles = (T + 31) / 32;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    if (wave >= tiles || (limit >= 0 && T > limit)) return;       // beyond the limit sa_mlp3_lists_mfma_kernel runs instead
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    auto kp = [&](int v) { return 8 * (v / 4) + 4 * h + (v % 4); };
    float a1[2], a2[V1], a3[NB3][V2], bb1[V1], bb2[V2], b3v[NB3];
#pragma unroll
    for (int j = 0; j < 2; ++j) a1[j] = c < C1 ? w1t[(2 * j + h) * C1 + c] : 0.f;
#pragma unroll
    for (int v = 0; v < V1; ++v) {
        a2[v] = c < C2 ? w2t[kp(v) * C2 + c] : 0.f;
        bb1[v] = b1[kp(v)];
    }
#pragma unroll
    for (int v = 0; v < V2; ++v) {
#pragma unroll
        for (int blk = 0; blk < NB3; ++blk) a3[blk][v] = w3t[kp(v) * C3 + blk * 32 + c];
        bb2[v] = b2[kp(v)];
    }
#pragma unroll
    for (int blk = 0; blk < NB3; ++blk) b3v[blk] = b3[blk * 32 + c];
    // the row of a tile is two dependent round trips away (compact row -> (centre, source point) -> coordinates): the indices are
    // fetched two tiles ahead and the coordinates one tile ahead, under the matrix chain of the tile in hand
    auto row_index = [&](long tile, int &cm, int &src) {
        const long t = min(tile * 32 + c, T - 1);                   // (tiles behind the end repeat the last row; never pooled)
        cm = rowc[t];
        src = rowsrc[t];
    };
    auto row_fetch = [&](int cm, int src) {
        const size_t p = (size_t)(cm / m) * n + (size_t)src;
        const float *pr = xyz + p * 3, *cr = new_xyz + (size_t)cm * 3;
        return make_float4(pr[0] - cr[0], pr[1] - cr[1], pr[2] - cr[2], feat[p]);
    };
    int cm_n, src_n;
    row_index(wave, cm_n, src_n);
    float4 xr_n = row_fetch(cm_n, src_n);
    row_index(wave + nwaves, cm_n, src_n);
    for (long tile = wave; tile < tiles; tile += nwaves) {
        float4 xr = xr_n;
#if SA1_ABL & 2     // (scripts/ubench/sa1_compact_ablation.sh: what the gather costs -- NOT the operator)
        xr = make_float4((float)c, (float)(tile & 255), 1.f, 0.5f);
#else
        if (tile + nwaves < tiles) {
            xr_n = row_fetch(cm_n, src_n);
            row_index(tile + 2 * nwaves, cm_n, src_n);
        }
#endif
        sa_f16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], h ? xr.y : xr.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], h ? xr.w : xr.z, acc, 0, 0, 0);
        float act[16];
#pragma unroll
        for (int v = 0; v < V1; ++v) act[v] = fmaxf(acc[v] + bb1[v], 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int v = 0; v < V1; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[v], act[v], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < V2; ++v) act[v] = fmaxf(acc[v] + bb2[v], 0.f);
        // the centres of the 16 consecutive rows this half pools (compact_pool.h), once for all channel blocks
        int cen[16];
#if SA1_ABL & 2
#pragma unroll
        for (int r = 0; r < 16; ++r) { const long t = tile * 32 + 16 * h + r; cen[r] = t < T ? (int)(t >> 4) : -1; }
#else
        compact_centres16(rowc, tile * 32 + 16 * h, T, cen);
#endif
#pragma unroll
        for (int blk = 0; blk < NB3; ++blk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int v = 0; v < V2; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(act[v], a3[blk][v], acc, 0, 0, 0);
            const float bias = b3v[blk];
#if SA1_ABL & 1     // (what the atomic epilogue costs: one plain store per tile and block instead)
            float mx = 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) mx = fmaxf(mx, cen[v] < 0 ? 0.f : fmaxf(acc[v] + bias, 0.f));
            out[(long)(tile & 1023) * out_stride + blk * 32 + c] = mx;
#else
            compact_pool_atomic(acc, bias, cen, out + blk * 32 + c, out_stride);
#endif
        }
    }
}

// ---- the same chain over ALL rows of the neighbour lists, without a grouped tensor: row r = (centre r / NS, source point
// nbr[r]) is built here like in the compact kernel, the pool runs over registers like in sa_mlp3_pool_mfma_kernel (a tile of 32
// rows holds one centre at NS = 32, two at 16) and is STORED -- no atomics, no zeroed output.  This is the dense side of the
// device-side dispatch: launched next to the compact kernel with a gate on the pair total (run iff *gate > gate_limit; a null
// gate always runs), it replaces ws3d_query_and_group_nlc + ws3d_sa_mlp3_pool (25 MB of grouped rows written and read back per
// batch of 8 scenes).  Bit-identical to both: a row's activations are its own, the maximum is order-free.
template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(256) void sa_mlp3_lists_mfma_kernel(long tiles, int n, int m, const float *__restrict__ xyz, const float *__restrict__ new_xyz,
                                                                 const float *__restrict__ feat, const int32_t *__restrict__ nbr,
                                                                 const float *__restrict__ w1t, const float *__restrict__ b1,
                                                                 const float *__restrict__ w2t, const float *__restrict__ b2,
                                                                 const float *__restrict__ w3t, const float *__restrict__ b3, int relu3,
                                                                 float *__restrict__ out, int out_stride, const int32_t *__restrict__ gate,
                                                                 long gate_limit) {
    static_assert((NS == 16 || NS == 32) && (C1 == 16 || C1 == 32) && (C2 == 16 || C2 == 32) && (C3 == 32 || C3 == 64), "shape");
    if (gate && (long)*gate <= gate_limit) return;
    constexpr int V1 = C1 / 2, V2 = C2 / 2, NB3 = C3 / 32;
    const int lane = threadIdx.x & 63, h = lane >> 5, c = lane & 31;
    auto kp = [&](int v) { return 8 * (v / 4) + 4 * h + (v % 4); };
    float a1[2], a2[V1], a3[NB3][V2], bb1[V1], bb2[V2], b3v[NB3];
#pragma unroll
    for (int j = 0; j < 2; ++j) a1[j] = c < C1 ? w1t[(2 * j + h) * C1 + c] : 0.f;
#pragma unroll
    for (int v = 0; v < V1; ++v) {
        a2[v] = c < C2 ? w2t[kp(v) * C2 + c] : 0.f;
        bb1[v] = b1[kp(v)];
    }
#pragma unroll
    for (int v = 0; v < V2; ++v) {
#pragma unroll
        for (int blk = 0; blk < NB3; ++blk) a3[blk][v] = w3t[kp(v) * C3 + blk * 32 + c];
        bb2[v] = b2[kp(v)];
    }
#pragma unroll
    for (int blk = 0; blk < NB3; ++blk) b3v[blk] = b3[blk * 32 + c];
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    for (long tile = wave; tile < tiles; tile += nwaves) {
        float4 xr;
        {
            const long r = tile * 32 + c;
            const long cm = r / NS;
            const size_t p = (size_t)(cm / m) * n + (size_t)nbr[r];
            const float *pr = xyz + p * 3, *cr = new_xyz + (size_t)cm * 3;
            xr = make_float4(pr[0] - cr[0], pr[1] - cr[1], pr[2] - cr[2], feat[p]);
        }
        sa_f16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[0], h ? xr.y : xr.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[1], h ? xr.w : xr.z, acc, 0, 0, 0);
        float act[16];
#pragma unroll
        for (int v = 0; v < V1; ++v) act[v] = fmaxf(acc[v] + bb1[v], 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int v = 0; v < V1; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[v], act[v], acc, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < V2; ++v) act[v] = fmaxf(acc[v] + bb2[v], 0.f);
#pragma unroll
        for (int blk = 0; blk < NB3; ++blk) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int v = 0; v < V2; ++v) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(act[v], a3[blk][v], acc, 0, 0, 0);
            const float bias = b3v[blk];
            if (NS == 32) {
                float mx = acc[0];
#pragma unroll
                for (int v = 1; v < 16; ++v) mx = fmaxf(mx, acc[v]);
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                mx += bias;
                if (relu3) mx = fmaxf(mx, 0.f);
                if (h == 0) out[tile * (long)out_stride + blk * 32 + c] = mx;
            } else {
                float m0 = acc[0], m1 = acc[8];
#pragma unroll
                for (int v = 1; v < 8; ++v) { m0 = fmaxf(m0, acc[v]); m1 = fmaxf(m1, acc[8 + v]); }
                m0 = fmaxf(m0, __shfl_xor(m0, 32));
                m1 = fmaxf(m1, __shfl_xor(m1, 32));
                float mx = (h ? m1 : m0) + bias;
                if (relu3) mx = fmaxf(mx, 0.f);
                out[(tile * 2 + h) * (long)out_stride + blk * 32 + c] = mx;
            }
        }
    }
}

// ---- two pointwise layers on rows, 128 -> 128 -> o2 (o2 <= 64): the two heads of the RPN --------------------------------------
// Same register chaining as above.  Layer 1 transposed (A = W1^T from LDS, B = the rows: lane (row, half h) holds the 64
// channels 64 h .. 64 h + 63 of its row, step s pairs channels s and 64 + s), four accumulators = all 128 output channels of
// the 32 rows; bias + ReLU in place; layer 2 the other way round (A = those registers, B = W2 from LDS, lane = output channel),
// so the result comes out row-major.  The (rows, 128) activation between the layers (67 MB per head at 131072 rows, written
// by one library GEMM and read back by the next, which is HBM-bound at o2 = 1 / 40) never exists.
template <int O2B>    // 32-column blocks of layer 2
__global__ __launch_bounds__(512) void mlp2_rows_kernel(long tiles, int o2, const float *__restrict__ x, const float *__restrict__ w1t,
                                                        const float *__restrict__ b1, int relu1, const float *__restrict__ w2t,
                                                        const float *__restrict__ b2, int relu2, float *__restrict__ out,
                                                        int *__restrict__ ticket) {
    extern __shared__ __attribute__((aligned(16))) float smem_m2[];
    // LDS banks: the two halves of a wave read rows 64 apart (layer 1) / 4 apart (layer 2); rows k >= 64 of W1 are stored with
    // their columns XOR 32, and W2's row stride is 40 / 72 floats (4 rows = 32 banks), so the halves hit disjoint banks
    constexpr int W2S = O2B == 1 ? 40 : 72;
    float *w1s = smem_m2;                       // [128][128]  W1^T as given: w1t[k][o]
    float *w2s = w1s + 128 * 128;               // [128][W2S], zero beyond o2
    float *b1s = w2s + 128 * W2S;               // [128]
    __shared__ int first_chunk[2];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    if (ticket && tid == 0) first_chunk[0] = atomicAdd(ticket, 1);
    for (int i = tid; i < 128 * 128 / 4; i += 512) {
        const int k = i >> 5, o4 = i & 31;
        reinterpret_cast<float4 *>(w1s)[k * 32 + (k >= 64 ? o4 ^ 8 : o4)] = reinterpret_cast<const float4 *>(w1t)[i];
    }
    for (int i = tid; i < 128 * O2B * 32; i += 512) {
        const int k = i / (O2B * 32), o = i - k * (O2B * 32);
        w2s[k * W2S + o] = o < o2 ? w2t[(long)k * o2 + o] : 0.f;
    }
    if (tid < 128) b1s[tid] = b1 ? b1[tid] : 0.f;
    __syncthreads();
    auto kp = [&](int v) { return 8 * (v / 4) + 4 * h + (v % 4); };
    const float *wbase = w1s + 64 * h * 128 + c + 32 * h;       // column block blk of this half's rows sits at (blk ^ h) * 32
    // (prefetching the next tile's rows into 64 more registers was measured: 60 vs 54 us at 131072 rows x (128 -> 128 -> 1))
    // chunks of 8 tiles (one per wave) are handed out by a ticket counter when the caller provides one (zero on entry): with a
    // static split a workgroup that has to wait for a CU (one per CU fits; a sampling kernel of another stream may hold eight of
    // them for 3 ms) would run its share after everybody else has finished -- measured 115 vs 68 us with 20 batches in flight.
    // One atomic per workgroup and chunk, issued a chunk ahead (per wave and tile they serialise on the one address: +40 us).
    long chunk = ticket ? first_chunk[0] : blockIdx.x;
    for (int par = 0; chunk * 8 < tiles; par ^= 1) {
        int ahead = 0;
        if (ticket && tid == 0) ahead = atomicAdd(ticket, 1);        // consumed after this chunk's work
        const long tile = chunk * 8 + (tid >> 6);
        if (tile < tiles) {
        float xv[64];
        const float4 *xp = reinterpret_cast<const float4 *>(x + (tile * 32 + c) * 128 + 64 * h);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float4 t = xp[q];
            xv[4 * q] = t.x; xv[4 * q + 1] = t.y; xv[4 * q + 2] = t.z; xv[4 * q + 3] = t.w;
        }
        sa_f16 acc[4];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[blk][i] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 64; ++s2) {
            const float *wrow = wbase + s2 * 128;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) acc[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(wrow[(blk >> 1) * 64 + ((blk & 1) ? 32 - 64 * h : 0)], xv[s2], acc[blk], 0, 0, 0);
            if ((s2 & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keeps the LDS reads of later steps from piling up in registers
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float y = acc[blk][v] + b1s[blk * 32 + kp(v)];
                acc[blk][v] = relu1 ? fmaxf(y, 0.f) : y;
            }
#pragma unroll
        for (int blk2 = 0; blk2 < O2B; ++blk2) {
            sa_f16 acc2;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int v = 0; v < 16; ++v)
                {
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[blk][v], w2s[(blk * 32 + kp(v)) * W2S + blk2 * 32 + c], acc2, 0, 0, 0);
                    if ((v & 7) == 7) __builtin_amdgcn_sched_barrier(0);
                }
            const int col = blk2 * 32 + c;
            if (col < o2) {
                const float bv = b2 ? b2[col] : 0.f;
                float *o = out + (tile * 32 + 4 * h) * (long)o2 + col;
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    float y = acc2[v] + bv;
                    if (relu2) y = fmaxf(y, 0.f);
                    o[(long)(8 * (v / 4) + (v % 4)) * o2] = y;
                }
            }
        }
        }
        if (ticket) {
            if (tid == 0) first_chunk[par ^ 1] = ahead;
            __syncthreads();
            chunk = first_chunk[par ^ 1];
        } else {
            chunk += gridDim.x;
        }
    }
}

}  // namespace ws3d

extern "C" int ws3d_sa_mlp3_pool(long rows, int nsample, int c1, int c2, int c3, const float *x_rows4,
                                 const float *w1t, const float *b1, const float *w2t, const float *b2,
                                 const float *w3t, const float *b3, int relu3, float *out, int out_stride,
                                 ws3d_stream_t stream) {
    using namespace ws3d;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x_rows4) | reinterpret_cast<uintptr_t>(out);
    if (rows < 0 || !x_rows4 || !w1t || !b1 || !w2t || !b2 || !w3t || !b3 || !out || out_stride < c3 || (out_stride & 3) ||
        (al & 15) || nsample <= 0 || rows % nsample) {
        set_error("ws3d_sa_mlp3_pool: invalid argument (rows=%ld nsample=%d stride=%d)", rows, nsample, out_stride);
        return WS3D_E_INVALID;
    }
    if (rows == 0) return WS3D_OK;
    const long blocks = (rows + 255) / 256;
    if (blocks > 0x7fffffffL) { set_error("ws3d_sa_mlp3_pool: too many rows"); return WS3D_E_UNSUPPORTED; }
    hipStream_t st = as_stream(stream);
    // on the matrix cores when the rows fill whole 32-row tiles (the VALU kernel below serves the ragged shapes)
    if (rows % 32 == 0) {
        const long tiles = rows / 32;
        const long cap1 = g_tune[TUNE_SA1_WGS] > 0 ? g_tune[TUNE_SA1_WGS] : 768;          // (ws3d_tune key 2)
    const unsigned grid = (unsigned)(tiles / 4 < cap1 ? (tiles + 3) / 4 : cap1);          // 3 workgroups per CU, waves walk over tiles
#define WS3D_SA_MFMA_CASE(A, B, C, N)                                                                                     \
        if (c1 == A && c2 == B && c3 == C && nsample == N) {                                                              \
            hipLaunchKernelGGL((sa_mlp3_pool_mfma_kernel<A, B, C, N>), dim3(grid), dim3(256), 0, st, tiles, x_rows4, w1t, b1, w2t, b2, \
                               w3t, b3, relu3, out, out_stride);                                                          \
            return check_launch("ws3d_sa_mlp3_pool");                                                                     \
        }
        WS3D_SA_MFMA_CASE(32, 32, 64, 32)
        WS3D_SA_MFMA_CASE(32, 32, 64, 16)
        WS3D_SA_MFMA_CASE(16, 16, 32, 16)
        WS3D_SA_MFMA_CASE(16, 16, 32, 32)
#undef WS3D_SA_MFMA_CASE
    }
#define WS3D_SA_MLP(A, B, C, N)                                                                                          \
    if (c1 == A && c2 == B && c3 == C && nsample == N) {                                                                 \
        hipLaunchKernelGGL((sa_mlp3_pool_kernel<A, B, C, N>), dim3((unsigned)blocks), dim3(256), 0, st, rows, x_rows4,   \
                           w1t, b1, w2t, b2, w3t, b3, relu3, out, out_stride);                                           \
        return check_launch("ws3d_sa_mlp3_pool");                                                                        \
    }
    WS3D_SA_MLP(16, 16, 32, 16)
    WS3D_SA_MLP(32, 32, 64, 32)
    WS3D_SA_MLP(16, 16, 32, 32)
    WS3D_SA_MLP(32, 32, 64, 16)
#undef WS3D_SA_MLP
    set_error("ws3d_sa_mlp3_pool: no kernel for widths (%d, %d, %d) x nsample %d", c1, c2, c3, nsample);
    return WS3D_E_UNSUPPORTED;
}

extern "C" int ws3d_mlp2_rows(long rows, int k_dim, int o1, int o2, const float *x_rows, const float *w1t, const float *b1, int relu1,
                              const float *w2t, const float *b2, int relu2, float *out, int *ticket, ws3d_stream_t stream) {
    using namespace ws3d;
    const uintptr_t al = reinterpret_cast<uintptr_t>(x_rows) | reinterpret_cast<uintptr_t>(w1t);
    if (rows < 0 || k_dim != 128 || o1 != 128 || o2 <= 0 || o2 > 64 || (rows & 31) || !x_rows || !w1t || !w2t || !out || (al & 15)) {
        set_error("ws3d_mlp2_rows: unsupported shape (rows=%ld k=%d o1=%d o2=%d; 128 -> 128 -> <= 64, rows %% 32)", rows, k_dim, o1, o2);
        return WS3D_E_UNSUPPORTED;
    }
    if (rows == 0) return WS3D_OK;
    const long tiles = rows / 32;
    const long cap = g_tune[TUNE_MLP2_WGS] > 0 ? g_tune[TUNE_MLP2_WGS] : 256;         // one 8-wave workgroup per CU, waves walk over tiles (ws3d_tune key 1)
    const unsigned grid = (unsigned)(tiles / 8 < cap ? (tiles + 7) / 8 : cap);
    const int o2b = o2 <= 32 ? 1 : 2;
    const size_t lds = sizeof(float) * (128 * 128 + 128 * (size_t)(o2b == 1 ? 40 : 72) + 128);
    auto go = [&](auto kern) -> int {
        if (int rc = raise_lds_cap((const void *)kern, lds, "ws3d_mlp2_rows")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, as_stream(stream), tiles, o2, x_rows, w1t, b1, relu1, w2t, b2, relu2, out, ticket);
        return WS3D_OK;
    };
    if (int rc = o2b == 1 ? go(mlp2_rows_kernel<1>) : go(mlp2_rows_kernel<2>)) return rc;
    return check_launch("ws3d_mlp2_rows");
}

extern "C" int ws3d_sa_mlp3_pool_compact(int b, int n, int m, long max_rows, int c1, int c2, int c3, const float *xyz, const float *new_xyz,
                                         const float *feat, const int32_t *rowc, const int32_t *rowsrc, const int32_t *total, const float *w1t,
                                         const float *b1, const float *w2t, const float *b2, const float *w3t, const float *b3, float *out,
                                         int out_stride, long limit, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b <= 0 || n <= 0 || m <= 0 || max_rows <= 0 || !xyz || !new_xyz || !feat || !rowc || !rowsrc || !total || !w1t || !b1 || !w2t || !b2 || !w3t ||
        !b3 || !out || out_stride < c3) {
        set_error("ws3d_sa_mlp3_pool_compact: invalid argument (b=%d n=%d m=%d rows<=%ld)", b, n, m, max_rows);
        return WS3D_E_INVALID;
    }
    const long tiles = (max_rows + 31) / 32;
    // 3 workgroups per CU, waves walk over tiles; ws3d_tune key 2 (a caller with many batches in flight asks for 192: +2 % on the
    // 20-deep c3 step, profiles/r06_tune_workgroups.txt -- ws3d_amd/pipeline.py)
    const long cap1 = g_tune[TUNE_SA1_WGS] > 0 ? g_tune[TUNE_SA1_WGS] : 768;
    const unsigned grid = (unsigned)(tiles / 4 < cap1 ? (tiles + 3) / 4 : cap1);
#define WS3D_SA_COMPACT(A, B, C)                                                                                                        \
    if (c1 == A && c2 == B && c3 == C) {                                                                                                \
        hipLaunchKernelGGL((sa_mlp3_compact_mfma_kernel<A, B, C>), dim3(grid), dim3(256), 0, as_stream(stream), n, m, xyz, new_xyz, feat, rowc, \
                           rowsrc, total, w1t, b1, w2t, b2, w3t, b3, out, out_stride, limit);                                           \
        return check_launch("ws3d_sa_mlp3_pool_compact");                                                                               \
    }
    WS3D_SA_COMPACT(32, 32, 64)
    WS3D_SA_COMPACT(16, 16, 32)
#undef WS3D_SA_COMPACT
    set_error("ws3d_sa_mlp3_pool_compact: no kernel for widths (%d, %d, %d)", c1, c2, c3);
    return WS3D_E_UNSUPPORTED;
}

extern "C" int ws3d_sa_mlp3_pool_lists(int b, int n, int m, int nsample, int c1, int c2, int c3, const float *xyz, const float *new_xyz,
                                       const float *feat, const int32_t *nbr, const float *w1t, const float *b1, const float *w2t, const float *b2,
                                       const float *w3t, const float *b3, int relu3, float *out, int out_stride, const int32_t *gate, long gate_limit,
                                       ws3d_stream_t stream) {
    using namespace ws3d;
    const long rows = (long)b * m * nsample;
    if (b <= 0 || n <= 0 || m <= 0 || (nsample != 16 && nsample != 32) || (rows & 31) || !xyz || !new_xyz || !feat || !nbr || !w1t || !b1 || !w2t || !b2 ||
        !w3t || !b3 || !out || out_stride < c3) {
        set_error("ws3d_sa_mlp3_pool_lists: invalid argument (b=%d n=%d m=%d nsample=%d)", b, n, m, nsample);
        return WS3D_E_INVALID;
    }
    const long tiles = rows / 32;
    const long cap1 = g_tune[TUNE_SA1_WGS] > 0 ? g_tune[TUNE_SA1_WGS] : 768;          // (ws3d_tune key 2)
    const unsigned grid = (unsigned)(tiles / 4 < cap1 ? (tiles + 3) / 4 : cap1);
#define WS3D_SA_LISTS(A, B, C, N)                                                                                                       \
    if (c1 == A && c2 == B && c3 == C && nsample == N) {                                                                                \
        hipLaunchKernelGGL((sa_mlp3_lists_mfma_kernel<A, B, C, N>), dim3(grid), dim3(256), 0, as_stream(stream), tiles, n, m, xyz, new_xyz, feat, nbr, \
                           w1t, b1, w2t, b2, w3t, b3, relu3, out, out_stride, gate, gate_limit);                                        \
        return check_launch("ws3d_sa_mlp3_pool_lists");                                                                                 \
    }
    WS3D_SA_LISTS(32, 32, 64, 32)
    WS3D_SA_LISTS(16, 16, 32, 16)
    WS3D_SA_LISTS(32, 32, 64, 16)
    WS3D_SA_LISTS(16, 16, 32, 32)
#undef WS3D_SA_LISTS
    set_error("ws3d_sa_mlp3_pool_lists: no kernel for widths (%d, %d, %d) x nsample %d", c1, c2, c3, nsample);
    return WS3D_E_UNSUPPORTED;
}
