// conv_wgrad.hip -- weight gradient of the 1x1 convolutions of the SharedMLP / Conv1d blocks (training):
//   dW[o, c] = sum_b sum_l dY[b, o, l] * X[b, c, l]          (pytorch_utils.py:35-101, nn.Conv1d/2d, kernel 1)
// The library's implicit-GEMM weight-gradient kernels want NHWC and split K with atomics: per training
// step 2.2 ms of kernels + 1.5 ms of NCHW<->NHWC transposes + ~130 small copies, and the atomics are why
// the step is not reproducible run to run.  Here both operands are read where they lie (channels-first:
// contiguous along l), a 64 x 64 tile of dW per workgroup is accumulated on the matrix cores
// (v_mfma_f32_32x32x2_f32, fp32 in / fp32 accumulate) over one (scene, l-range) slice of K, the slices'
// partial tiles go to a workspace and a second kernel adds them in a FIXED order: deterministic.
#include "common.h"

namespace ws3d {

typedef float floatx16 __attribute__((ext_vector_type(16)));

// ROWS rows of each operand are staged per tile (64, or 16 for layers with <= 16 channels on both sides),
// KT values of l per row: 128 / 256 contiguous bytes per row and tile -- with 64-byte pieces (KT = 16) the
// kernel ran at 1.1 TB/s on the early layers, whose l extent is 65536..131072.
// grid (ceil(c/ROWS), ceil(o/ROWS), b * lsplit); partial[(z * o + oo) * c + cc]
template <int KT, int ROWS>
__global__ __launch_bounds__(256) void conv_wgrad_partial_kernel(int o_dim, int c_dim, long l_dim, int lsplit, long l_chunk,
                                                                 const float *__restrict__ dy, const float *__restrict__ x,
                                                                 float *__restrict__ partial) {
    constexpr int RL = ROWS == 64 ? 65 : 33;             // padded row length; the MFMA tile reads rows 0..31 at least
    constexpr int F4 = KT / 4;                            // float4 per staged row
    constexpr int NL = (ROWS * F4 + 255) / 256;           // staging loads per thread and operand
    __shared__ float as_[2][KT][RL];                      // [l][o]
    __shared__ float bs_[2][KT][RL];                      // [l][c]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w & 1, wn = w >> 1;
    const int c0 = blockIdx.x * ROWS, o0 = blockIdx.y * ROWS;
    const int b = blockIdx.z / lsplit, sp = blockIdx.z - b * lsplit;
    const long l_lo = (long)sp * l_chunk, l_hi = min(l_dim, l_lo + l_chunk);
    if (ROWS < 32) {                                      // rows ROWS..31 of the tiles stay zero
        for (int i = tid; i < 2 * KT * RL; i += 256) { (&as_[0][0][0])[i] = 0.f; (&bs_[0][0][0])[i] = 0.f; }
        __syncthreads();
    }
    const bool vec = (l_dim & 3) == 0 && (((uintptr_t)dy | (uintptr_t)x) & 15) == 0;
    const float *ab = dy + (size_t)b * o_dim * l_dim, *bb = x + (size_t)b * c_dim * l_dim;
    auto load = [&](const float *base, int row0, int dim, int slot, long l_tile) {
        const int r = slot / F4;
        const long l = l_tile + (slot % F4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= ROWS || row0 + r >= dim) return v;
        const float *p = base + (size_t)(row0 + r) * l_dim;
        if (vec && l + 3 < l_hi) return *reinterpret_cast<const float4 *>(p + l);
        if (l < l_hi) v.x = p[l];
        if (l + 1 < l_hi) v.y = p[l + 1];
        if (l + 2 < l_hi) v.z = p[l + 2];
        if (l + 3 < l_hi) v.w = p[l + 3];
        return v;
    };
    auto put = [&](float (*dst)[RL], int slot, const float4 v) {
        const int r = slot / F4, lk = (slot % F4) * 4;
        if (r < ROWS) { dst[lk + 0][r] = v.x; dst[lk + 1][r] = v.y; dst[lk + 2][r] = v.z; dst[lk + 3][r] = v.w; }
    };
    floatx16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const long ntiles = (l_hi - l_lo + KT - 1) / KT;
    const bool mfma_wave = ROWS == 64 || w == 0;          // 16-row tiles: one 32 x 32 MFMA tile covers them
    if (ntiles > 0) {
        float4 av[NL], bv[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) { av[j] = load(ab, o0, o_dim, tid + 256 * j, l_lo); bv[j] = load(bb, c0, c_dim, tid + 256 * j, l_lo); }
#pragma unroll
        for (int j = 0; j < NL; ++j) { put(as_[0], tid + 256 * j, av[j]); put(bs_[0], tid + 256 * j, bv[j]); }
        __syncthreads();
        const int ar = (ROWS == 64 ? wm * 32 : 0) + (lane & 31), bc = (ROWS == 64 ? wn * 32 : 0) + (lane & 31), kh = lane >> 5;
        for (long t = 0; t < ntiles; ++t) {
            const int cur = (int)(t & 1);
            if (t + 1 < ntiles) {
#pragma unroll
                for (int j = 0; j < NL; ++j) {
                    av[j] = load(ab, o0, o_dim, tid + 256 * j, l_lo + (t + 1) * KT);
                    bv[j] = load(bb, c0, c_dim, tid + 256 * j, l_lo + (t + 1) * KT);
                }
            }
            if (mfma_wave) {
#pragma unroll
                for (int k = 0; k < KT; k += 2)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(as_[cur][k + kh][ar], bs_[cur][k + kh][bc], acc, 0, 0, 0);
            }
            if (t + 1 < ntiles) {
#pragma unroll
                for (int j = 0; j < NL; ++j) { put(as_[cur ^ 1], tid + 256 * j, av[j]); put(bs_[cur ^ 1], tid + 256 * j, bv[j]); }
            }
            __syncthreads();
        }
    }
    // register v of lane l holds row (o) 8*(v/4) + 4*(l/32) + v%4, column (c) l%32 of the wave's 32 x 32 sub-tile
    if (!mfma_wave) return;
    float *pt = partial + (size_t)blockIdx.z * o_dim * c_dim;
    const int cc = c0 + (ROWS == 64 ? wn * 32 : 0) + (lane & 31);
    const bool c_ok = cc < c_dim && (ROWS == 64 || (lane & 31) < ROWS);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int ro = 8 * (v / 4) + 4 * (lane >> 5) + (v & 3);
        const int oo = o0 + (ROWS == 64 ? wm * 32 : 0) + ro;
        if (oo < o_dim && c_ok && (ROWS == 64 || ro < ROWS)) pt[(size_t)oo * c_dim + cc] = acc[v];
    }
}

// dW[i] = sum over the slices of partial[z][i], in a FIXED order: 64 lanes per element each add their
// slices z = lane, lane + 64, ... sequentially, then a fixed tree over the 64 lane sums (a thread per
// element walking 1024 slices alone is a 1024-deep chain of dependent loads: 0.3 ms on the early layers)
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(long elems, int slices, const float *__restrict__ partial,
                                                                float *__restrict__ dw) {
    const int lane = threadIdx.x & 63;
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per element
    if (i >= elems) return;
    float s = 0.f;
    for (int z = lane; z < slices; z += 64) s = s + partial[(size_t)z * elems + i];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s = s + __shfl_xor(s, off);
    if (lane == 0) dw[i] = s;
}

// few slices (deep layers: many tiles, short l): a thread per element adds them in slice order
__global__ __launch_bounds__(256) void conv_wgrad_reduce_few_kernel(long elems, int slices, const float *__restrict__ partial,
                                                                    float *__restrict__ dw) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    float s = 0.f;
    for (int z = 0; z < slices; ++z) s = s + partial[(size_t)z * elems + i];
    dw[i] = s;
}

static int wgrad_rows(int o, int c) { return (o <= 16 && c <= 16) ? 16 : 64; }

static int wgrad_lsplit(int b, int o, int c, long l) {
    const int rows = wgrad_rows(o, c);
    const long tiles = (long)((o + rows - 1) / rows) * ((c + rows - 1) / rows) * b;
    int sp = 1;
    while (tiles * sp < 1024 && l / (sp * 2) >= 256 && sp < 256) sp *= 2;    // >= 256 l per slice, ~1024 workgroups
    return sp;
}

}  // namespace ws3d

extern "C" size_t ws3d_conv1x1_wgrad_workspace_bytes(int b, int o, int c, long l) {
    if (b <= 0 || o <= 0 || c <= 0 || l <= 0) return 256;
    return (size_t)b * ws3d::wgrad_lsplit(b, o, c, l) * (size_t)o * c * sizeof(float) + 256;
}

extern "C" int ws3d_conv1x1_wgrad(int b, int o, int c, long l, const float *grad_out, const float *x, float *grad_w,
                                  void *workspace, size_t workspace_bytes, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b <= 0 || o <= 0 || c <= 0 || l <= 0 || !grad_out || !x || !grad_w || !workspace) {
        set_error("ws3d_conv1x1_wgrad: invalid argument (b=%d o=%d c=%d l=%ld)", b, o, c, l);
        return WS3D_E_INVALID;
    }
    const int sp = wgrad_lsplit(b, o, c, l);
    const size_t need = (size_t)b * sp * (size_t)o * c * sizeof(float);
    if (workspace_bytes < need || (long)b * sp > 65535 || (reinterpret_cast<uintptr_t>(workspace) & 3)) {
        set_error("ws3d_conv1x1_wgrad: workspace too small (%zu < %zu) or too many slices", workspace_bytes, need);
        return WS3D_E_WORKSPACE;
    }
    hipStream_t st = as_stream(stream);
    const long l_chunk = ((l + sp - 1) / sp + 3) & ~3L;
    float *partial = reinterpret_cast<float *>(workspace);
    if (wgrad_rows(o, c) == 16)
        hipLaunchKernelGGL((conv_wgrad_partial_kernel<64, 16>), dim3((c + 15) / 16, (o + 15) / 16, b * sp), dim3(256), 0, st, o, c, l, sp,
                           l_chunk, grad_out, x, partial);
    else
        hipLaunchKernelGGL((conv_wgrad_partial_kernel<32, 64>), dim3((c + 63) / 64, (o + 63) / 64, b * sp), dim3(256), 0, st, o, c, l, sp,
                           l_chunk, grad_out, x, partial);
    const long elems = (long)o * c;
    if (b * sp > 32)
        hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((elems + 3) / 4)), dim3(256), 0, st, elems, b * sp, partial, grad_w);
    else
        hipLaunchKernelGGL(conv_wgrad_reduce_few_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, st, elems, b * sp, partial,
                           grad_w);
    return check_launch("ws3d_conv1x1_wgrad");
}
