// scatter_det.hip -- deterministic backward of the gather-type ops.
//
// The reference's gather_points_grad / group_points_grad / three_interpolate_grad
// (sampling_gpu.cu:49-76, group_points_gpu.cu:8-44, interpolate_gpu.cu:120-160) scatter with float
// atomicAdd: the summation order, and with it the low bits of every gradient, change from run to
// run.  Here the scatter is turned into a gather over the INVERSE index:
//   1. key = scene * n + idx[slot] for every slot (slot = position in the (m, s) / (n, 3) plane),
//      stable LSD radix sort of (key, slot) pairs (rocPRIM device radix sort -- a plain library
//      sort, like rocBLAS for the plain GEMMs); stability keeps the slots of one target ascending;
//   2. segment bounds [lo, hi) of every target from one pass over the sorted keys;
//   3. one WAVE per target, lanes over channels: every lane adds the contributions of its channels
//      in ascending slot order -- the order of the CPU loop `for slot: dst[idx[slot]] += v[slot]`,
//      so the result is bit-identical to that loop (oracle) and identical from run to run.  All
//      lanes of a wave walk the same segment (no divergence: ball-query padding makes segment
//      lengths very uneven), the slot ids are fetched 64 at a time and broadcast by readlane.
// The sorted inverse index depends only on idx and is shared by all channels of a call.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace ws3d {

__global__ __launch_bounds__(256) void det_keys_kernel(long total, int plane, int n,
                                                       const int32_t *__restrict__ idx,
                                                       uint32_t *__restrict__ keys, uint32_t *__restrict__ slots) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int b = (int)(i / plane);
    keys[i] = (uint32_t)b * (uint32_t)n + (uint32_t)idx[i];
    slots[i] = (uint32_t)(i - (long)b * plane);
}

// segment bounds: lo[key] / hi[key] = first / one-past-last sorted position of key (both 0 if absent)
__global__ __launch_bounds__(256) void det_bounds_kernel(long total, const uint32_t *__restrict__ keys,
                                                         uint32_t *__restrict__ lo, uint32_t *__restrict__ hi) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const uint32_t k = keys[i];
    if (i == 0 || keys[i - 1] != k) lo[k] = (uint32_t)i;
    if (i == total - 1 || keys[i + 1] != k) hi[k] = (uint32_t)(i + 1);
}

constexpr int DET_WAVES = 4;   // targets per workgroup (one per wave)

// grad_points[b, ch, t] = sum over the slots of target t, ascending, of
//   src[b, ch, slot / div] * (weight ? weight[b, slot] : 1)
// div = 1 (group / gather), 3 (three_interpolate: slot = point * 3 + k).
// One wave per target; lane l owns channels c0 + l, c0 + 64 + l, ... (KC of them per pass).
// ROWS: src was transposed to (b, slots/div, c) first, so the 64 lanes read 256 contiguous bytes per
// slot; without it every lane touches its own 64-byte sector of a channels-first row (16x the bytes
// through L2 -- measured 1.6 ms instead of 0.2 ms for the FP1 backward, c=256, 49152 slots/scene).
template <int KC, bool ROWS>
__global__ __launch_bounds__(64 * DET_WAVES) void det_wave_sum_kernel(int c, int n, int plane, int div, long targets,
                                                                     const uint32_t *__restrict__ seg_lo,
                                                                     const uint32_t *__restrict__ seg_hi,
                                                                     const uint32_t *__restrict__ slots,
                                                                     const float *__restrict__ src,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const long target = (long)blockIdx.x * DET_WAVES + (threadIdx.x >> 6);
    if (target >= targets) return;
    const int b = (int)(target / n);
    const int t = (int)(target - (long)b * n);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(seg_lo[target]);
    const uint32_t hi = __builtin_amdgcn_readfirstlane(seg_hi[target]);
    const int src_len = plane / div;
    const float *w = weight ? weight + (size_t)b * plane : nullptr;
    for (int c0 = 0; c0 < c; c0 += 64 * KC) {
        const float *s[KC];
        float acc[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ch = min(c0 + k * 64 + lane, c - 1);     // clamped lanes compute a duplicate, never stored
            s[k] = ROWS ? src + (size_t)b * src_len * c + ch : src + ((size_t)b * c + ch) * src_len;
            acc[k] = 0.f;
        }
        for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
            const int cnt = (int)min(64u, hi - i0);
            const uint32_t my_slot = lane < cnt ? slots[i0 + lane] : 0u;
            const float my_w = (w && lane < cnt) ? w[my_slot] : 1.0f;
            const uint32_t my_from = div == 1 ? my_slot : my_slot / (uint32_t)div;
#pragma unroll 4
            for (int j = 0; j < cnt; ++j) {
                const uint32_t from = (uint32_t)__builtin_amdgcn_readlane((int)my_from, j);
                const float wv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_w), j));
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const float g = ROWS ? s[k][(size_t)from * c] : s[k][from];
                    acc[k] = acc[k] + (w ? g * wv : g);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int ch = c0 + k * 64 + lane;
            if (ch < c) dst[((size_t)b * c + ch) * n + t] = acc[k];
        }
    }
}

// (b, c, len) -> (b, len, c) through a 64 x 64 LDS tile, both sides coalesced
__global__ __launch_bounds__(256) void det_to_rows_kernel(int c, int len, const float *__restrict__ src,
                                                          float *__restrict__ dst) {
    __shared__ float tile[64][65];
    const int b = blockIdx.z, c0 = blockIdx.y * 64, l0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {
        const int ch = c0 + r, col = l0 + tx;
        tile[r][tx] = (ch < c && col < len) ? src[((size_t)b * c + ch) * len + col] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {
        const int col = l0 + r, ch = c0 + tx;
        if (ch < c && col < len) dst[((size_t)b * len + col) * c + ch] = tile[tx][r];
    }
}

constexpr int DET_ROWS_MIN_C = 32;   // narrower inputs are summed from the channels-first layout directly

struct DetLayout {
    uint32_t *keys_in, *keys_out, *slots_in, *slots_out, *seg_lo, *seg_hi;
    float *rows;                 // (b, src_len, c) copy of src, nullptr when c < DET_ROWS_MIN_C
    size_t seg_bytes;
    void *temp;
    size_t temp_bytes, total_bytes;
};

static int det_layout(long total, long targets, size_t rows_floats, unsigned end_bit, void *workspace, DetLayout &L) {
    size_t temp = 0;
    const hipError_t e = rocprim::radix_sort_pairs(nullptr, temp, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                                   (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)total, 0u, end_bit);
    if (e != hipSuccess) return WS3D_E_LAUNCH;
    const size_t arr = (((size_t)total * sizeof(uint32_t)) + 255) & ~(size_t)255;
    char *p = reinterpret_cast<char *>(workspace);
    L.keys_in = reinterpret_cast<uint32_t *>(p);
    L.keys_out = reinterpret_cast<uint32_t *>(p + arr);
    L.slots_in = reinterpret_cast<uint32_t *>(p + 2 * arr);
    L.slots_out = reinterpret_cast<uint32_t *>(p + 3 * arr);
    L.seg_bytes = (((size_t)targets * sizeof(uint32_t)) + 255) & ~(size_t)255;
    L.seg_lo = reinterpret_cast<uint32_t *>(p + 4 * arr);                 // seg_lo and seg_hi are adjacent:
    L.seg_hi = reinterpret_cast<uint32_t *>(p + 4 * arr + L.seg_bytes);   // one memset clears both
    L.temp = p + 4 * arr + 2 * L.seg_bytes;
    L.temp_bytes = temp;
    const size_t head = 4 * arr + 2 * L.seg_bytes + ((temp + 255) & ~(size_t)255);
    L.rows = rows_floats ? reinterpret_cast<float *>(p + head) : nullptr;
    L.total_bytes = head + ((rows_floats * sizeof(float) + 255) & ~(size_t)255);
    return WS3D_OK;
}

static unsigned key_bits(long span) {  // bits needed for keys in [0, span)
    unsigned bits = 1;
    while (bits < 32 && (1L << bits) < span) ++bits;
    return bits;
}

static int det_scatter(int b, int c, int n, long plane, int div, const float *src, const int32_t *idx,
                       const float *weight, float *dst, void *workspace, size_t workspace_bytes, hipStream_t st,
                       const char *what) {
    if (b < 0 || c < 0 || n <= 0 || plane < 0 || ((!src || !idx) && plane > 0) || !dst) {
        set_error("%s: invalid argument (b=%d c=%d n=%d plane=%ld)", what, b, c, n, plane);
        return WS3D_E_INVALID;
    }
    if (b == 0 || c == 0) return WS3D_OK;
    const long total = (long)b * plane;
    if ((long)b * n > 0x7fffffffL || total > 0x7fffffffL || plane > 0x7fffffffL) {
        set_error("%s: shape too large (b=%d n=%d plane=%ld)", what, b, n, plane);
        return WS3D_E_UNSUPPORTED;
    }
    if (total == 0) {
        (void)hipMemsetAsync(dst, 0, sizeof(float) * (size_t)b * c * n, st);
        return WS3D_OK;
    }
    DetLayout L;
    const unsigned end_bit = key_bits((long)b * n);
    const int src_len = (int)(plane / div);
    const size_t rows_floats = c >= DET_ROWS_MIN_C ? (size_t)b * c * src_len : 0;
    int rc = det_layout(total, (long)b * n, rows_floats, end_bit, workspace, L);
    if (rc != WS3D_OK) { set_error("%s: rocprim size query failed", what); return rc; }
    if (!workspace || workspace_bytes < L.total_bytes) {
        set_error("%s: workspace too small (%zu < %zu)", what, workspace_bytes, L.total_bytes);
        return WS3D_E_WORKSPACE;
    }
    hipLaunchKernelGGL(det_keys_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, (int)plane, n, idx,
                       L.keys_in, L.slots_in);
    size_t temp = L.temp_bytes;
    const hipError_t e = rocprim::radix_sort_pairs(L.temp, temp, L.keys_in, L.keys_out, L.slots_in, L.slots_out,
                                                   (size_t)total, 0u, end_bit, st);
    if (e != hipSuccess) { set_error("%s: radix sort failed: %s", what, hipGetErrorString(e)); return WS3D_E_LAUNCH; }
    const long targets = (long)b * n;
    (void)hipMemsetAsync(L.seg_lo, 0, 2 * L.seg_bytes, st);
    hipLaunchKernelGGL(det_bounds_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, total, L.keys_out, L.seg_lo,
                       L.seg_hi);
    const dim3 grid((unsigned)((targets + DET_WAVES - 1) / DET_WAVES)), block(64 * DET_WAVES);
#define WS3D_DET_LAUNCH(KC, ROWS, SRC)                                                                                  \
    hipLaunchKernelGGL((det_wave_sum_kernel<KC, ROWS>), grid, block, 0, st, c, n, (int)plane, div, targets, L.seg_lo,   \
                       L.seg_hi, L.slots_out, SRC, weight, dst)
    if (L.rows) {
        if (b > 65535 || (c + 63) / 64 > 65535) { set_error("%s: shape too large (b=%d c=%d)", what, b, c); return WS3D_E_UNSUPPORTED; }
        hipLaunchKernelGGL(det_to_rows_kernel, dim3((src_len + 63) / 64, (c + 63) / 64, b), dim3(256), 0, st, c, src_len, src,
                           L.rows);
        if (c <= 64) WS3D_DET_LAUNCH(1, true, L.rows);
        else if (c <= 128) WS3D_DET_LAUNCH(2, true, L.rows);
        else if (c <= 256) WS3D_DET_LAUNCH(4, true, L.rows);
        else WS3D_DET_LAUNCH(8, true, L.rows);
    } else {
        WS3D_DET_LAUNCH(1, false, src);
    }
#undef WS3D_DET_LAUNCH
    return check_launch(what);
}

}  // namespace ws3d

extern "C" size_t ws3d_scatter_workspace_bytes(int b, int c, int n, long plane) {
    using namespace ws3d;
    if (b <= 0 || c <= 0 || n <= 0 || plane <= 0 || (long)b * plane > 0x7fffffffL) return 256;
    DetLayout L;
    // the (b, slots, c) staging copy is sized for group/gather (div = 1), which covers div = 3 too
    const size_t rows_floats = c >= DET_ROWS_MIN_C ? (size_t)b * c * (size_t)plane : 0;
    if (det_layout((long)b * plane, (long)b * n, rows_floats, key_bits((long)b * n), nullptr, L) != WS3D_OK) return 0;
    return L.total_bytes;
}

extern "C" int ws3d_group_points_grad_det(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                          const int32_t *idx, float *grad_points, void *workspace,
                                          size_t workspace_bytes, ws3d_stream_t stream) {
    if (npoints < 0 || nsample < 0) { ws3d::set_error("ws3d_group_points_grad_det: negative shape"); return WS3D_E_INVALID; }
    return ws3d::det_scatter(b, c, n, (long)npoints * nsample, 1, grad_out, idx, nullptr, grad_points, workspace,
                             workspace_bytes, ws3d::as_stream(stream), "ws3d_group_points_grad_det");
}

extern "C" int ws3d_three_interpolate_grad_det(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                                               const float *weight, float *grad_points, void *workspace,
                                               size_t workspace_bytes, ws3d_stream_t stream) {
    if (!weight || n < 0) { ws3d::set_error("ws3d_three_interpolate_grad_det: invalid argument"); return WS3D_E_INVALID; }
    return ws3d::det_scatter(b, c, m, (long)n * 3, 3, grad_out, idx, weight, grad_points, workspace, workspace_bytes,
                             ws3d::as_stream(stream), "ws3d_three_interpolate_grad_det");
}
