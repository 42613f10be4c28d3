// scatter_det.hip -- deterministic backward of the gather-type ops.
//
// The reference's gather_points_grad / group_points_grad / three_interpolate_grad
// (sampling_gpu.cu:49-76, group_points_gpu.cu:8-44, interpolate_gpu.cu:120-160) scatter with float
// atomicAdd: the summation order, and with it the low bits of every gradient, change from run to
// run.  Here the scatter is turned into a gather over the INVERSE index:
//   1. key = scene * n + idx[slot] for every slot (slot = position in the (m, s) / (n, 3) plane),
//      stable LSD radix sort of (key, slot) pairs (rocPRIM device radix sort -- a plain library
//      sort, like rocBLAS for the plain GEMMs); stability keeps the slots of one target ascending;
//   2. one lane per target point finds its segment by binary search and adds the contributions
//      in ascending slot order -- the order of the CPU loop `for slot: dst[idx[slot]] += v[slot]`,
//      so the result is bit-identical to that loop (oracle) and identical from run to run.
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

__device__ __forceinline__ long lower_bound_u32(const uint32_t *__restrict__ a, long len, uint32_t key) {
    long lo = 0, hi = len;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int DET_CCH = 8;  // channels per workgroup row: the segment is walked once per 8 channels

// grad_points[b, ch, t] = sum over the slots of target t, ascending, of
//   src[b, ch, slot / div] * (weight ? weight[b, slot] : 1)
// div = 1 (group / gather), 3 (three_interpolate: slot = point * 3 + k).
__global__ __launch_bounds__(256) void det_segment_sum_kernel(int c, int n, int plane, int div, long total,
                                                              const uint32_t *__restrict__ keys,
                                                              const uint32_t *__restrict__ slots,
                                                              const float *__restrict__ src,
                                                              const float *__restrict__ weight,
                                                              float *__restrict__ dst) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * DET_CCH;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const uint32_t key = (uint32_t)b * (uint32_t)n + (uint32_t)t;
    const long lo = lower_bound_u32(keys, total, key);
    const int src_len = plane / div;
    const float *s = src + ((size_t)b * c + c0) * src_len;
    const float *w = weight ? weight + (size_t)b * plane : nullptr;
    const int cc = min(DET_CCH, c - c0);
    float acc[DET_CCH];
#pragma unroll
    for (int q = 0; q < DET_CCH; ++q) acc[q] = 0.f;
    for (long i = lo; i < total && keys[i] == key; ++i) {
        const uint32_t slot = slots[i];
        const uint32_t from = div == 1 ? slot : slot / (uint32_t)div;
        const float wv = w ? w[slot] : 1.0f;
#pragma unroll
        for (int q = 0; q < DET_CCH; ++q)
            if (q < cc) {
                const float g = s[(size_t)q * src_len + from];
                acc[q] = acc[q] + (w ? g * wv : g);
            }
    }
    float *d = dst + ((size_t)b * c + c0) * n + t;
#pragma unroll
    for (int q = 0; q < DET_CCH; ++q)
        if (q < cc) d[(size_t)q * n] = acc[q];
}

struct DetLayout {
    uint32_t *keys_in, *keys_out, *slots_in, *slots_out;
    void *temp;
    size_t temp_bytes, total_bytes;
};

static int det_layout(long total, unsigned end_bit, void *workspace, DetLayout &L) {
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
    L.temp = p + 4 * arr;
    L.temp_bytes = temp;
    L.total_bytes = 4 * arr + ((temp + 255) & ~(size_t)255);
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
    if ((long)b * n > 0x7fffffffL || total > 0x7fffffffL || plane > 0x7fffffffL || b > 65535 ||
        (c + DET_CCH - 1) / DET_CCH > 65535) {
        set_error("%s: shape too large (b=%d n=%d plane=%ld)", what, b, n, plane);
        return WS3D_E_UNSUPPORTED;
    }
    if (total == 0) {
        (void)hipMemsetAsync(dst, 0, sizeof(float) * (size_t)b * c * n, st);
        return WS3D_OK;
    }
    DetLayout L;
    const unsigned end_bit = key_bits((long)b * n);
    int rc = det_layout(total, end_bit, workspace, L);
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
    hipLaunchKernelGGL(det_segment_sum_kernel, dim3((n + 255) / 256, (c + DET_CCH - 1) / DET_CCH, b), dim3(256), 0, st, c, n,
                       (int)plane, div, total, L.keys_out, L.slots_out, src, weight, dst);
    return check_launch(what);
}

}  // namespace ws3d

extern "C" size_t ws3d_scatter_workspace_bytes(int b, int n, long plane) {
    using namespace ws3d;
    if (b <= 0 || n <= 0 || plane <= 0 || (long)b * plane > 0x7fffffffL) return 256;
    DetLayout L;
    if (det_layout((long)b * plane, key_bits((long)b * n), nullptr, L) != WS3D_OK) return 0;
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
