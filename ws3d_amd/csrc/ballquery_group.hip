// ballquery_group.hip -- ball query, grouping, gather and the fused QueryAndGroup for
// gfx950.  Replaces pointnet2_cuda.{ball_query_wrapper, group_points_wrapper,
// group_points_grad_wrapper, gather_points_wrapper, gather_points_grad_wrapper}
// (ball_query_gpu.cu:9-67, group_points_gpu.cu:8-86, sampling_gpu.cu:8-76) and the
// Python composition QueryAndGroup.forward (pointnet2_utils.py:241-264).
//
// Design.  Irregular gather/scatter: no MFMA.  One lane per
// query centre; the scene's points stream through LDS in float4 tiles (every lane of a
// wave reads the SAME LDS address = broadcast, conflict-free), so each point is read
// from global memory once per wave of 64 centres instead of once per centre.  A wave-uniform
// one-axis reject (|dx| >= r  =>  d2 >= r*r, exact in fp32 because rounding is
// monotone) skips the distance for points no lane can accept.  Neighbour lists are
// built in LDS rows (stride nsample+1: conflict-free per-lane appends) in ascending
// point order -- first-nsample-by-index, pad-with-first-hit and no-hit rules are part
// of the contract -- and leave the chip with coalesced stores.  In the fused kernel the
// rows never go back to global memory: the same workgroup emits the centred xyz and
// the feature channels straight into the (B, 3+C, M, ns) tensor the SharedMLP consumes.
#include "common.h"

namespace ws3d {

constexpr int BQ_TILE = 256;  // points per wave-private LDS tile
constexpr int BQ_NW = 4;      // waves per workgroup: they split the scene's point range

// One workgroup = 64 query centres (one per lane) x BQ_NW waves.  Wave w scans the w-th
// quarter of the scene through its OWN LDS tile (no workgroup barrier inside the scan) and
// appends hits to its own LDS rows; quarters are ordered, so concatenating the four rows
// gives the ascending-index neighbour list the reference's serial scan produces.
template <typename IDX, bool FUSED>
__global__ __launch_bounds__(64 * BQ_NW) void ball_query_kernel(int n, int m, int c_feat, float radius,
                                                                int nsample, int use_xyz,
                                                                const float *__restrict__ xyz,
                                                                const float *__restrict__ new_xyz,
                                                                const float *__restrict__ features,
                                                                int32_t *__restrict__ idx_out,
                                                                float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *tiles = reinterpret_cast<float4 *>(smem);                       // BQ_NW * BQ_TILE
    float4 *cen = tiles + BQ_NW * BQ_TILE;                                  // 64
    int *cnt_s = reinterpret_cast<int *>(cen + 64);                         // BQ_NW * 64
    IDX *rows = reinterpret_cast<IDX *>(cnt_s + BQ_NW * 64);                // BQ_NW * 64 * rstride
    const int rstride = nsample + 1;                                        // odd stride: conflict-free appends

    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.x * 64;
    const int mi = m0 + lane;
    const bool active = mi < m;
    xyz += (size_t)b * n * 3;
    new_xyz += (size_t)b * m * 3;

    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (active) { cx = new_xyz[mi * 3 + 0]; cy = new_xyz[mi * 3 + 1]; cz = new_xyz[mi * 3 + 2]; }
    if (w == 0) cen[lane] = make_float4(cx, cy, cz, 0.f);
    const float radius2 = radius * radius;
    const float rabs = fabsf(radius);
    float4 *tile = tiles + w * BQ_TILE;
    IDX *row = rows + (size_t)(w * 64 + lane) * rstride;
    int cnt = active ? 0 : nsample;  // inactive lanes are "full": they never append

    const int Q = ((n + BQ_NW - 1) / BQ_NW + BQ_TILE - 1) / BQ_TILE * BQ_TILE;
    const int start = min(w * Q, n), end = min(start + Q, n);
    for (int base = start; base < end; base += BQ_TILE) {
        const int lim = min(BQ_TILE, end - base);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // previous tile fully consumed (wave-local)
#pragma unroll
        for (int q = 0; q < BQ_TILE / 64; ++q) {
            const int i = lane + 64 * q;
            if (i < lim) {
                const float *p = xyz + (size_t)(base + i) * 3;
                tile[i] = make_float4(p[0], p[1], p[2], 0.f);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // tile visible to every lane of this wave
        auto visit = [&](const float4 p, const int k, const bool near) {
            if (near && cnt < nsample) {
                const float d2 = sqdist3(cx - p.x, cy - p.y, cz - p.z);
                if (d2 < radius2) { row[cnt] = (IDX)k; ++cnt; }
            }
        };
        int i = 0;
        for (; i + 4 <= lim; i += 4) {
            const float4 p0 = tile[i], p1 = tile[i + 1], p2 = tile[i + 2], p3 = tile[i + 3];
            const bool a0 = fabsf(cx - p0.x) < rabs, a1 = fabsf(cx - p1.x) < rabs;
            const bool a2 = fabsf(cx - p2.x) < rabs, a3 = fabsf(cx - p3.x) < rabs;
            if (__any((a0 | a1 | a2 | a3) & (cnt < nsample))) {
                visit(p0, base + i, a0);
                visit(p1, base + i + 1, a1);
                visit(p2, base + i + 2, a2);
                visit(p3, base + i + 3, a3);
            }
        }
        for (; i < lim; ++i) {
            const float4 p = tile[i];
            visit(p, base + i, fabsf(cx - p.x) < rabs);
        }
        if (__all(cnt >= nsample)) break;
    }
    cnt_s[w * 64 + lane] = active ? cnt : 0;
    __syncthreads();

    // concatenate the per-wave rows into wave 0's row (ascending index), truncate to nsample,
    // pad with the first hit; a centre without any hit groups index 0 / leaves idx untouched
    if (w == 0 && active) {
        int total = cnt;  // own (wave 0) hits are already in place
#pragma unroll
        for (int ww = 1; ww < BQ_NW; ++ww) {
            const int cw = cnt_s[ww * 64 + lane];
            const IDX *src = rows + (size_t)(ww * 64 + lane) * rstride;
            for (int s = 0; s < cw && total < nsample; ++s) row[total++] = src[s];
        }
        const IDX first = total > 0 ? row[0] : (IDX)0;
        for (int s = total; s < nsample; ++s) row[s] = first;
        cnt_s[lane] = total;
    }
    __syncthreads();

    constexpr int NT = 64 * BQ_NW;
    const int total_e = 64 * nsample;
    if (!FUSED) {
        // ball_query contract: rows without any hit are left untouched (ball_query_gpu.cu:29-44)
        int32_t *o = idx_out + ((size_t)b * m + m0) * nsample;
        for (int e = tid; e < total_e; e += NT) {
            const int c = e / nsample, s = e - c * nsample;
            if (m0 + c < m && cnt_s[c] > 0) o[e] = (int32_t)rows[(size_t)c * rstride + s];
        }
        return;
    }
    if (idx_out) {
        int32_t *o = idx_out + ((size_t)b * m + m0) * nsample;
        for (int e = tid; e < total_e; e += NT) {
            const int c = e / nsample, s = e - c * nsample;
            if (m0 + c < m) o[e] = (int32_t)rows[(size_t)c * rstride + s];
        }
    }
    const int c_xyz = use_xyz ? 3 : 0;
    const int c_out = c_xyz + c_feat;
    const size_t plane = (size_t)m * nsample;
    float *ob = out + (size_t)b * c_out * plane + (size_t)m0 * nsample;
    const float *fb = features ? features + (size_t)b * c_feat * n : nullptr;
    for (int e = tid; e < total_e; e += NT) {
        const int c = e / nsample, s = e - c * nsample;
        if (m0 + c >= m) continue;
        const int id = (int)rows[(size_t)c * rstride + s];
        if (use_xyz) {
            const float4 ce = cen[c];
            const float *p = xyz + (size_t)id * 3;
            ob[e] = p[0] - ce.x;                 // grouped_xyz -= new_xyz (pointnet2_utils.py:252)
            ob[plane + e] = p[1] - ce.y;
            ob[2 * plane + e] = p[2] - ce.z;
        }
        for (int ch = 0; ch < c_feat; ++ch) ob[(size_t)(c_xyz + ch) * plane + e] = fb[(size_t)ch * n + id];
    }
}

static size_t bq_smem(int nsample, size_t idx_bytes) {
    return sizeof(float4) * (BQ_NW * BQ_TILE + 64) + sizeof(int) * BQ_NW * 64 +
           idx_bytes * (size_t)BQ_NW * 64 * (nsample + 1);
}

template <bool FUSED>
static int bq_launch(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                     const float *xyz, const float *new_xyz, const float *features, int32_t *idx,
                     float *out, hipStream_t st, const char *what) {
    if (b < 0 || n <= 0 || m < 0 || nsample <= 0 || c < 0 || !xyz || !new_xyz) {
        set_error("%s: invalid argument (b=%d n=%d m=%d nsample=%d c=%d)", what, b, n, m, nsample, c);
        return WS3D_E_INVALID;
    }
    if (!FUSED && !idx) { set_error("%s: idx is NULL", what); return WS3D_E_INVALID; }
    if (FUSED && (!out || (c > 0 && !features) || (!use_xyz && c == 0))) {
        set_error("%s: out/features NULL or no channels", what);
        return WS3D_E_INVALID;
    }
    if (b == 0 || m == 0) return WS3D_OK;
    const bool small_idx = n <= 65536;  // neighbour lists held as uint16 in LDS
    const size_t smem = bq_smem(nsample, small_idx ? 2 : 4);
    if (smem > 160 * 1024 || b > 65535) {
        set_error("%s: nsample=%d needs %zu B of LDS (> 160 KiB) or batch > 65535", what, nsample, smem);
        return WS3D_E_UNSUPPORTED;
    }
    dim3 grid((m + 63) / 64, b);
    if (small_idx) {
        if (smem > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)ball_query_kernel<uint16_t, FUSED>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((ball_query_kernel<uint16_t, FUSED>), grid, dim3(64 * BQ_NW), smem, st, n, m, c,
                           radius, nsample, use_xyz, xyz, new_xyz, features, idx, out);
    } else {
        if (smem > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)ball_query_kernel<int32_t, FUSED>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((ball_query_kernel<int32_t, FUSED>), grid, dim3(64 * BQ_NW), smem, st, n, m, c,
                           radius, nsample, use_xyz, xyz, new_xyz, features, idx, out);
    }
    return check_launch(what);
}

// ---- grouping_operation / gather_operation (gather == grouping with nsample == 1) ----
constexpr int GRP_CCH = 8;  // channels per workgroup: idx is read once per 8 channels

__global__ __launch_bounds__(256) void group_points_kernel(int c, int n, int plane,
                                                           const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx,
                                                           float *__restrict__ out) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * GRP_CCH;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= plane) return;
    const int id = idx[(size_t)b * plane + e];
    const float *src = points + ((size_t)b * c + c0) * n + id;
    float *dst = out + ((size_t)b * c + c0) * plane + e;
    const int cc = min(GRP_CCH, c - c0);
    for (int ch = 0; ch < cc; ++ch) dst[(size_t)ch * plane] = src[(size_t)ch * n];
}

__global__ __launch_bounds__(256) void group_points_grad_kernel(int c, int n, int plane,
                                                                const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * GRP_CCH;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= plane) return;
    const int id = idx[(size_t)b * plane + e];
    float *dst = grad_points + ((size_t)b * c + c0) * n + id;
    const float *src = grad_out + ((size_t)b * c + c0) * plane + e;
    const int cc = min(GRP_CCH, c - c0);
    for (int ch = 0; ch < cc; ++ch) atomicAdd(dst + (size_t)ch * n, src[(size_t)ch * plane]);
}

static int group_launch(bool grad, int b, int c, int n, int npoints, int nsample, const float *src,
                        const int32_t *idx, float *dst, hipStream_t st, const char *what) {
    if (b < 0 || c < 0 || n <= 0 || npoints < 0 || nsample < 0 || !src || !idx || !dst) {
        set_error("%s: invalid argument (b=%d c=%d n=%d npoints=%d nsample=%d)", what, b, c, n, npoints, nsample);
        return WS3D_E_INVALID;
    }
    const long plane = (long)npoints * nsample;
    if (b == 0 || c == 0 || plane == 0) return WS3D_OK;
    if (plane > 0x7fffffffL || (c + GRP_CCH - 1) / GRP_CCH > 65535 || b > 65535) {
        set_error("%s: shape too large for one launch", what);
        return WS3D_E_UNSUPPORTED;
    }
    dim3 grid((unsigned)((plane + 255) / 256), (c + GRP_CCH - 1) / GRP_CCH, b);
    if (grad)
        hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(256), 0, st, c, n, (int)plane, src, idx, dst);
    else
        hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, st, c, n, (int)plane, src, idx, dst);
    return check_launch(what);
}

}  // namespace ws3d

extern "C" int ws3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                               const float *xyz, int32_t *idx, ws3d_stream_t stream) {
    return ws3d::bq_launch<false>(b, n, m, 0, radius, nsample, 0, xyz, new_xyz, nullptr, idx, nullptr,
                                  ws3d::as_stream(stream), "ws3d_ball_query");
}

extern "C" int ws3d_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                                    int use_xyz, const float *xyz, const float *new_xyz,
                                    const float *features, int32_t *idx_out, float *out,
                                    ws3d_stream_t stream) {
    return ws3d::bq_launch<true>(b, n, m, features ? c : 0, radius, nsample, use_xyz, xyz, new_xyz,
                                 features, idx_out, out, ws3d::as_stream(stream), "ws3d_query_and_group");
}

extern "C" int ws3d_group_points(int b, int c, int n, int npoints, int nsample, const float *points,
                                 const int32_t *idx, float *out, ws3d_stream_t stream) {
    return ws3d::group_launch(false, b, c, n, npoints, nsample, points, idx, out,
                              ws3d::as_stream(stream), "ws3d_group_points");
}

extern "C" int ws3d_group_points_grad(int b, int c, int n, int npoints, int nsample,
                                      const float *grad_out, const int32_t *idx, float *grad_points,
                                      ws3d_stream_t stream) {
    return ws3d::group_launch(true, b, c, n, npoints, nsample, grad_out, idx, grad_points,
                              ws3d::as_stream(stream), "ws3d_group_points_grad");
}

extern "C" int ws3d_gather_points(int b, int c, int n, int npoints, const float *points,
                                  const int32_t *idx, float *out, ws3d_stream_t stream) {
    return ws3d::group_launch(false, b, c, n, npoints, 1, points, idx, out, ws3d::as_stream(stream),
                              "ws3d_gather_points");
}

extern "C" int ws3d_gather_points_grad(int b, int c, int n, int npoints, const float *grad_out,
                                       const int32_t *idx, float *grad_points, ws3d_stream_t stream) {
    return ws3d::group_launch(true, b, c, n, npoints, 1, grad_out, idx, grad_points,
                              ws3d::as_stream(stream), "ws3d_gather_points_grad");
}
