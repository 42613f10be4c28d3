// proposals.hip -- the glue of the on-device proposal stage as two launches instead of ~25 tiny library
// kernels: the reference does these steps in Python on the host side of its ops (generate_box_dataset.py:92-140,
// kitti_utils.py:134-160, roipool3d_utils.py:19); here they sit between ws3d_topk_sorted, ws3d_nms_batched and
// ws3d_roipool3d inside the captured Stage-1 step, where every launch costs 3-5 us of a 1.9 ms batch.
//   ws3d_gather_boxes_bev   box rows in score order + their BEV rectangles (boxes3d_to_bev, kitti_utils.py:134-147)
//   ws3d_select_proposals   the first K survivors of the NMS as zero-padded (K,7) rows + scores + counts, and the rows
//                           enlarged for RoI pooling (enlarge_box3d, kitti_utils.py:150-160)
// Pure copies and single fp32 operations in the order the torch composition applies them: bit-identical to it.
#include <algorithm>
#include "common.h"
#include "decode.h"

namespace ws3d {

__global__ __launch_bounds__(256) void gather_boxes_bev_kernel(long total, int n, int top, const float *__restrict__ box,
                                                               const int64_t *__restrict__ order, float *__restrict__ box_sorted,
                                                               float *__restrict__ bev) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (scene, rank)
    if (i >= total) return;
    const long b = i / top;
    const int64_t src = order[i];
    const float *p = box + ((size_t)b * n + (size_t)src) * 7;
    float v[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) v[q] = p[q];
    float *o = box_sorted + (size_t)i * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) o[q] = v[q];
    const float half_l = v[5] / 2.0f, half_w = v[4] / 2.0f;     // kitti_utils.py:139-140
    float *e = bev + (size_t)i * 5;
    e[0] = v[0] - half_l; e[1] = v[2] - half_w; e[2] = v[0] + half_l; e[3] = v[2] + half_w; e[4] = v[6];
}

// the same rows without the (B, N, 7) tensor of every point's box: only the `top` points the top-k chose are decoded
// (ws3d_decode_center_boxes' arithmetic, decode.h), in score order, with their BEV rectangles
__global__ __launch_bounds__(256) void decode_gather_boxes_bev_kernel(long total, int n, int top, int bins, float loc_scope, float bin_size, float h,
                                                                      float w, float l, const float *__restrict__ xyz, const float *__restrict__ reg,
                                                                      const int64_t *__restrict__ order, float *__restrict__ box_sorted,
                                                                      float *__restrict__ bev) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;      // (scene, rank)
    if (i >= total) return;
    const long b = i / top;
    const int64_t src = order[i];
    float v[7];
    decode_center_box(b * n + (long)src, (int)src, bins, loc_scope, bin_size, h, w, l, xyz, reg, v);
    float *o = box_sorted + (size_t)i * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) o[q] = v[q];
    const float half_l = v[5] / 2.0f, half_w = v[4] / 2.0f;     // kitti_utils.py:139-140
    float *e = bev + (size_t)i * 5;
    e[0] = v[0] - half_l; e[1] = v[2] - half_w; e[2] = v[0] + half_l; e[3] = v[2] + half_w; e[4] = v[6];
}

// The step's prologue in one launch: the (rows, c) input rows split into coordinates (rows, 3) and features (rows, c - 3) -- what
// pointcloud[..., 0:3].contiguous() / [..., 3:].contiguous() do with a strided-copy launch each -- and the pass's zero arena cleared
// (the pooled outputs the compact SharedMLPs reduce into with an atomic max, pair totals, tickets).  Blocks [0, split_blocks) split,
// the others clear 16 bytes per lane and trip.
__global__ __launch_bounds__(256) void split_points_clear_kernel(long rows, int c, int split_blocks, const float *__restrict__ pc,
                                                                 float *__restrict__ xyz, float *__restrict__ feats, uint4 *__restrict__ clear,
                                                                 long clear_vec) {
    if ((int)blockIdx.x < split_blocks) {
        const long i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= rows) return;
        const float *p = pc + i * c;
        float *o = xyz + i * 3;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        for (int q = 3; q < c; ++q) feats[i * (c - 3) + (q - 3)] = p[q];
        return;
    }
    const long stride = (long)(gridDim.x - split_blocks) * 256;
    for (long i = (long)(blockIdx.x - split_blocks) * 256 + threadIdx.x; i < clear_vec; i += stride) clear[i] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(256) void select_proposals_kernel(int nb, int top, int keep_stride, int K, const float *__restrict__ box_sorted,
                                                               const float *__restrict__ sc, const int64_t *__restrict__ keep,
                                                               const int32_t *__restrict__ num, float extra2, float extra,
                                                               float *__restrict__ boxes_out, float *__restrict__ scores_out,
                                                               int64_t *__restrict__ count, float *__restrict__ pooled_boxes,
                                                               float *__restrict__ packed, long packed_stride, int count_in_row) {
    const int i = blockIdx.x * 256 + threadIdx.x;             // (scene, slot)
    if (i >= nb * K) return;
    const int b = i / K, pos = i - b * K;
    const int cnt = min(num[b], K);
    if (pos == 0) count[b] = (int64_t)cnt;
    const bool valid = pos < cnt;
    // iou3d_ops.nms_gpu_padded_batched + stage1.proposals_from_rpn: idx = -1 where invalid, clamp(min=0), gather, * valid
    int64_t src = valid ? keep[(size_t)b * keep_stride + pos] : 0;
    src = src < 0 ? 0 : (src > top - 1 ? top - 1 : src);
    const float m = valid ? 1.0f : 0.0f;
    const float *p = box_sorted + ((size_t)b * top + (size_t)src) * 7;
    float v[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) v[q] = p[q] * m;
    float *o = boxes_out + (size_t)i * 7;
#pragma unroll
    for (int q = 0; q < 7; ++q) o[q] = v[q];
    const float s_ = sc[(size_t)b * top + (size_t)src] * m;
    scores_out[i] = s_;
    if (packed) {                                             // (K, 8) rows = box + score: what ws3d_amd.dist gathers across ranks
        // packed_stride floats per scene (K * 8: the dense (B, K, 8) tensor; K * 8 + 1 with count_in_row: the exchange's send row,
        // the scene's count as the float behind its K rows -- exact, counts are < 2^24)
        float *g = packed + (size_t)b * packed_stride + (size_t)pos * 8;
        if (count_in_row && pos == 0) packed[(size_t)b * packed_stride + (size_t)K * 8] = (float)cnt;
#pragma unroll
        for (int q = 0; q < 7; ++q) g[q] = v[q];
        g[7] = s_;
    }
    if (pooled_boxes) {                                       // enlarge_box3d: h, w, l += 2 e;  y_bottom += e
        float *g = pooled_boxes + (size_t)i * 7;
        g[0] = v[0]; g[1] = v[1] + extra; g[2] = v[2]; g[3] = v[3] + extra2; g[4] = v[4] + extra2; g[5] = v[5] + extra2; g[6] = v[6];
    }
}

}  // namespace ws3d

extern "C" int ws3d_gather_boxes_bev(int b, int n, int top, const float *box, const int64_t *order, float *box_sorted, float *bev,
                                     ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || top < 0 || top > n || !box || !order || !box_sorted || !bev) {
        set_error("ws3d_gather_boxes_bev: invalid argument (b=%d n=%d top=%d)", b, n, top);
        return WS3D_E_INVALID;
    }
    const long total = (long)b * top;
    if (total == 0) return WS3D_OK;
    hipLaunchKernelGGL(gather_boxes_bev_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), total, n, top, box,
                       order, box_sorted, bev);
    return check_launch("ws3d_gather_boxes_bev");
}

extern "C" int ws3d_decode_gather_boxes_bev(int b, int n, int top, int bins, float loc_scope, float loc_bin_size, float h, float w, float l,
                                            const float *xyz, const float *rpn_reg, const int64_t *order, float *box_sorted, float *bev,
                                            ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || top < 0 || top > n || bins <= 0 || !xyz || !rpn_reg || !order || !box_sorted || !bev) {
        set_error("ws3d_decode_gather_boxes_bev: invalid argument (b=%d n=%d top=%d bins=%d)", b, n, top, bins);
        return WS3D_E_INVALID;
    }
    const long total = (long)b * top;
    if (total == 0) return WS3D_OK;
    hipLaunchKernelGGL(decode_gather_boxes_bev_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), total, n, top, bins,
                       loc_scope, loc_bin_size, h, w, l, xyz, rpn_reg, order, box_sorted, bev);
    return check_launch("ws3d_decode_gather_boxes_bev");
}

extern "C" int ws3d_split_points_clear(long rows, int c, const float *pc, float *xyz, float *feats, void *clear, size_t clear_bytes,
                                       ws3d_stream_t stream) {
    using namespace ws3d;
    if (rows < 0 || c < 3 || (rows > 0 && (!pc || !xyz || (c > 3 && !feats))) || (clear_bytes && (!clear || (clear_bytes & 15) ||
        (reinterpret_cast<uintptr_t>(clear) & 15)))) {
        set_error("ws3d_split_points_clear: invalid argument (rows=%ld c=%d clear_bytes=%zu; the cleared range is 16-byte aligned and sized)", rows, c,
                  clear_bytes);
        return WS3D_E_INVALID;
    }
    const long split_blocks = (rows + 255) / 256;
    const long clear_vec = (long)(clear_bytes / 16);
    const long clear_blocks = clear_vec ? std::min<long>((clear_vec + 1023) / 1024, 2048) : 0;       // >= 4 trips of 16 bytes per lane
    if (split_blocks + clear_blocks == 0) return WS3D_OK;
    if (split_blocks + clear_blocks > 0x7fffffffL) { set_error("ws3d_split_points_clear: too many rows"); return WS3D_E_UNSUPPORTED; }
    hipLaunchKernelGGL(split_points_clear_kernel, dim3((unsigned)(split_blocks + clear_blocks)), dim3(256), 0, as_stream(stream), rows, c,
                       (int)split_blocks, pc, xyz, feats, reinterpret_cast<uint4 *>(clear), clear_vec);
    return check_launch("ws3d_split_points_clear");
}

static int select_proposals_impl(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                 const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                 int64_t *count, float *pooled_boxes, float *packed, long packed_stride, int count_in_row,
                                 ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || top <= 0 || k <= 0 || keep_stride < (k < top ? k : top) || !box_sorted || !scores_sorted || !keep || !num || !boxes_out ||
        !scores_out || !count) {
        set_error("ws3d_select_proposals: invalid argument (b=%d top=%d k=%d keep_stride=%d)", b, top, k, keep_stride);
        return WS3D_E_INVALID;
    }
    if (b == 0) return WS3D_OK;
    // extra_width * 2 is formed in double and rounded once, as `large[:, 3:6] += extra_width * 2` does with a Python float
    hipLaunchKernelGGL(select_proposals_kernel, dim3((unsigned)(((long)b * k + 255) / 256)), dim3(256), 0, as_stream(stream), b, top,
                       keep_stride, k, box_sorted, scores_sorted, keep, num, (float)((double)extra_width * 2.0), extra_width, boxes_out,
                       scores_out, count, pooled_boxes, packed, packed_stride, count_in_row);
    return check_launch("ws3d_select_proposals");
}

extern "C" int ws3d_select_proposals(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                     const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                     int64_t *count, float *pooled_boxes, ws3d_stream_t stream) {
    return select_proposals_impl(b, top, keep_stride, k, box_sorted, scores_sorted, keep, num, extra_width, boxes_out, scores_out, count,
                                 pooled_boxes, nullptr, 0, 0, stream);
}

extern "C" int ws3d_select_proposals_packed(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                            const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                            int64_t *count, float *pooled_boxes, float *packed, ws3d_stream_t stream) {
    return select_proposals_impl(b, top, keep_stride, k, box_sorted, scores_sorted, keep, num, extra_width, boxes_out, scores_out, count,
                                 pooled_boxes, packed, (long)k * 8, 0, stream);
}

extern "C" int ws3d_select_proposals_send(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                          const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                          int64_t *count, float *pooled_boxes, float *send, long send_stride, ws3d_stream_t stream) {
    if (!send || send_stride < (long)k * 8 + 1) {
        ws3d::set_error("ws3d_select_proposals_send: send rows need k * 8 + 1 = %ld floats (stride %ld)", (long)k * 8 + 1, send_stride);
        return WS3D_E_INVALID;
    }
    return select_proposals_impl(b, top, keep_stride, k, box_sorted, scores_sorted, keep, num, extra_width, boxes_out, scores_out, count,
                                 pooled_boxes, send, send_stride, 1, stream);
}
