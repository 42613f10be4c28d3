// proposals.hip -- the glue of the on-device proposal stage as two launches instead of ~25 tiny library
// kernels: the reference does these steps in Python on the host side of its ops (generate_box_dataset.py:92-140,
// kitti_utils.py:134-160, roipool3d_utils.py:19); here they sit between ws3d_topk_sorted, ws3d_nms_batched and
// ws3d_roipool3d inside the captured Stage-1 step, where every launch costs 3-5 us of a 1.9 ms batch.
//   ws3d_gather_boxes_bev   box rows in score order + their BEV rectangles (boxes3d_to_bev, kitti_utils.py:134-147)
//   ws3d_select_proposals   the first K survivors of the NMS as zero-padded (K,7) rows + scores + counts, and the rows
//                           enlarged for RoI pooling (enlarge_box3d, kitti_utils.py:150-160)
// Pure copies and single fp32 operations in the order the torch composition applies them: bit-identical to it.
#include "common.h"

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

__global__ __launch_bounds__(256) void select_proposals_kernel(int nb, int top, int keep_stride, int K, const float *__restrict__ box_sorted,
                                                               const float *__restrict__ sc, const int64_t *__restrict__ keep,
                                                               const int32_t *__restrict__ num, float extra2, float extra,
                                                               float *__restrict__ boxes_out, float *__restrict__ scores_out,
                                                               int64_t *__restrict__ count, float *__restrict__ pooled_boxes) {
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
    scores_out[i] = sc[(size_t)b * top + (size_t)src] * m;
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

extern "C" int ws3d_select_proposals(int b, int top, int keep_stride, int k, const float *box_sorted, const float *scores_sorted,
                                     const int64_t *keep, const int32_t *num, float extra_width, float *boxes_out, float *scores_out,
                                     int64_t *count, float *pooled_boxes, ws3d_stream_t stream) {
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
                       scores_out, count, pooled_boxes);
    return check_launch("ws3d_select_proposals");
}
