// decode_center_target + the proposal row of one point (see decode_center_boxes_kernel, iou3d.hip): shared by the per-point kernel
// and the decode-behind-the-top-k kernel of proposals.hip.  Every float operation is a separate fp32 op in the order of the torch
// composition (ws3d_amd/stage1.py): bit-identical to it.
#pragma once
#include "common.h"

namespace ws3d {

// t = scene * n + k: the point's row in xyz (.., 3) / reg (.., 4 * bins); k = its index inside the scene (the synthetic heading)
__device__ __forceinline__ void decode_center_box(long t, int k_in_scene, int bins, float loc_scope, float bin_size, float h, float w, float l,
                                                  const float *__restrict__ xyz, const float *__restrict__ reg, float (&o)[7]) {
    const float *r = reg + t * 4 * bins;
    const float half = bin_size / 2;
    float pos[2];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float *bl = r + a * bins;
        int best = 0;
        float bv = bl[0];
        for (int i = 1; i < bins; ++i) {
            const float v = bl[i];
            if (v > bv || (v != v && bv == bv)) { bv = v; best = i; }
        }
        float p = (float)best * bin_size;
        p = p + half;
        p = p - loc_scope;
        const float res = r[(2 + a) * bins + best] * half;
        pos[a] = p + res;
    }
    const float *q = xyz + t * 3;
    const unsigned long long k = (unsigned long long)k_in_scene;
    const double hk = (double)((k * 2654435761ull) % 4294967296ull);
    const float ry = (float)(hk / 4294967296.0 * (2.0 * 3.141592653589793) - 3.141592653589793);
    o[0] = pos[0] + q[0];
    o[1] = q[1] + h / 2;
    o[2] = pos[1] + q[2];
    o[3] = h; o[4] = w; o[5] = l;
    o[6] = ry;
}

}  // namespace ws3d
