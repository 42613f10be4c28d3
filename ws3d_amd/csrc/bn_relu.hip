// bn_relu.hip -- training-mode BatchNorm (+ ReLU) of a channels-first activation (b, c, l).
//
// Every SharedMLP / Conv1d block of the reference is conv -> BatchNorm -> ReLU
// (pointnet2_lib/pointnet2/pytorch_utils.py:35-101).  In the Stage-1 training step the library
// BatchNorm + the separate ReLU passes were 8 of 22 ms, 5 of them in the first SA level: its
// tensors are the largest of the network (up to 268 MB) but have only 16..64 channels, and a
// one-workgroup-per-channel reduction leaves most of the chip idle.  Here the statistics are reduced
// per (scene, channel, 8192-element chunk) -- thousands of workgroups whatever c is -- into fp64
// partials that the second pass folds in a fixed order (deterministic, and more accurate than an
// fp32 tree); ReLU rides in the normalisation pass and its mask is re-derived in the backward
// pass from x, so neither the ReLU output mask nor a second activation copy is read.
// HBM passes: forward x, x -> y (3); backward (dy, x), (dy, x) -> dx (5); 10+ for the library pair.
//
//   y   = relu(((x - mean) * invstd) * gamma + beta)          (torch's native evaluation order)
//   dx  = gamma * invstd * (g - sum(g)/N - xhat * sum(g * xhat)/N),  g = dy * [pre-activation > 0]
//   dgamma = sum(g * xhat), dbeta = sum(g)
#include "common.h"

namespace ws3d {

constexpr int BN_CHUNK = 8192;   // elements of one (scene, channel) row per workgroup: 256 threads x 8 float4

// sum of a and b over the 256 threads of the block, valid in thread 0
__device__ __forceinline__ void block_sum2(double &a, double &b) {
    __shared__ double red[8];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[2 * w] = a; red[2 * w + 1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = (red[0] + red[2]) + (red[4] + red[6]);
        b = (red[1] + red[3]) + (red[5] + red[7]);
    }
}

struct BnRow {          // the chunk of one (scene, channel) row this workgroup owns
    size_t base;        // element offset of the row
    long start, end;    // element range inside the row
};

__device__ __forceinline__ BnRow bn_row(int c, long l) {
    BnRow r;
    r.base = ((size_t)blockIdx.z * c + blockIdx.y) * (size_t)l;
    r.start = (long)blockIdx.x * BN_CHUNK;
    r.end = min(l, r.start + BN_CHUNK);
    return r;
}

// MODE 0: sum x, sum x^2.   MODE 1: sum g, sum g * xhat  (g = dy masked by the re-derived ReLU mask)
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void bn_partial_kernel(int c, long l, int relu, const float *__restrict__ x,
                                                         const float *__restrict__ dy, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd, double *__restrict__ partial) {
    const BnRow r = bn_row(c, l);
    const int ch = blockIdx.y;
    float mu = 0.f, is = 0.f, ga = 0.f, be = 0.f;
    if (MODE == 1) { mu = mean[ch]; is = invstd[ch]; ga = gamma[ch]; be = beta[ch]; }
    float s1 = 0.f, s2 = 0.f;
    auto take = [&](float xv, float gv) {
        if (MODE == 0) {
            s1 = s1 + xv;
            s2 = s2 + xv * xv;
        } else {
            const float xh = (xv - mu) * is;
            const float g = (!relu || xh * ga + be > 0.f) ? gv : 0.f;
            s1 = s1 + g;
            s2 = s2 + g * xh;
        }
    };
    if (VEC) {
        for (long i = r.start + 4 * threadIdx.x; i < r.end; i += 1024) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + r.base + i);
            float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 1) gv = *reinterpret_cast<const float4 *>(dy + r.base + i);
            take(xv.x, gv.x); take(xv.y, gv.y); take(xv.z, gv.z); take(xv.w, gv.w);
        }
    } else {
        for (long i = r.start + threadIdx.x; i < r.end; i += 256) take(x[r.base + i], MODE == 1 ? dy[r.base + i] : 0.f);
    }
    double a = (double)s1, b = (double)s2;
    block_sum2(a, b);
    if (threadIdx.x == 0) {
        const size_t parts = (size_t)gridDim.z * gridDim.x;
        double *p = partial + ((size_t)ch * parts + (size_t)blockIdx.z * gridDim.x + blockIdx.x) * 2;
        p[0] = a; p[1] = b;
    }
}

// every workgroup of the second pass folds the channel's partials itself, in the same fixed order
// (<= a few hundred fp64 pairs from L2) -- cheaper than a separate finishing launch per layer
__device__ __forceinline__ void fold_partials(const double *__restrict__ partial, int ch, long parts, double &a, double &b) {
    __shared__ double folded[2];
    const double *p = partial + (size_t)ch * parts * 2;
    a = 0.0; b = 0.0;
    for (long i = threadIdx.x; i < parts; i += 256) { a += p[2 * i]; b += p[2 * i + 1]; }
    block_sum2(a, b);
    if (threadIdx.x == 0) { folded[0] = a; folded[1] = b; }
    __syncthreads();
    a = folded[0]; b = folded[1];
}

// MODE 0: y = relu(bn(x)).   MODE 1: dx from dy (dgamma / dbeta already reduced).
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void bn_apply_kernel(int c, long l, int relu, double count, float eps, float momentum,
                                                       const double *__restrict__ partial, const float *__restrict__ x,
                                                       const float *__restrict__ dy, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, float *__restrict__ mean,
                                                       float *__restrict__ invstd, float *__restrict__ running_mean,
                                                       float *__restrict__ running_var, float *__restrict__ dgamma,
                                                       float *__restrict__ dbeta, long long *__restrict__ batches_tracked,
                                                       float *__restrict__ out) {
    const BnRow r = bn_row(c, l);
    const int ch = blockIdx.y;
    const bool scribe = blockIdx.x == 0 && blockIdx.z == 0 && threadIdx.x == 0;   // writes the per-channel results
    double s1, s2;
    fold_partials(partial, ch, (long)gridDim.z * gridDim.x, s1, s2);
    float mu, is;
    const float ga = gamma[ch], be = beta[ch];
    float k0 = 0.f, k1 = 0.f, k2 = 0.f;
    if (MODE == 0) {
        const double m = s1 / count;
        double var = s2 / count - m * m;             // biased variance, fp64: no cancellation trouble for fp32 data
        if (var < 0.0) var = 0.0;
        mu = (float)m;
        is = (float)(1.0 / sqrt(var + (double)eps));
        if (scribe) {
            mean[ch] = mu;
            invstd[ch] = is;
            if (running_mean) running_mean[ch] = (1.0f - momentum) * running_mean[ch] + momentum * mu;
            if (running_var) running_var[ch] = (1.0f - momentum) * running_var[ch] + momentum * (float)(var * (count / (count - 1.0)));
            if (batches_tracked && ch == 0) *batches_tracked += 1;
        }
    } else {
        mu = mean[ch]; is = invstd[ch];
        const float db = (float)s1, dg = (float)s2;
        if (scribe) { dbeta[ch] = db; dgamma[ch] = dg; }
        const float inv_count = (float)(1.0 / count);
        k0 = ga * is; k1 = db * inv_count; k2 = dg * inv_count;
    }
    auto f = [&](float xv, float gv) -> float {
        const float xh = (xv - mu) * is;
        const float z = xh * ga + be;
        if (MODE == 0) return (relu && z < 0.f) ? 0.f : z;          // NaN passes through like clamp_min
        const float g = (!relu || z > 0.f) ? gv : 0.f;
        return k0 * ((g - k1) - xh * k2);
    };
    if (VEC) {
        for (long i = r.start + 4 * threadIdx.x; i < r.end; i += 1024) {
            const float4 xv = *reinterpret_cast<const float4 *>(x + r.base + i);
            float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 1) gv = *reinterpret_cast<const float4 *>(dy + r.base + i);
            *reinterpret_cast<float4 *>(out + r.base + i) = make_float4(f(xv.x, gv.x), f(xv.y, gv.y), f(xv.z, gv.z), f(xv.w, gv.w));
        }
    } else {
        for (long i = r.start + threadIdx.x; i < r.end; i += 256) out[r.base + i] = f(x[r.base + i], MODE == 1 ? dy[r.base + i] : 0.f);
    }
}

static bool bn_shape_ok(int b, int c, long l, const char *what) {
    if (b <= 0 || c <= 0 || l <= 0 || (long)b * l < 2 || b > 65535 || c > 65535 || (l + BN_CHUNK - 1) / BN_CHUNK > 0x7fffffffL) {
        set_error("%s: invalid shape (b=%d c=%d l=%ld; training statistics need more than one value per channel)", what, b, c, l);
        return false;
    }
    return true;
}

static size_t bn_ws_bytes(int b, int c, long l) {
    const size_t nchunk = (size_t)((l + BN_CHUNK - 1) / BN_CHUNK);
    return (size_t)c * (size_t)b * nchunk * 2 * sizeof(double);
}

static bool bn_vec_ok(long l, const void *p0, const void *p1, const void *p2) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(p0) | reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2);
    return (l & 3) == 0 && (al & 15) == 0;
}

}  // namespace ws3d

extern "C" size_t ws3d_bn_workspace_bytes(int b, int c, long l) {
    if (b <= 0 || c <= 0 || l <= 0) return 256;
    return ws3d::bn_ws_bytes(b, c, l) + 256;
}

extern "C" int ws3d_bn_relu_train_fwd(int b, int c, long l, const float *x, const float *gamma, const float *beta, float eps,
                                      float momentum, int relu, float *running_mean, float *running_var,
                                      int64_t *num_batches_tracked, float *y, float *save_mean, float *save_invstd,
                                      void *workspace, size_t workspace_bytes, ws3d_stream_t stream) {
    using namespace ws3d;
    if (!bn_shape_ok(b, c, l, "ws3d_bn_relu_train_fwd")) return WS3D_E_INVALID;
    if (!x || !gamma || !beta || !y || !save_mean || !save_invstd || !workspace || workspace_bytes < bn_ws_bytes(b, c, l) ||
        (reinterpret_cast<uintptr_t>(workspace) & 7)) {
        set_error("ws3d_bn_relu_train_fwd: null argument or workspace too small (%zu < %zu)", workspace_bytes, bn_ws_bytes(b, c, l));
        return WS3D_E_INVALID;
    }
    hipStream_t st = as_stream(stream);
    const int nchunk = (int)((l + BN_CHUNK - 1) / BN_CHUNK);
    const dim3 grid(nchunk, c, b), block(256);
    double *partial = reinterpret_cast<double *>(workspace);
    const bool vec = bn_vec_ok(l, x, y, x);
    if (vec) hipLaunchKernelGGL((bn_partial_kernel<0, true>), grid, block, 0, st, c, l, relu, x, nullptr, gamma, beta, nullptr, nullptr, partial);
    else hipLaunchKernelGGL((bn_partial_kernel<0, false>), grid, block, 0, st, c, l, relu, x, nullptr, gamma, beta, nullptr, nullptr, partial);
    const double count = (double)b * (double)l;
    if (vec) hipLaunchKernelGGL((bn_apply_kernel<0, true>), grid, block, 0, st, c, l, relu, count, eps, momentum, partial, x, nullptr, gamma, beta, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, reinterpret_cast<long long *>(num_batches_tracked), y);
    else hipLaunchKernelGGL((bn_apply_kernel<0, false>), grid, block, 0, st, c, l, relu, count, eps, momentum, partial, x, nullptr, gamma, beta, save_mean, save_invstd, running_mean, running_var, nullptr, nullptr, reinterpret_cast<long long *>(num_batches_tracked), y);
    return check_launch("ws3d_bn_relu_train_fwd");
}

extern "C" int ws3d_bn_relu_train_bwd(int b, int c, long l, const float *x, const float *dy, const float *gamma, const float *beta,
                                      const float *save_mean, const float *save_invstd, int relu, float *dx, float *dgamma,
                                      float *dbeta, void *workspace, size_t workspace_bytes, ws3d_stream_t stream) {
    using namespace ws3d;
    if (!bn_shape_ok(b, c, l, "ws3d_bn_relu_train_bwd")) return WS3D_E_INVALID;
    if (!x || !dy || !gamma || !beta || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !workspace ||
        workspace_bytes < bn_ws_bytes(b, c, l) || (reinterpret_cast<uintptr_t>(workspace) & 7)) {
        set_error("ws3d_bn_relu_train_bwd: null argument or workspace too small (%zu < %zu)", workspace_bytes, bn_ws_bytes(b, c, l));
        return WS3D_E_INVALID;
    }
    hipStream_t st = as_stream(stream);
    const int nchunk = (int)((l + BN_CHUNK - 1) / BN_CHUNK);
    const dim3 grid(nchunk, c, b), block(256);
    double *partial = reinterpret_cast<double *>(workspace);
    const bool vec = bn_vec_ok(l, x, dy, dx);
    if (vec) hipLaunchKernelGGL((bn_partial_kernel<1, true>), grid, block, 0, st, c, l, relu, x, dy, gamma, beta, save_mean, save_invstd, partial);
    else hipLaunchKernelGGL((bn_partial_kernel<1, false>), grid, block, 0, st, c, l, relu, x, dy, gamma, beta, save_mean, save_invstd, partial);
    const double count = (double)b * (double)l;
    float *mean_rw = const_cast<float *>(save_mean), *invstd_rw = const_cast<float *>(save_invstd);   // read only in this mode
    if (vec) hipLaunchKernelGGL((bn_apply_kernel<1, true>), grid, block, 0, st, c, l, relu, count, 0.f, 0.f, partial, x, dy, gamma, beta, mean_rw, invstd_rw, nullptr, nullptr, dgamma, dbeta, nullptr, dx);
    else hipLaunchKernelGGL((bn_apply_kernel<1, false>), grid, block, 0, st, c, l, relu, count, 0.f, 0.f, partial, x, dy, gamma, beta, mean_rw, invstd_rw, nullptr, nullptr, dgamma, dbeta, nullptr, dx);
    return check_launch("ws3d_bn_relu_train_bwd");
}
