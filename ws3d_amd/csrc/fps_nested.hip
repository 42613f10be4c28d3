// fps_nested.hip -- furthest point sampling of a cloud that is ALREADY in sampling order (gfx950).
//
// The set-abstraction levels of PointNet++ sample each other's output: level l+1 runs furthest_point_sample on the
// centres level l selected, in the order level l selected them (pointnet2_modules.py:30-36, pointnet2_msg.py:56-70).  Greedy
// furthest-point selection has a nesting property: the point the sweep over ALL n points picked at step k maximises the
// running minimum distance over every subset that contains it, so the sweep over the selected subset picks the same point
// again -- sampling a cloud that is the output of a sampling pass returns idx = 0, 1, 2, ..., m-1, unless a step has TWO
// points holding the maximum (then the reference's tie order, which differs between the two launches, decides).  The
// reference fixture shows it: tests/golden/stage1_forward.npz has fps_idx_1..3 = arange.
//
// ws3d_furthest_point_sampling_nested therefore does not trust, it VERIFIES, in parallel instead of m dependent steps:
//   A  W[k] = min(1e10, d(x_k, x_0), ..., d(x_k, x_{k-1}))            k < m   (the running minimum of point k when it is picked)
//   B  every point p walks k = 1 .. m-1 with its own running minimum t_p(k) (the reference's arithmetic: the same sqdist3,
//      fminf, 1e10) and checks t_p(k) < W[k] for p != k -- STRICTLY.  If that holds for all p and k, point k is the unique
//      maximum of step k, and the reference selects it whatever its tie order: idx = arange, new_xyz = xyz[:m], exactly.
//   C  a scene that fails the check (ties, or a cloud that was not in sampling order to begin with) is sampled by a literal
//      restatement of the reference kernel (strided ownership, strict '>', shared-memory tree that keeps the lower slot:
//      sampling_gpu.cu:86-209) -- slow and rare.
// Cost at 4096 -> 1024 points x 8 scenes: three short launches (~130 us stand-alone) against 600 us of dependent steps.
// (Giving every point a 16-lane DPP row in A and B -- row-wide minimum / prefix minimum over the samples 16 j + c -- cut that to
// 44 us but doubles the instructions: c3 lost 1.5 % throughput for no latency gain -- the caller's stream does not wait here --
// so the one-thread-per-point form stays.)
// Scratch: W lives in idx (as float bits), the per-scene verdict in new_xyz[0] (as int bits); both are overwritten by C.
#include <cmath>

#include "common.h"

namespace ws3d {

static int nested_opt_n_threads(int work_size) {      // cuda_utils.h:10-14 (the launcher's own libm expression)
    const int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

// The samples are staged in LDS once per workgroup (coalesced loads) and read back as broadcast 16-byte records, 8 per trip:
// a loop that fetches sample k through scalar loads waits a full cache round trip per step (measured: 0.5 us per step).
__global__ __launch_bounds__(256) void fps_nested_w_kernel(int n, int m, const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                           float *__restrict__ new_xyz) {
    extern __shared__ __attribute__((aligned(16))) float4 smp[];                         // samples 0 .. kmax-1
    const int b = blockIdx.y, tid = threadIdx.x, k = blockIdx.x * 256 + tid;
    xyz += (size_t)b * n * 3;
    if (k == 0) reinterpret_cast<int32_t *>(new_xyz + (size_t)b * m * 3)[0] = 0;       // the verdict: 0 = in sampling order
    const int kmax = min(blockIdx.x * 256 + 255, m - 1);                                 // workgroup-uniform trip count
    for (int i = tid; i < kmax; i += 256) smp[i] = make_float4(xyz[i * 3 + 0], xyz[i * 3 + 1], xyz[i * 3 + 2], 0.f);
    const int kc = min(k, m - 1);
    const float x = xyz[kc * 3 + 0], y = xyz[kc * 3 + 1], z = xyz[kc * 3 + 2];
    __syncthreads();
    float w = 1e10f;
    int i = 0;
    for (; i + 8 <= kmax; i += 8) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = smp[i + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w2 = min_f32(sqdist3(x - q[u].x, y - q[u].y, z - q[u].z), w);
            w = i + u < k ? w2 : w;
        }
    }
    for (; i < kmax; ++i) {
        const float4 q = smp[i];
        const float w2 = min_f32(sqdist3(x - q.x, y - q.y, z - q.z), w);
        w = i < k ? w2 : w;
    }
    if (k < m) idx[(size_t)b * m + k] = __builtin_bit_cast(int32_t, w);
}

__global__ __launch_bounds__(256) void fps_nested_check_kernel(int n, int m, const float *__restrict__ xyz, const int32_t *__restrict__ idx,
                                                               float *__restrict__ new_xyz) {
    extern __shared__ __attribute__((aligned(16))) float4 smp[];                         // record k: {sample k-1, W[k]}, k = 1 .. m-1
    const int b = blockIdx.y, tid = threadIdx.x, p = blockIdx.x * 256 + tid;
    xyz += (size_t)b * n * 3;
    const float *wk = reinterpret_cast<const float *>(idx + (size_t)b * m);
    for (int k = 1 + tid; k < m; k += 256) smp[k] = make_float4(xyz[(k - 1) * 3 + 0], xyz[(k - 1) * 3 + 1], xyz[(k - 1) * 3 + 2], wk[k]);
    const int pc = min(p, n - 1);
    const float x = xyz[pc * 3 + 0], y = xyz[pc * 3 + 1], z = xyz[pc * 3 + 2];
    __syncthreads();
    float t = 1e10f;
    bool bad = false;
    int k = 1;
    for (; k + 8 <= m; k += 8) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = smp[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            t = min_f32(sqdist3(x - q[u].x, y - q[u].y, z - q[u].z), t);     // the running minimum of point p before step k + u
            bad = bad || (p != k + u && !(t < q[u].w));                       // not strictly below the picked point's: a tie
        }
    }
    for (; k < m; ++k) {
        const float4 q = smp[k];
        t = min_f32(sqdist3(x - q.x, y - q.y, z - q.z), t);
        bad = bad || (p != k && !(t < q.w));
    }
    if (__ballot(bad && p < n) != 0 && (tid & 63) == 0) atomicOr(reinterpret_cast<int32_t *>(new_xyz + (size_t)b * m * 3), 1);
}

// The fallback of a scene whose check failed: sampling_gpu.cu:93-209 with temp = 1e10 held in registers and the SAME selection -- the
// reference's launch picks, among the points holding the maximum, the one whose owner thread has the smallest bit-reversed tid (its
// shared-memory tree keeps the lower slot at every level: the lowest differing bit decides) and, inside a thread, the smallest k
// (strict '>' over k = tid, tid + bs, ..): a total order (value descending, rank = bitrev(tid) * 4 + s ascending), so ANY reduction
// order finds the reference's pick.  Round 6: a wave reduces with DPP (wave_max, then wave_min_u32 of the ranks that hold it), the 16
// waves exchange one 32-byte record each through double-buffered LDS -- the picked point's coordinates travel in the record, so a step
// reads no global memory -- ONE barrier per step instead of the tree's ten: 1.6 ms -> ~0.5 ms for 4096 -> 1024 (the literal tree was
// 1.5 us per step; 6 % of the hdl64 scenes take this path, profiles/r06_nested_fallback.txt).  One workgroup of 1024 threads, pointers
// of ONE scene; n <= 4096.
static __device__ __forceinline__ unsigned nested_wave_min_u32(unsigned v) {
    unsigned r;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(r));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(r));
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

struct NestedRec { float v; unsigned rank; int k; float x, y, z; int pad0, pad1; };
__device__ __forceinline__ void nested_literal_fps(int n, int m, int bs, const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                   float *__restrict__ new_xyz, float *dists, int *dists_i) {
    NestedRec *rec = reinterpret_cast<NestedRec *>(dists);          // [2][16] records: 1 KB of the caller's 4 KB
    (void)dists_i;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int lb = 0;
    while ((1 << (lb + 1)) <= bs) ++lb;                              // bs is a power of two (cuda_utils.h:10-14)
    const unsigned brev = lb == 0 ? 0u : (__builtin_bitreverse32((unsigned)tid) >> (32 - lb));
    float px[4], py[4], pz[4], tmp[4];
    bool own[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int k = tid + s * bs;
        own[s] = tid < bs && k < n;
        px[s] = own[s] ? xyz[k * 3 + 0] : 0.f; py[s] = own[s] ? xyz[k * 3 + 1] : 0.f; pz[s] = own[s] ? xyz[k * 3 + 2] : 0.f;
        tmp[s] = 1e10f;
    }
    float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
    if (tid == 0) { idx[0] = 0; new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
    for (int j = 1; j < m; ++j) {
        float best = -1.f;
        int bs_ = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (own[s]) {
                const float d = sqdist3(px[s] - x1, py[s] - y1, pz[s] - z1);
                const float d2 = min_f32(d, tmp[s]);
                tmp[s] = d2;
                bs_ = d2 > best ? s : bs_;
                best = d2 > best ? d2 : best;
            }
        }
        const unsigned rank = tid < bs ? brev * 4u + (unsigned)bs_ : 0xffffffffu;
        const float vmax = wave_max(best);
        const unsigned rmin = nested_wave_min_u32(best == vmax ? rank : 0xffffffffu);
        float4 *slot = reinterpret_cast<float4 *>(rec + (j & 1) * 16);      // two 16-byte halves per record: {v, rank, k, x} {y, z, -, -}
        if (rank == rmin && rmin != 0xffffffffu) {
            const float qx = bs_ == 0 ? px[0] : bs_ == 1 ? px[1] : bs_ == 2 ? px[2] : px[3];
            const float qy = bs_ == 0 ? py[0] : bs_ == 1 ? py[1] : bs_ == 2 ? py[2] : py[3];
            const float qz = bs_ == 0 ? pz[0] : bs_ == 1 ? pz[1] : bs_ == 2 ? pz[2] : pz[3];
            slot[2 * w] = make_float4(vmax, __uint_as_float(rmin), __int_as_float(tid + bs_ * bs), qx);
            slot[2 * w + 1] = make_float4(qy, qz, 0.f, 0.f);
        } else if (lane == 0 && rmin == 0xffffffffu) {
            slot[2 * w] = make_float4(-2.f, __uint_as_float(0xffffffffu), 0.f, 0.f);       // a wave without a point of its own
            slot[2 * w + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        lds_barrier();
        // every wave folds the 16 records once more with the same two DPP reductions -- lane q < 16 holds record q -- instead of every
        // THREAD reading all of them (16 waves x 32 broadcast reads per step kept the LDS pipe busy for 2-4 k clocks: 2.8 ms per scene in
        // the first build); scalars, not a struct: a select between two aggregates goes through scratch memory
        const float4 a0 = lane < 16 ? slot[2 * lane] : make_float4(-3.f, __uint_as_float(0xffffffffu), 0.f, 0.f);
        const float4 a1 = lane < 16 ? slot[2 * lane + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float gmax = wave_max(a0.x);
        const unsigned gmin = nested_wave_min_u32(a0.x == gmax ? __float_as_uint(a0.y) : 0xffffffffu);
        const int src = __builtin_ctzll(__ballot(a0.x == gmax && __float_as_uint(a0.y) == gmin));      // exactly one lane
        const int rk = __builtin_amdgcn_readlane(__float_as_int(a0.z), src);
        const float rx = readlane_f(a0.w, src), ry = readlane_f(a1.x, src), rz = readlane_f(a1.y, src);
        x1 = rx; y1 = ry; z1 = rz;
        if (tid == 0) { idx[j] = rk; new_xyz[j * 3 + 0] = x1; new_xyz[j * 3 + 1] = y1; new_xyz[j * 3 + 2] = z1; }
    }
}

// C: verified scenes get idx = arange and new_xyz = xyz[:m]; the others the reference kernel, restated literally.
// keep_verdict (chain entry): the scene's verdict is also left in keep_verdict[b * keep_stride] for fps_nested_tail_kernel.
__global__ __launch_bounds__(1024) void fps_nested_finish_kernel(int n, int m, int bs, const float *__restrict__ xyz, int32_t *__restrict__ idx,
                                                                 float *__restrict__ new_xyz, int32_t *__restrict__ keep_verdict, long keep_stride) {
    __shared__ float dists[1024];
    __shared__ int dists_i[1024];
    __shared__ int verdict;
    const int b = blockIdx.x, tid = threadIdx.x;
    xyz += (size_t)b * n * 3;
    idx += (size_t)b * m;
    new_xyz += (size_t)b * m * 3;
    if (tid == 0) {
        verdict = reinterpret_cast<const int32_t *>(new_xyz)[0];
        if (keep_verdict) keep_verdict[(size_t)b * keep_stride] = verdict;
    }
    __syncthreads();
    if (verdict == 0) {
        for (int k = tid; k < m; k += 1024) idx[k] = k;
        for (int e = tid; e < m * 3; e += 1024) new_xyz[e] = xyz[e];
        return;
    }
    nested_literal_fps(n, m, bs, xyz, idx, new_xyz, dists, dists_i);
}

// The levels BELOW a verified level, all in one launch (round 5).  Level A's check proved, for every point p of its input and every
// step k < m_A, that p's running minimum stays strictly below the picked point's: the same inequality for p < m_A and k < m_B <= m_A
// is what level B's own check would ask of level A's OUTPUT (new_xyz_A = xyz[:m_A], the same floats, the same arithmetic) -- so when
// level A verified, every deeper level of the chain is verified with it: idx = arange, new_xyz = the prefix.  A scene whose level A
// did NOT verify (ties, or an input that was not in sampling order) takes the literal restatement level by level: exact either way.
struct NestedTail { int levels; int m[4]; int bs[4]; int32_t *idx[4]; float *new_xyz[4]; };
__global__ __launch_bounds__(1024) void fps_nested_tail_kernel(int n_a, const float *__restrict__ xyz_a, NestedTail t) {
    __shared__ float dists[1024];
    __shared__ int dists_i[1024];
    __shared__ int verdict;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) verdict = t.idx[0][(size_t)b * t.m[0]];              // left there by level A's finish kernel
    __syncthreads();
    const int v = verdict;
    __syncthreads();
    const float *src = xyz_a + (size_t)b * n_a * 3;
    int n_src = n_a;
    for (int l = 0; l < t.levels; ++l) {
        const int m = t.m[l];
        int32_t *idx = t.idx[l] + (size_t)b * m;
        float *nx = t.new_xyz[l] + (size_t)b * m * 3;
        if (v == 0) {
            for (int k = tid; k < m; k += 1024) idx[k] = k;
            for (int e = tid; e < m * 3; e += 1024) nx[e] = src[e];
        } else {
            nested_literal_fps(n_src, m, t.bs[l], src, idx, nx, dists, dists_i);
            __syncthreads();
            __threadfence_block();
        }
        src = nx;            // (a verified chain keeps reading prefixes of prefixes: the same values)
        n_src = m;
        __syncthreads();
    }
}

}  // namespace ws3d

extern "C" int ws3d_furthest_point_sampling_nested(int b, int n, int m, const float *xyz, int32_t *idx, float *new_xyz,
                                                   ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || m < 0 || m > n || !xyz || ((!idx || !new_xyz) && m > 0)) {
        set_error("ws3d_furthest_point_sampling_nested: invalid argument (b=%d n=%d m=%d)", b, n, m);
        return WS3D_E_INVALID;
    }
    if (b == 0 || m == 0) return WS3D_OK;
    if (n > 4096 || b > 65535) {
        set_error("ws3d_furthest_point_sampling_nested: n=%d > 4096 (or b=%d > 65535) is not covered, use ws3d_furthest_point_sampling_gather", n, b);
        return WS3D_E_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    const size_t lds = sizeof(float) * 4 * (size_t)m;                                    // <= 64 KB (m <= n <= 4096)
    hipLaunchKernelGGL(fps_nested_w_kernel, dim3((m + 255) / 256, b), dim3(256), lds, st, n, m, xyz, idx, new_xyz);
    hipLaunchKernelGGL(fps_nested_check_kernel, dim3((n + 255) / 256, b), dim3(256), lds, st, n, m, xyz, idx, new_xyz);
    hipLaunchKernelGGL(fps_nested_finish_kernel, dim3(b), dim3(1024), 0, st, n, m, nested_opt_n_threads(n), xyz, idx, new_xyz, (int32_t *)nullptr, 0L);
    return check_launch("furthest_point_sampling_nested");
}

extern "C" int ws3d_furthest_point_sampling_nested_chain(int b, int n, int levels, const int *m, const float *xyz, int32_t *const *idx,
                                                         float *const *new_xyz, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || levels < 1 || levels > 5 || !m || !xyz || !idx || !new_xyz) {
        set_error("ws3d_furthest_point_sampling_nested_chain: invalid argument (b=%d n=%d levels=%d)", b, n, levels);
        return WS3D_E_INVALID;
    }
    int prev = n;
    for (int l = 0; l < levels; ++l) {
        if (m[l] <= 0 || m[l] > prev || !idx[l] || !new_xyz[l]) {
            set_error("ws3d_furthest_point_sampling_nested_chain: level %d: m=%d of %d points (or a NULL output)", l, m[l], prev);
            return WS3D_E_INVALID;
        }
        prev = m[l];
    }
    if (b == 0) return WS3D_OK;
    if (n > 4096 || b > 65535) {
        set_error("ws3d_furthest_point_sampling_nested_chain: n=%d > 4096 (or b=%d > 65535) is not covered", n, b);
        return WS3D_E_UNSUPPORTED;
    }
    hipStream_t st = as_stream(stream);
    const int m0 = m[0];
    const size_t lds = sizeof(float) * 4 * (size_t)m0;
    hipLaunchKernelGGL(fps_nested_w_kernel, dim3((m0 + 255) / 256, b), dim3(256), lds, st, n, m0, xyz, idx[0], new_xyz[0]);
    hipLaunchKernelGGL(fps_nested_check_kernel, dim3((n + 255) / 256, b), dim3(256), lds, st, n, m0, xyz, idx[0], new_xyz[0]);
    int32_t *keep = levels > 1 ? idx[1] : nullptr;           // the verdict waits in the next level's first index slot
    hipLaunchKernelGGL(fps_nested_finish_kernel, dim3(b), dim3(1024), 0, st, n, m0, nested_opt_n_threads(n), xyz, idx[0], new_xyz[0], keep,
                       (long)(levels > 1 ? m[1] : 0));
    if (levels > 1) {
        NestedTail t;
        t.levels = levels - 1;
        for (int l = 1; l < levels; ++l) {
            t.m[l - 1] = m[l]; t.bs[l - 1] = nested_opt_n_threads(m[l - 1]); t.idx[l - 1] = idx[l]; t.new_xyz[l - 1] = new_xyz[l];
        }
        for (int l = levels - 1; l < 4; ++l) { t.m[l] = 0; t.bs[l] = 1; t.idx[l] = nullptr; t.new_xyz[l] = nullptr; }
        hipLaunchKernelGGL(fps_nested_tail_kernel, dim3(b), dim3(1024), 0, st, m0, (const float *)new_xyz[0], t);
    }
    return check_launch("furthest_point_sampling_nested_chain");
}
