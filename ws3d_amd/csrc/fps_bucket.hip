// fps_bucket.hip -- pruned furthest point sampling for 4096 < n <= 16384 (gfx950).
//
// STATUS: experimental, selected only with WS3D_FPS_BUCKET=1.  Bit-exact (tests/
// test_gpu_parity.py::test_fps_bucket_kernel_subprocess) but NOT faster than fps.hip on MI355X
// (1.18 vs 1.17 us/step at n=16384): pruning removes most of the distance arithmetic, yet the
// step is bounded by the cross-lane chain (DPP reductions, readlanes, VGPR-indexed moves, LDS
// exchange), which pruning does not shorten, and the re-scan of the 32 cached per-lane values
// costs half of the dense sweep.  Kept as the starting point for round 2.
//
// Same contract as fps.hip (bit-exact indices incl. the reference's tie order), different
// work per step.  After j samples the running min-distance of every point is <= G_j (the
// maximum that selected sample j), so a new sample q can only change points closer to q than
// sqrt(G_j): a few percent of the scene after the first hundred samples.  The kernel therefore
//   * counting-sorts the scene into Z-order cells of the (x,z) plane in LDS and cuts the
//     sorted sequence into 256 buckets of 64 points; bucket b lives in slot b/8 of wave b%8,
//     one point per lane, coordinates and running min-distance in VGPRs (as in fps.hip);
//   * keeps a bounding box per bucket.  Per step each wave evaluates, for its 32 buckets at
//     once (one bucket per lane), the SAME fp32 distance expression on the box-to-q gaps:
//     by monotonicity of fp32 sub/mul/fma rounding it is a lower bound L of the distance the
//     kernel would compute for any point of the bucket, so L >= G_j proves min(d, temp) = temp
//     for the whole bucket -- the skip is exact, not approximate;
//   * updates only the surviving buckets (wave-uniform loop, VGPR-indexed moves), and only a
//     wave that updated something re-derives its candidate; the others republish their cached
//     record.  One LDS-only barrier per step, as before.
// Ties (exact duplicates are real in KITTI: scans shorter than 16384 points are padded by
// re-sampling, kitti_rcnn_dataset.py:435-441) cannot be resolved by slot/lane order any more
// because ownership is spatial.  Every wave therefore reports whether its maximum is attained
// more than once; if the global maximum is ambiguous the workgroup takes a (rare) resolution
// round that enumerates all points holding the maximum and selects the smallest reference
// rank  bitrev(k mod bs) * S + k / bs  -- the winner of the reference's reduction tree.
#include "common.h"

namespace ws3d {

constexpr int FB_NW = 8;        // waves
constexpr int FB_SL = 32;       // slots (buckets) per wave
constexpr int FB_NT = FB_NW * 64;
constexpr int FB_MAXN = FB_NW * FB_SL * 64;  // 16384
constexpr int FB_CELLS = 1024;  // 32 x 32 Z-order cells

typedef float f32x32 __attribute__((ext_vector_type(32)));

struct FbRec { float v; int pos; float x, y, z; int tie; int pad0, pad1; };  // 32 bytes

__device__ __forceinline__ int zcell(float x, float z, float xmin, float zmin, float ix, float iz) {
    const float fx = (x - xmin) * ix, fz = (z - zmin) * iz;
    const unsigned cx = fx > 0.f ? (fx < 31.f ? (unsigned)fx : 31u) : 0u;  // NaN -> 0
    const unsigned cz = fz > 0.f ? (fz < 31.f ? (unsigned)fz : 31u) : 0u;
    // 5+5 bit Morton code by table-free bit tricks
    unsigned a = cx, b = cz, code = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) code |= ((a >> i) & 1u) << (2 * i) | ((b >> i) & 1u) << (2 * i + 1);
    return (int)code;
}

__device__ __forceinline__ unsigned row16_min_u32(unsigned v) {
    unsigned r;
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(r));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    asm("s_nop 1\n\tv_min_u32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(r));
    return v;
}
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = row16_min_u32(v);
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }

__device__ __forceinline__ int fb_bitrev(int v, int bits) {
    return bits == 0 ? 0 : (int)(__builtin_bitreverse32((uint32_t)v) >> (32 - bits));
}

__global__ __launch_bounds__(FB_NT) void fps_bucket_kernel(const float *__restrict__ xyz,
                                                           float *__restrict__ temp,
                                                           int32_t *__restrict__ idx,
                                                           float *__restrict__ new_xyz, int n, int m,
                                                           int bs, int log2bs, int S) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint16_t *order = reinterpret_cast<uint16_t *>(smem);            // FB_MAXN: sorted position -> point index
    int *hist = reinterpret_cast<int *>(order + FB_MAXN);            // FB_CELLS
    float *bbox = reinterpret_cast<float *>(hist + FB_CELLS);        // FB_NW * FB_SL * 6
    FbRec *rec = reinterpret_cast<FbRec *>(bbox + FB_NW * FB_SL * 6);  // 2 * FB_NW
    unsigned *tiekey = reinterpret_cast<unsigned *>(rec + 2 * FB_NW);  // FB_NW
    float4 *tiept = reinterpret_cast<float4 *>(tiekey + FB_NW);      // 1 (16-byte aligned by construction)
    float *red = reinterpret_cast<float *>(tiept + 1);               // 4 * FB_NW
    int *wsum = reinterpret_cast<int *>(red + 4 * FB_NW);            // FB_NW

    const int b = blockIdx.x;
    xyz += (size_t)b * n * 3;
    idx += (size_t)b * m;
    if (temp) temp += (size_t)b * n;
    if (new_xyz) new_xyz += (size_t)b * m * 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    // ---------------- stage A: Z-order counting sort of the scene into `order`
    float xmn = INFINITY, xmx = -INFINITY, zmn = INFINITY, zmx = -INFINITY;
    for (int k = tid; k < n; k += FB_NT) {
        const float x = xyz[(size_t)k * 3], z = xyz[(size_t)k * 3 + 2];
        if (fabsf(x) < INFINITY) { xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); }
        if (fabsf(z) < INFINITY) { zmn = fminf(zmn, z); zmx = fmaxf(zmx, z); }
    }
    xmn = wave_min(xmn); xmx = wave_max(xmx); zmn = wave_min(zmn); zmx = wave_max(zmx);
    if (lane == 0) { red[w * 4 + 0] = xmn; red[w * 4 + 1] = xmx; red[w * 4 + 2] = zmn; red[w * 4 + 3] = zmx; }
    for (int i = tid; i < FB_CELLS; i += FB_NT) hist[i] = 0;
    for (int i = tid; i < FB_MAXN; i += FB_NT) order[i] = 0xFFFFu;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FB_NW; ++i) {
        xmn = fminf(xmn, red[i * 4 + 0]); xmx = fmaxf(xmx, red[i * 4 + 1]);
        zmn = fminf(zmn, red[i * 4 + 2]); zmx = fmaxf(zmx, red[i * 4 + 3]);
    }
    const float x0 = xmn <= xmx ? xmn : 0.f, z0 = zmn <= zmx ? zmn : 0.f;
    const float ix = (xmn < xmx) ? 32.0f / (xmx - xmn) : 0.f, iz = (zmn < zmx) ? 32.0f / (zmx - zmn) : 0.f;
    for (int k = tid; k < n; k += FB_NT)
        atomicAdd(&hist[zcell(xyz[(size_t)k * 3], xyz[(size_t)k * 3 + 2], x0, z0, ix, iz)], 1);
    __syncthreads();
    {   // exclusive scan of 1024 counters: 2 per thread
        const int a0 = hist[2 * tid], a1 = hist[2 * tid + 1];
        int v = a0 + a1;
        for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(v, o); if (lane >= o) v += t2; }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int off = 0;
        for (int i = 0; i < w; ++i) off += wsum[i];
        const int excl = off + v - (a0 + a1);
        hist[2 * tid] = excl;
        hist[2 * tid + 1] = excl + a0;
    }
    __syncthreads();
    for (int k = tid; k < n; k += FB_NT) {
        const int pos = atomicAdd(&hist[zcell(xyz[(size_t)k * 3], xyz[(size_t)k * 3 + 2], x0, z0, ix, iz)], 1);
        order[pos] = (uint16_t)k;
    }
    __syncthreads();

    // ---------------- registers: slot s of this lane = sorted position ((s*NW + w)*64 + lane)
    f32x32 px, py, pz, t;
#pragma unroll
    for (int s = 0; s < FB_SL; ++s) {
        const int pos = ((s * FB_NW + w) << 6) + lane;
        const int k = (int)order[pos];
        const bool valid = k != 0xFFFF;
        px[s] = valid ? xyz[(size_t)k * 3 + 0] : 0.f;
        py[s] = valid ? xyz[(size_t)k * 3 + 1] : 0.f;
        pz[s] = valid ? xyz[(size_t)k * 3 + 2] : 0.f;
        t[s] = valid ? (temp ? temp[k] : 1e10f) : -1.0f;   // -1 never beats a real candidate (>= 0)
        // bucket bounding box (NaN coordinates are ignored by fminf/fmaxf; their running
        // distance can never change -- min(NaN, temp) = temp -- so they need no coverage)
        const float lx = wave_min(valid ? px[s] : INFINITY), hx = wave_max(valid ? px[s] : -INFINITY);
        const float ly = wave_min(valid ? py[s] : INFINITY), hy = wave_max(valid ? py[s] : -INFINITY);
        const float lz = wave_min(valid ? pz[s] : INFINITY), hz = wave_max(valid ? pz[s] : -INFINITY);
        if (lane == 0) {
            float *bb = bbox + (w * FB_SL + s) * 6;
            bb[0] = lx; bb[1] = hx; bb[2] = ly; bb[3] = hy; bb[4] = lz; bb[5] = hz;
        }
    }
    __syncthreads();
    float blx = INFINITY, bhx = -INFINITY, bly = INFINITY, bhy = -INFINITY, blz = INFINITY, bhz = -INFINITY;
    if (lane < FB_SL) {
        const float *bb = bbox + (w * FB_SL + lane) * 6;
        blx = bb[0]; bhx = bb[1]; bly = bb[2]; bhy = bb[3]; blz = bb[4]; bhz = bb[5];
    }

    float qx = xyz[0], qy = xyz[1], qz = xyz[2];
    float G = INFINITY;  // upper bound of every running min-distance
    if (tid == 0) {
        idx[0] = 0;
        if (new_xyz) { new_xyz[0] = qx; new_xyz[1] = qy; new_xyz[2] = qz; }
    }
    // cached candidate of this wave
    float wv = -1.0f, wx = 0.f, wy = 0.f, wz = 0.f;
    int wpos = 0, wtie = 0;
    bool have = false;

    for (int j = 1; j < m; ++j) {
        // ---- which of my 32 buckets can change?  L = the kernel's own distance expression on
        // the per-axis gaps between q and the box (0 inside): a lower bound of d for every point
        unsigned need;
        {
            const float gx = fmaxf(fmaxf(blx - qx, qx - bhx), 0.f);
            const float gy = fmaxf(fmaxf(bly - qy, qy - bhy), 0.f);
            const float gz = fmaxf(fmaxf(blz - qz, qz - bhz), 0.f);
            const float L = sqdist3(gx, gy, gz);
            need = (unsigned)(__ballot(lane < FB_SL && L < G) & 0xFFFFFFFFull);
        }
        if (need != 0u || !have) {
            unsigned mm = need;
            while (mm) {  // wave-uniform; VGPR-indexed moves (a branch per slot measured slower)
                const int s = (int)__builtin_ctz(mm);
                mm &= mm - 1u;
                const float d = sqdist3(px[s] - qx, py[s] - qy, pz[s] - qz);
                t[s] = min_f32(d, t[s]);  // == fminf: t is never NaN
            }
            float best = -1.0f;
            int bslot = 0;
#pragma unroll
            for (int s = 0; s < FB_SL; ++s) {
                const bool gt = t[s] > best;
                bslot = gt ? s : bslot;
                best = gt ? t[s] : best;
            }
            const float wmax = wave_max(best);
            const uint64_t eq = __ballot(best == wmax);
            const int wl = (int)__builtin_ctzll(eq);
            const int wslot = __builtin_amdgcn_readlane(bslot, wl);
            int cnt = 0;  // how many of MY points hold the wave maximum
#pragma unroll
            for (int s = 0; s < FB_SL; ++s) cnt += (t[s] == wmax) ? 1 : 0;
            wtie = (__builtin_popcountll(eq) > 1 || __builtin_amdgcn_readlane(cnt, wl) > 1) ? 1 : 0;
            wv = wmax;
            wpos = ((wslot * FB_NW + w) << 6) + wl;
            wx = readlane_f(px[wslot], wl);
            wy = readlane_f(py[wslot], wl);
            wz = readlane_f(pz[wslot], wl);
            have = true;
        }
        const int buf = j & 1;
        if (lane == 0) {
            FbRec r;
            r.v = wv; r.pos = wpos; r.x = wx; r.y = wy; r.z = wz; r.tie = wtie; r.pad0 = 0; r.pad1 = 0;
            rec[buf * FB_NW + w] = r;
        }
        lds_barrier();
        const FbRec r = rec[buf * FB_NW + (lane & 7)];
        float vm;
        {   // max over the 8 records (lanes hold record lane&7): quad xor1, quad xor2, half mirror
            float a, c;
            asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(r.v));
            asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(a));
            asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(vm) : "v"(c));
        }
        const unsigned eq2 = (unsigned)(__ballot(r.v == vm) & 0xFFull);
        const int sel = (int)__builtin_ctz(eq2);
        const bool ambiguous = __builtin_popcount(eq2) > 1 || __builtin_amdgcn_readlane(r.tie, sel) != 0;
        int ipos;
        if (!ambiguous) {
            qx = readlane_f(r.x, sel); qy = readlane_f(r.y, sel); qz = readlane_f(r.z, sel);
            ipos = __builtin_amdgcn_readlane(r.pos, sel);
        } else {
            // ---- resolution round: smallest reference rank among ALL points holding vm
            const float gmax = readlane_f(vm, 0);
            unsigned key = 0xFFFFFFFFu;
#pragma unroll
            for (int s = 0; s < FB_SL; ++s) {
                if (t[s] == gmax) {
                    const int pos = ((s * FB_NW + w) << 6) + lane;
                    const int k = (int)order[pos];
                    const unsigned rank = (unsigned)(fb_bitrev(k & (bs - 1), log2bs) * S + (k >> log2bs));
                    key = min(key, (rank << 14) | (unsigned)pos);
                }
            }
            const unsigned wkey = wave_min_u32(key);
            if (lane == 0) tiekey[w] = wkey;
            lds_barrier();
            unsigned gk = tiekey[lane & 7];
            gk = min(gk, (unsigned)__builtin_amdgcn_update_dpp(0, (int)gk, DPP_QUAD_XOR1, 0xF, 0xF, false));
            gk = min(gk, (unsigned)__builtin_amdgcn_update_dpp(0, (int)gk, DPP_QUAD_XOR2, 0xF, 0xF, false));
            gk = min(gk, (unsigned)__builtin_amdgcn_update_dpp(0, (int)gk, DPP_ROW_HALF_MIRROR, 0xF, 0xF, false));
            ipos = (int)(__builtin_amdgcn_readfirstlane(gk) & 0x3FFFu);
            const int ob = ipos >> 6, ol = ipos & 63;           // owning bucket / lane
            if ((ob & (FB_NW - 1)) == w) {
                const int os = ob / FB_NW;                       // wave-uniform slot
                const float ox = readlane_f(px[os], ol), oy = readlane_f(py[os], ol), oz = readlane_f(pz[os], ol);
                if (lane == 0) *tiept = make_float4(ox, oy, oz, 0.f);
            }
            lds_barrier();
            const float4 c = *tiept;
            qx = c.x; qy = c.y; qz = c.z;
        }
        G = readlane_f(vm, 0);
        if (tid == 0) {
            idx[j] = ipos;  // sorted position; translated after the loop
            if (new_xyz) { new_xyz[j * 3 + 0] = qx; new_xyz[j * 3 + 1] = qy; new_xyz[j * 3 + 2] = qz; }
        }
    }

    // sorted positions -> point indices
    __syncthreads();
    for (int j = 1 + tid; j < m; j += FB_NT) idx[j] = (int)order[idx[j]];

    if (temp) {
#pragma unroll
        for (int s = 0; s < FB_SL; ++s) {
            const int k = (int)order[((s * FB_NW + w) << 6) + lane];
            if (k != 0xFFFF) temp[k] = t[s];
        }
    }
}

size_t fps_bucket_smem() {
    return sizeof(uint16_t) * FB_MAXN + sizeof(int) * FB_CELLS + sizeof(float) * FB_NW * FB_SL * 6 +
           sizeof(FbRec) * 2 * FB_NW + sizeof(unsigned) * FB_NW + sizeof(float4) + sizeof(float) * 4 * FB_NW +
           sizeof(int) * FB_NW + 64;
}

int fps_bucket_launch(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, float *new_xyz,
                      int bs, int log2bs, int S, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)fps_bucket_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)fps_bucket_smem());
        attr_set = true;
    }
    hipLaunchKernelGGL(fps_bucket_kernel, dim3(b), dim3(FB_NT), fps_bucket_smem(), st, xyz, temp, idx, new_xyz, n,
                       m, bs, log2bs, S);
    return check_launch("furthest_point_sampling(bucket)");
}

}  // namespace ws3d
