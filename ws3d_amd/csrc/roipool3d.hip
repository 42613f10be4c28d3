// roipool3d.hip -- RoI point pooling for gfx950.  Replaces roipool3d_cuda.forward /
// forward_slow (roipool3d.cpp:15-79 -> roipool3d_kernel.cu:31-237) and gives a device
// twin of pts_in_boxes3d_cpu (roipool3d.cpp:97-124).
//
// Design.  The reference runs three kernels through a
// B*N*M int32 scratch (cudaMalloc/cudaFree per call, 134 MB/scene at config 5) and
// scans with ONE THREAD per box.  Here one 256-lane workgroup owns 1 or 4 boxes of a scene:
//   1. the box frames (cy, cos, sin, half extents) are computed once;
//   2. each of the 4 waves scans a contiguous quarter of the scene, 64 points per
//      step: in-box test -> __ballot -> mbcnt prefix -> ordered append to the wave's
//      LDS list (ascending point index; no atomics, no barrier inside the scan);
//   3. the 4 lists are concatenated (their ranges are ordered), truncated to S and
//      wrap-padded (idx[k] = idx[k % cnt]) in LDS;
//   4. the S x (3+C) output block of a box is contiguous: the workgroup streams it out
//      with fully coalesced 16-byte stores, gathering 512-byte feature rows.
// No scratch, no allocation, no host sync; the output write (B*M*S*(3+C)*4 bytes) is
// the only large HBM stream, which makes this an HBM-roofline kernel.
#include <cstdlib>

#include "common.h"

#ifdef WS3D_ROI_PROF   // scripts/ubench/roi_prof.hip: per-workgroup wall-clock timeline (100 MHz)
__device__ long long g_roi_prof[8192 * 4];
#define ROI_PROF(slot) if (threadIdx.x == 0) g_roi_prof[blockIdx.x * 4 + slot] = wall_clock64();
#else
#define ROI_PROF(slot)
#endif

namespace ws3d {

typedef float float4v __attribute__((ext_vector_type(4)));
typedef float4v float4u __attribute__((aligned(4)));  // 16-byte access, 4-byte aligned
typedef float f3v __attribute__((ext_vector_type(3)));
typedef f3v f3u __attribute__((aligned(4)));           // 12-byte access, 4-byte aligned

struct BoxFrame {
    float cx, cy, cz, hh, hw, hl, cosa, sina;
};

// roipool3d_kernel.cu:14-28 pt_in_box3d, box-constant part.  h/2.0, l/2.0, w/2.0 are
// exact in both double and float (division by two), so the reference's mixed
// float/double comparisons reduce to float comparisons against these halves.
__device__ __forceinline__ BoxFrame make_frame(const float *bx) {
    BoxFrame f;
    f.cx = bx[0];
    f.cz = bx[2];
    const float h = bx[3], w = bx[4], l = bx[5], angle = bx[6];
    f.cy = (float)((double)bx[1] - (double)h / 2.0);
    f.hh = h * 0.5f;
    f.hw = w * 0.5f;
    f.hl = l * 0.5f;
    f.cosa = cosf_cr(angle);
    f.sina = sinf_cr(angle);
    return f;
}

// roipool3d_kernel.cu:14-28 pt_in_box3d as ONE ordered comparison.  The reference's early
// "return 0" exits only skip work, so the flag is the AND of
//   !(|dx| > 10), !(|dy| > hh), !(|dz| > 10), -hl <= x_rot <= hl, -hw <= z_rot <= hw.
// For fp32 a <= b  <=>  fl(a - b) <= 0 (the sign of a difference is exact, subnormals kept), and
// -e <= v <= e  <=>  |v| <= e, so with t = (|x_rot|-hl, |z_rot|-hw, |dx|-10, |dy|-hh, |dz|-10) the
// flag is max(t) <= 0 -- except for NaN, which v_max3_f32 drops while the reference's >= / <= on a
// NaN x_rot / z_rot (NaN or inf coordinates, angle or extent) say "outside" and its '>' tests on a
// NaN |dy| say "keep going".  x_rot and z_rot are NaN together (shared operands; |dx|,|dz| <= 10
// and |cos|,|sin| <= 1 exclude inf - inf), so one ordered-compare of t0, t1 restores that.
// SMALL (frame_is_small: hl^2 + hw^2 < 98, i.e. a BEV half-diagonal under 9.9 m -- every car-sized box): the two "|dx| > 10",
// "|dz| > 10" terms cannot decide anything and are left out.  Proof: the rotation is an isometry up to rounding, so a point
// that passes |x_rot| <= hl and |z_rot| <= hw has dx^2 + dz^2 <= (hl^2 + hw^2)(1 + 1e-5) < 100, hence |dx|, |dz| < 10; a point
// that fails them is outside either way.  (x_rot, z_rot carry an absolute error <= 2^-22 (|dx| + |dz|); for |dx| or |dz| so
// large that this matters, one of |x_rot|, |z_rot| is of the same magnitude and fails its test.  inf / NaN coordinates give
// inf / NaN x_rot or z_rot: outside under both spellings.)
template <bool SMALL = false>
__device__ __forceinline__ bool pt_in_frame(const BoxFrame &f, float x, float y, float z) {
    const float dx = x - f.cx, dy = y - f.cy, dz = z - f.cz;
    const float x_rot = dx * f.cosa + dz * (-f.sina);
    const float z_rot = dx * f.sina + dz * f.cosa;
    const float t0 = fabsf(x_rot) - f.hl, t1 = fabsf(z_rot) - f.hw;
    const float t3 = fabsf(dy) - f.hh;
    float m;
    if (SMALL) m = __builtin_fmaxf(__builtin_fmaxf(t0, t1), t3);
    else {
        const float t2 = fabsf(dx) - 10.0f, t4 = fabsf(dz) - 10.0f;
        m = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(t0, t1), __builtin_fmaxf(t2, t3)), t4);
    }
    return !(m > 0.0f) && !__builtin_isunordered(t0, t1);      // (&&: two lane masks and one s_and; '&' materialises both bools in VGPRs)
}

__device__ __forceinline__ bool frame_is_small(const BoxFrame &f) { return f.hl * f.hl + f.hw * f.hw < 98.0f; }   // false for NaN

// XCD-aware workgroup -> (scene, box group) mapping for a 1-D grid of batch * groups workgroups: workgroup g runs on XCD
// g % 8 (observed dispatch order), so with a batch that is a multiple of 8 scene = (g / 8 / groups) * 8 + g % 8 keeps all
// workgroups of a scene on ONE XCD -- every one of them streams the whole scene (786 KB at 65536 points), which then stays in
// that XCD's 4 MB L2 instead of being pulled through all eight (8 scenes = 6.3 MB do not fit one L2).
__device__ __forceinline__ void roi_scene_group(int batch, int groups, int &b, int &grp) {
    const int g = blockIdx.x;
    if ((batch & 7) == 0) {
        const int j = g >> 3;
        b = (j / groups) * 8 + (g & 7);
        grp = j - (j / groups) * groups;
    } else {
        b = g / groups;
        grp = g - b * groups;
    }
}

// BG boxes of one scene per workgroup: every point loaded by the scan is tested against BG box
// frames, so the scene is streamed from L2 once per BG boxes instead of once per box (at config 5
// the per-box rescans were 3x the output bytes).
// CR > 0: the copy phase stages CR rows at a time through LDS (see there).
template <int BG, int CR>
__global__ __launch_bounds__(256) void roipool3d_kernel(int pts_num, int boxes_num, int feat_len,
                                                        int S, const float *__restrict__ xyz,
                                                        const float *__restrict__ boxes3d,
                                                        const float *__restrict__ pts_feature,
                                                        float *__restrict__ pooled,
                                                        int32_t *__restrict__ empty_flag,
                                                        int32_t *__restrict__ pts_idx, int fill, int batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int *lists = reinterpret_cast<int *>(smem);  // BG * 4 * S
    int *sel = lists + BG * 4 * S;               // S
    __shared__ int wcnt_s[BG * 4];

    ROI_PROF(0)
    int b, grp;
    roi_scene_group(batch, (boxes_num + BG - 1) / BG, b, grp);
    const int box0 = grp * BG;
    const int nb = min(BG, boxes_num - box0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    xyz += (size_t)b * pts_num * 3;
    pts_feature += (size_t)b * pts_num * feat_len;
    BoxFrame f[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) f[g] = make_frame(boxes3d + ((size_t)b * boxes_num + box0 + min(g, nb - 1)) * 7);
    bool all_small = true;
#pragma unroll
    for (int g = 0; g < BG; ++g) all_small = all_small && frame_is_small(f[g]);

    ROI_PROF(3)
    const int Q = (((pts_num + 3) / 4 + 63) / 64) * 64;
    const int start = min(w * Q, pts_num), end = min(start + Q, pts_num);
    int wcnt[BG];
#pragma unroll
    for (int g = 0; g < BG; ++g) wcnt[g] = g < nb ? 0 : S;
    // 4 sub-blocks of 64 points per trip, software-pipelined: the 12 loads of trip t+1 are issued
    // before the tests of trip t, so the L2 round trip hides behind ~500 instructions of tests
    float nx[4], ny[4], nz[4];
    auto load_trip = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 64 + lane;   // loads are clamped, never branched; a lane past the
            // wave's range gets x = NaN = "outside".  ONE 12-byte load per point: three dword loads per point made the scan
            // address-bound (12 wave loads per trip, each spanning six cache lines)
            const f3v p = *reinterpret_cast<const f3u *>(xyz + (size_t)min(k, pts_num - 1) * 3);
            nx[u] = k < end ? p.x : __builtin_nanf(""); ny[u] = p.y; nz[u] = p.z;
        }
    };
#if defined(WS3D_ROI_NO_SCAN)   // ablation: pretend every wave found 40 points
    for (int g = 0; g < BG; ++g) { if (lane < 40) lists[(g * 4 + w) * S + lane] = start + lane * 7; wcnt[g] = g < nb ? 40 : S; }
    for (int k0 = end; k0 < end; k0 += 256) {
#else
    if (start < end) load_trip(start);
    for (int k0 = start; k0 < end; k0 += 256) {
#endif
        float x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = nx[u]; y[u] = ny[u]; z[u] = nz[u]; }
        load_trip(k0 + 256);
        // all 4 x BG tests first (independent VALU work, masks in SGPRs), bookkeeping afterwards:
        // a test followed directly by its own ballot branch serialises on VALU->SALU round trips
        uint64_t mask[4][BG];
        if (all_small) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < BG; ++g) mask[u][g] = __builtin_amdgcn_ballot_w64(pt_in_frame<true>(f[g], x[u], y[u], z[u]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int g = 0; g < BG; ++g) mask[u][g] = __builtin_amdgcn_ballot_w64(pt_in_frame<false>(f[g], x[u], y[u], z[u]));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = k0 + u * 64 + lane;
#pragma unroll
            for (int g = 0; g < BG; ++g) {
                const uint64_t mk = mask[u][g];
                if (mk) {  // wave-uniform; a full list (wcnt >= S) takes no more appends: pos < S fails
                    const int wc = __builtin_amdgcn_readfirstlane(wcnt[g]);
                    const int pos = wc + mbcnt(mk);
                    if (((mk >> lane) & 1ull) && pos < S) lists[(g * 4 + w) * S + pos] = k;
                    wcnt[g] = min(wc + (int)__builtin_popcountll(mk), S);
                }
            }
        }
        bool all_full = true;
#pragma unroll
        for (int g = 0; g < BG; ++g) all_full = all_full && wcnt[g] >= S;
        if (all_full) break;
    }
    if (lane == 0) {
#pragma unroll
        for (int g = 0; g < BG; ++g) wcnt_s[g * 4 + w] = min(wcnt[g], S);
    }
    __syncthreads();
    ROI_PROF(1)

    const int row = 3 + feat_len;
    const int total = S * row;
    const bool vec_rows = feat_len >= 4 && (feat_len & 3) == 0 && (reinterpret_cast<uintptr_t>(pts_feature) & 15) == 0;
    for (int g = 0; g < nb; ++g) {
        const size_t bm = (size_t)b * boxes_num + box0 + g;
        const int *lg = lists + g * 4 * S;
        const int c0 = wcnt_s[g * 4 + 0], c1 = wcnt_s[g * 4 + 1], c2 = wcnt_s[g * 4 + 2], c3 = wcnt_s[g * 4 + 3];
        const int cnt = min(c0 + c1 + c2 + c3, S);
        if (cnt == 0) {  // roipool3d_kernel.cu:147-149,181-183: flag the box, leave its rows untouched
            if (tid == 0) empty_flag[bm] = 1;
            if (pts_idx)
                for (int q = tid; q < S; q += 256) pts_idx[bm * S + q] = 0;
            if (fill) {  // ws3d_roipool3d_fill: the caller did not pre-zero, write the zeros of an empty box here
                float *o = pooled + bm * (size_t)total;
                const float4v zero = {0.f, 0.f, 0.f, 0.f};
                for (int q = tid * 4; q + 3 < total; q += 1024) *reinterpret_cast<float4u *>(o + q) = zero;
                for (int q = (total & ~3) + tid; q < total; q += 256) o[q] = 0.f;
            }
            continue;  // wave-uniform for the whole workgroup
        }
        if (fill && tid == 0) empty_flag[bm] = 0;
        for (int q = tid; q < S; q += 256) {
            int t = q < cnt ? q : q % cnt;  // duplicate_idx = k % cnt (roipool3d_kernel.cu:153-157)
            int v;
            if (t < c0) v = lg[t];
            else if ((t -= c0) < c1) v = lg[S + t];
            else if ((t -= c1) < c2) v = lg[2 * S + t];
            else v = lg[3 * S + (t - c2)];
            sel[q] = v;
            if (pts_idx) pts_idx[bm * S + q] = v;
        }
        __syncthreads();
#ifdef WS3D_ROI_NO_COPY
        if (pts_num >= 0) continue;
#endif
        float *out = pooled + bm * (size_t)S * row;
        auto fetch = [&](int sr, int j) -> float {
            const int src = sel[sr];
            return j < 3 ? xyz[(size_t)src * 3 + j] : pts_feature[(size_t)src * feat_len + (j - 3)];
        };
        if (CR > 0) {
            // Through LDS: a pooled row is 3 + C floats = 524 bytes at C = 128, so a feature row lands 12 bytes off a
            // 16-byte boundary, and 16-byte stores that straddle 16-byte slots run at HALF the write bandwidth of aligned
            // ones on this chip (scripts/ubench/row_copy.hip: 2.5 vs 5.1 TB/s, stores only).  The S x (3+C) block of a box is
            // one contiguous, 16-byte-aligned stream: CR rows are gathered with aligned 16-byte loads (32 lanes per row)
            // into a tight copy in LDS, and the CR * (3+C) / 4 units of that stretch are stored aligned; two buffers, one
            // barrier per stretch.  (host: C % 4 == 0, C <= 128, S * (3+C) % 4 == 0, 16-byte-aligned pointers)
            float *stage = reinterpret_cast<float *>(smem) + (((BG * 4 + 1) * S + 3) & ~3);
            const int half = tid >> 5, l32 = tid & 31;
            const int f4 = feat_len >> 2;
            constexpr int RPH = CR > 0 ? CR / 8 : 1;
            for (int c0 = 0, it = 0; c0 < S; c0 += CR, ++it) {
                float *st = stage + (it & 1) * CR * row;
                const int nr = min(CR, S - c0);
                float4v v[RPH];
                float p3[RPH];
#pragma unroll
                for (int u = 0; u < RPH; ++u) {
                    const int src = sel[min(c0 + half * RPH + u, S - 1)];
                    if (l32 < f4) v[u] = reinterpret_cast<const float4v *>(pts_feature + (size_t)src * feat_len)[l32];
                    if (l32 < 3) p3[u] = xyz[(size_t)src * 3 + l32];
                }
#pragma unroll
                for (int u = 0; u < RPH; ++u) {
                    const int r = half * RPH + u;
                    if (r < nr) {
                        if (l32 < f4) *reinterpret_cast<float4u *>(st + r * row + 3 + 4 * l32) = v[u];
                        if (l32 < 3) st[r * row + l32] = p3[u];
                    }
                }
                __syncthreads();
                float4v *o4 = reinterpret_cast<float4v *>(out + (size_t)c0 * row);
                const int units = (nr * row) >> 2;
                for (int q = tid; q < units; q += 256)
                    __builtin_nontemporal_store(*reinterpret_cast<const float4v *>(st + 4 * q), o4 + q);
            }
        } else if (vec_rows) {
            // feature rows are 16-byte aligned in the source: 32 lanes move one row with aligned
            // 16-byte loads and (4-byte aligned) 16-byte stores, lanes 0-2 carry x, y, z; 4 rows
            // per half-wave and trip keep 8 loads in flight per lane
            const int half = tid >> 5, l32 = tid & 31;
            const int f4 = feat_len >> 2;
            for (int sr0 = half * 4; sr0 < S; sr0 += 32) {
                int src[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) src[u] = sel[min(sr0 + u, S - 1)];
                for (int c = l32; c < f4; c += 32) {
                    float4v v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        v[u] = reinterpret_cast<const float4v *>(pts_feature + (size_t)src[u] * feat_len)[c];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (sr0 + u < S) {
                            // streaming store: the output is never re-read here, the gathered rows are
                            // (overlapping boxes share points) -- keep those in L2/MALL (measured -7 %)
                            __builtin_nontemporal_store(v[u], reinterpret_cast<float4u *>(out + (size_t)(sr0 + u) * row + 3 + 4 * c));
                        }
                }
                if (l32 < 3) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (sr0 + u < S) out[(size_t)(sr0 + u) * row + l32] = xyz[(size_t)src[u] * 3 + l32];
                }
            }
        } else if ((total & 3) == 0 && ((bm * (size_t)total) & 3) == 0) {
            // the S x (3+C) block of a box is contiguous and 16-byte aligned: 4 elements per store
            int sr = (4 * tid) / row, j = 4 * tid - sr * row;
            const int ds = 1024 / row, dj = 1024 - ds * row;
            for (int e = 4 * tid; e < total; e += 1024) {
                float4 v4;
                int s1 = sr, j1 = j;
                v4.x = fetch(s1, j1); if (++j1 == row) { j1 = 0; ++s1; }
                v4.y = fetch(s1, j1); if (++j1 == row) { j1 = 0; ++s1; }
                v4.z = fetch(s1, j1); if (++j1 == row) { j1 = 0; ++s1; }
                v4.w = fetch(s1, j1);
                *reinterpret_cast<float4 *>(out + e) = v4;
                sr += ds; j += dj;
                if (j >= row) { j -= row; ++sr; }
            }
        } else {
            int sr = tid / row, j = tid - sr * row;
            const int ds = 256 / row, dj = 256 - ds * row;
            for (int e = tid; e < total; e += 256) {
                out[e] = fetch(sr, j);
                sr += ds; j += dj;
                if (j >= row) { j -= row; ++sr; }
            }
        }
        __syncthreads();  // sel is reused by the next box
    }
    ROI_PROF(2)
}

// ---- the large-scene variant: scan and copy overlapped inside the workgroup -------------------------------------------------
// In roipool3d_kernel every workgroup of the launch scans (VALU-bound), then copies (HBM-bound), all of them in lock-step, so
// the two phases add (c5: 0.28 + 0.22 ms).  Here the 4 boxes of a workgroup are handled in 4 / SG passes of SG boxes, and the
// scan loop of pass p carries the copy of pass p-1: each trip of the scan (256 points per wave, ~170 * SG VALU instructions)
// issues the gathers of one stretch of CR pooled rows before its tests and stores that stretch (through LDS, aligned: see the
// copy phase of roipool3d_kernel) after them.  The lists hold 16-bit offsets into the wave's quarter of the scene, so a pass needs
// SG * 4 * S * 2 bytes of lists + SG * S * 4 of selected indices + 2 * CR * (3 + C) * 4 of staging.
// Measured at c5 (scripts/ablate_roi.sh): 0.56 ms (direct copy) -> 0.52 (staged copy) -> 0.47 (this kernel, SG = 2, CR = 16);
// scan alone 0.26, copy alone 0.25.  What was tried on top and did not help: one box per pass (0.53: the per-pass overhead
// of the scan), 32-row stretches (140 VGPRs, 3 waves per SIMD: 0.57), stretches owned by single waves without the workgroup
// barrier (137 VGPRs: 0.57), starting every CU's third and fourth workgroup 16-130 us late to break the lock-step (slower by
// a quarter of the delay).
template <int SG, int CR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void roipool3d_pipe_kernel(int pts_num, int boxes_num, int feat_len, int S, const float *__restrict__ xyz,
                                                             const float *__restrict__ boxes3d, const float *__restrict__ pts_feature,
                                                             float *__restrict__ pooled, int32_t *__restrict__ empty_flag,
                                                             int32_t *__restrict__ pts_idx, int fill, int batch) {
    constexpr int BG = 4, RPH = CR / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int row = 3 + feat_len;
    float *stage = reinterpret_cast<float *>(smem);                                     // 2 * CR * row
    int *sel = reinterpret_cast<int *>(stage + 2 * CR * row);                           // SG * S: the subgroup being copied
    uint16_t *lists = reinterpret_cast<uint16_t *>(sel + SG * S);                       // SG * 4 * S: the subgroup being scanned
    __shared__ int wcnt_s[SG * 4];
    __shared__ int cnt_s[SG];

    int b, grp;
    roi_scene_group(batch, (boxes_num + BG - 1) / BG, b, grp);
    const int box0 = grp * BG;
    const int nb = min(BG, boxes_num - box0);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int half = tid >> 5, l32 = tid & 31, f4 = feat_len >> 2;
    xyz += (size_t)b * pts_num * 3;
    pts_feature += (size_t)b * pts_num * feat_len;
    const int Q = (((pts_num + 3) / 4 + 63) / 64) * 64;
    const int start = min(w * Q, pts_num), end = min(start + Q, pts_num);
    const int trips = (Q + 255) / 256;
    const int cpb = (S + CR - 1) / CR;                                                  // stretches per box

    // ---- the copy of one stretch of CR pooled rows, in two halves around the tests of a scan trip: gathers (aligned 16-byte
    // loads, 32 lanes per row), then -- through LDS, two buffers, one barrier -- CR * (3+C) / 4 aligned 16-byte stores
    float4v cv[RPH];
    float cp3[RPH];
    auto issue = [&](int c) {
        const int g = c / cpb, c0 = (c - g * cpb) * CR;
        if (cnt_s[g] == 0) return;
#pragma unroll
        for (int u = 0; u < RPH; ++u) {
            const int src = sel[g * S + min(c0 + half * RPH + u, S - 1)];
            if (l32 < f4) cv[u] = reinterpret_cast<const float4v *>(pts_feature + (size_t)src * feat_len)[l32];
            if (l32 < 3) cp3[u] = xyz[(size_t)src * 3 + l32];
        }
    };
    auto finish = [&](int c, int sub0, int parity) {
        const int g = c / cpb, c0 = (c - g * cpb) * CR;
        if (cnt_s[g] == 0) return;                                                      // workgroup-uniform
        float *st = stage + parity * CR * row;
        const int nr = min(CR, S - c0);
#pragma unroll
        for (int u = 0; u < RPH; ++u) {
            const int r = half * RPH + u;
            if (r < nr) {
                if (l32 < f4) *reinterpret_cast<float4u *>(st + r * row + 3 + 4 * l32) = cv[u];
                if (l32 < 3) st[r * row + l32] = cp3[u];
            }
        }
        __syncthreads();
        const size_t bm = (size_t)b * boxes_num + box0 + sub0 + g;
        float4v *o4 = reinterpret_cast<float4v *>(pooled + (bm * S + c0) * (size_t)row);
        const int units = (nr * row) >> 2;
        for (int q = tid; q < units; q += 256) __builtin_nontemporal_store(*reinterpret_cast<const float4v *>(st + 4 * q), o4 + q);
    };

    int parity = 0;
    for (int sub0 = 0; sub0 < BG + SG; sub0 += SG) {             // pass: scan boxes sub0.., copy boxes sub0 - SG..
        const bool scanning = sub0 < nb;
#ifdef WS3D_ROI_NO_COPY
        const int nchunks = 0;
#else
        const int nchunks = sub0 > 0 ? min(SG, nb - (sub0 - SG)) * cpb : 0;             // (sub0 - SG < nb always holds here)
#endif
        if (!scanning && nchunks == 0) break;
        BoxFrame f[SG];
        int wcnt[SG];
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            f[g] = make_frame(boxes3d + ((size_t)b * boxes_num + box0 + min(sub0 + g, nb - 1)) * 7);
            wcnt[g] = (scanning && sub0 + g < nb) ? 0 : S;
        }
        bool all_small = true;
#pragma unroll
        for (int g = 0; g < SG; ++g) all_small = all_small && frame_is_small(f[g]);
        float nx[4], ny[4], nz[4];
        auto load_trip = [&](int k0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * 64 + lane;
                const f3v pp = *reinterpret_cast<const f3u *>(xyz + (size_t)min(k, pts_num - 1) * 3);      // one 12-byte load per point
                nx[u] = k < end ? pp.x : __builtin_nanf(""); ny[u] = pp.y; nz[u] = pp.z;
            }
        };
        bool done = !scanning || start >= end;
#ifdef WS3D_ROI_NO_SCAN   // ablation: pretend every wave found 40 points
        for (int g = 0; g < SG; ++g) { if (lane < 40) lists[(g * 4 + w) * S + lane] = (uint16_t)(lane * 7); if (sub0 + g < nb) wcnt[g] = 40; }
        done = true;
#endif
        if (!done) load_trip(start);
        int next_chunk = 0, acc = 0, upto = 0;
        const int ntrips = scanning ? trips : 0;
        for (int t = 0; t < ntrips; ++t) {
            const int k0 = start + t * 256;
            acc += nchunks;                                         // upto = floor((t + 1) * nchunks / ntrips), without the division
            while (acc >= ntrips) { acc -= ntrips; ++upto; }
            const bool has = next_chunk < upto;
            if (has) issue(next_chunk);
            if (!done && k0 < end) {
                float x[4], y[4], z[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { x[u] = nx[u]; y[u] = ny[u]; z[u] = nz[u]; }
                load_trip(k0 + 256);
                uint64_t mask[4][SG];
                if (all_small) {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int g = 0; g < SG; ++g) mask[u][g] = __builtin_amdgcn_ballot_w64(pt_in_frame<true>(f[g], x[u], y[u], z[u]));
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int g = 0; g < SG; ++g) mask[u][g] = __builtin_amdgcn_ballot_w64(pt_in_frame<false>(f[g], x[u], y[u], z[u]));
                }
                bool all_full = true;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = t * 256 + u * 64 + lane;                              // offset into the wave's quarter
#pragma unroll
                    for (int g = 0; g < SG; ++g) {
#if defined(WS3D_ROI_SCAN_ABL) && WS3D_ROI_SCAN_ABL == 1      // ablation: tests without the list appends (one hit keeps them alive)
                        const uint64_t mk = mask[u][g] & (k0 == start ? 1ull : 0ull);
#elif defined(WS3D_ROI_SCAN_ABL) && WS3D_ROI_SCAN_ABL == 2    // ablation: loads and loop only
                        const uint64_t mk = (k0 == start && x[u] == 12345.f) ? 1ull : 0ull;
#else
                        const uint64_t mk = mask[u][g];
#endif
                        if (mk) {
                            const int wc = __builtin_amdgcn_readfirstlane(wcnt[g]);
                            const int pos = wc + mbcnt(mk);
                            if (((mk >> lane) & 1ull) && pos < S) lists[(g * 4 + w) * S + pos] = (uint16_t)k;
                            wcnt[g] = min(wc + (int)__builtin_popcountll(mk), S);
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < SG; ++g) all_full = all_full && wcnt[g] >= S;
                done = all_full;
            }
            if (has) { finish(next_chunk, sub0 - SG, parity); parity ^= 1; ++next_chunk; }
            while (next_chunk < upto) { issue(next_chunk); finish(next_chunk, sub0 - SG, parity); parity ^= 1; ++next_chunk; }
        }
        while (next_chunk < nchunks) { issue(next_chunk); finish(next_chunk, sub0 - SG, parity); parity ^= 1; ++next_chunk; }
        if (!scanning) break;
        if (lane == 0) {
#pragma unroll
            for (int g = 0; g < SG; ++g) wcnt_s[g * 4 + w] = min(wcnt[g], S);
        }
        __syncthreads();                                            // lists complete; every stretch of the previous subgroup stored
        // ---- selected indices of this subgroup: concatenate, truncate, wrap-pad (roipool3d_kernel.cu:139-157)
        for (int g = 0; g < SG && sub0 + g < nb; ++g) {
            const size_t bm = (size_t)b * boxes_num + box0 + sub0 + g;
            const uint16_t *lg = lists + g * 4 * S;
            const int c0 = wcnt_s[g * 4 + 0], c1 = wcnt_s[g * 4 + 1], c2 = wcnt_s[g * 4 + 2], c3 = wcnt_s[g * 4 + 3];
            const int cnt = min(c0 + c1 + c2 + c3, S);
            if (tid == 0) cnt_s[g] = cnt;
            if (cnt == 0) {      // roipool3d_kernel.cu:147-149,181-183: flag the box, leave its rows untouched
                if (tid == 0) empty_flag[bm] = 1;
                if (pts_idx)
                    for (int q = tid; q < S; q += 256) pts_idx[bm * S + q] = 0;
                if (fill) {
                    float4v *o = reinterpret_cast<float4v *>(pooled + bm * (size_t)S * row);
                    const float4v zero = {0.f, 0.f, 0.f, 0.f};
                    for (int q = tid; q < (S * row) >> 2; q += 256) o[q] = zero;
                }
                continue;
            }
            if (fill && tid == 0) empty_flag[bm] = 0;
            for (int q = tid; q < S; q += 256) {
                int t = q < cnt ? q : q % cnt;
                int v;
                if (t < c0) v = lg[t];
                else if ((t -= c0) < c1) v = Q + lg[S + t];
                else if ((t -= c1) < c2) v = 2 * Q + lg[2 * S + t];
                else v = 3 * Q + lg[3 * S + (t - c2)];
                sel[g * S + q] = v;
                if (pts_idx) pts_idx[bm * S + q] = v;
            }
        }
        for (int g = min(SG, nb - sub0); g < SG; ++g)
            if (tid == 0) cnt_s[g] = 0;
        __syncthreads();
    }
}

// ---- the binned variant (round 5): the scene is counting-sorted ONCE into an (x, z) grid, a box then tests only the cells its
// footprint covers.  Why: at config 5 (65536 points, 512 boxes per scene) every workgroup of the scanning kernels streams the whole
// 786 KB scene out of L2 and runs 65536 box tests per box -- 1.6 GB of L2 reads and 2 x 10^9 tests per launch, a lock-step scan
// phase of 0.26 ms beside a copy phase of 0.25 ms.  A box's enlarged footprint (~6 x 3.6 m) holds ~2,800 points of such a scene.
//   * The scene is cut into CHUNKS of ROI_CH consecutive point indices, each binned on its own (one workgroup per chunk: the points
//     stay in registers between the count and the scatter, no cross-workgroup step).  The grid covers the union of the boxes'
//     footprints only -- the boxes are an input -- and points outside it are dropped: they lie in no box.
//   * The reference's result is ORDER dependent -- the first S in-box points in point-index order, wrap-padded
//     (roipool3d_kernel.cu:97-160) -- and a cell list is not in index order, so the selection is a radix select on the index:
//       pass 1  chunk by chunk, every in-box candidate counts into one of 256 index buckets (bucket = index >> shift); chunks are
//               index ranges, so the pass ends with the chunk at which the count reaches S (the scanning kernels' early exit);
//       scan    T = the first bucket at which the running count reaches S (or the last one): buckets < T are taken whole;
//       pass 2  the candidates of buckets <= T are placed bucket by bucket (<= S - 1 + bucket width of them);
//       rank    inside its bucket every index counts the smaller ones: position = bucket offset + rank; positions < S are the list.
// Exact for any input: the in-box test is pt_in_frame itself (same arithmetic as the scanning kernels), the cells visited are a
// conservative cover of the footprint (roi_row_range), and a point's cell is computed by the same roi_coord in both kernels.

constexpr int ROI_CH = 8192;          // points per chunk
constexpr int ROI_CELLS = 4096;       // grid cells per chunk
struct RoiBinHeader { float xmin, inv_wx, zmin, inv_wz; int gx, gz, kept, pad; };    // 32 bytes

__host__ __device__ inline size_t roi_chunk_stride() { return (size_t)ROI_CH * 16 + sizeof(RoiBinHeader) + (size_t)(ROI_CELLS + 4) * sizeof(int); }
__host__ __device__ inline int roi_chunks(int n) { return (n + ROI_CH - 1) / ROI_CH; }
__host__ __device__ inline size_t roi_bin_scene_stride(int n) { return (size_t)roi_chunks(n) * roi_chunk_stride(); }

__device__ __forceinline__ int roi_coord(float v, float vmin, float inv_w, int cells) {      // monotone non-decreasing; NaN -> 0
    const float t = (v - vmin) * inv_w;
    return t > 0.f ? (t < (float)(cells - 1) ? (int)t : cells - 1) : 0;
}

// bounding interval of a box's footprint along x and z, and the slack every range of this variant carries.  An in-box point
// satisfies |x_rot| <= hl, |z_rot| <= hw up to the rounding of the test (relative 2^-22 of |dx| + |dz|); `pad` (1e-3 m + 1e-5 of
// the magnitudes) is four orders of magnitude above that.  The |dx|, |dz| <= 10 terms of the test bound a box that is not `small`.
__device__ __forceinline__ void roi_footprint(const BoxFrame &f, bool small, float &ex, float &ez, float &pad) {
    ex = fabsf(f.cosa) * f.hl + fabsf(f.sina) * f.hw;
    ez = fabsf(f.sina) * f.hl + fabsf(f.cosa) * f.hw;
    if (!small) { ex = fminf(ex, 10.0f); ez = fminf(ez, 10.0f); }
    pad = 1e-3f + 1e-5f * (fabsf(f.cx) + fabsf(f.cz) + ex + ez);
}

// one workgroup per chunk of ROI_CH points: grid over the union of the scene's box footprints, histogram, scan, scatter.
// chunk layout: [ROI_CH x {x, y, z, bits(index)}][RoiBinHeader][gx * gz + 1 cell starts]
__global__ __launch_bounds__(1024) void roi_bin_kernel(int n, int boxes_num, const float *__restrict__ xyz, const float *__restrict__ boxes3d,
                                                       char *__restrict__ ws) {
    __shared__ int hist[ROI_CELLS];
    __shared__ float red[4][16];
    __shared__ int wsum[16];
    const int ch = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    xyz += (size_t)b * n * 3;
    boxes3d += (size_t)b * boxes_num * 7;
    char *base = ws + (size_t)b * roi_bin_scene_stride(n) + (size_t)ch * roi_chunk_stride();
    float4 *sorted = reinterpret_cast<float4 *>(base);
    RoiBinHeader *hdr = reinterpret_cast<RoiBinHeader *>(base + (size_t)ROI_CH * 16);
    int *start = reinterpret_cast<int *>(base + (size_t)ROI_CH * 16 + sizeof(RoiBinHeader));
    // ---- the points of the chunk: ONE coalesced 12-byte load each, kept in registers until the scatter
    constexpr int PPT = ROI_CH / 1024;
    const int i0 = ch * ROI_CH;
    float px[PPT], py[PPT], pz[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = i0 + u * 1024 + tid;
        const f3v p = *reinterpret_cast<const f3u *>(xyz + (size_t)min(i, n - 1) * 3);
        px[u] = i < n ? p.x : __builtin_nanf(""); py[u] = p.y; pz[u] = p.z;          // NaN: dropped below
    }
    // ---- union of the footprints of the scene's boxes (every workgroup of the scene computes the same numbers)
    float lo_x = INFINITY, hi_x = -INFINITY, lo_z = INFINITY, hi_z = -INFINITY;
    for (int m = tid; m < boxes_num; m += 1024) {
        const BoxFrame f = make_frame(boxes3d + (size_t)m * 7);
        float ex, ez, pad;
        roi_footprint(f, frame_is_small(f), ex, ez, pad);
        const float x0 = f.cx - ex - 2.0f * pad, x1 = f.cx + ex + 2.0f * pad, z0 = f.cz - ez - 2.0f * pad, z1 = f.cz + ez + 2.0f * pad;
        // a box with a non-finite centre, extent or angle holds no point (its test compares NaN or inf): it claims no cell
        if (fabsf(x0) < INFINITY && fabsf(x1) < INFINITY && fabsf(z0) < INFINITY && fabsf(z1) < INFINITY) {
            lo_x = fminf(lo_x, x0); hi_x = fmaxf(hi_x, x1); lo_z = fminf(lo_z, z0); hi_z = fmaxf(hi_z, z1);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, o)); hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
        lo_z = fminf(lo_z, __shfl_xor(lo_z, o)); hi_z = fmaxf(hi_z, __shfl_xor(hi_z, o));
    }
    if (lane == 0) { red[0][w] = lo_x; red[1][w] = hi_x; red[2][w] = lo_z; red[3][w] = hi_z; }
    for (int i = tid; i < ROI_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    for (int i = 0; i < 16; ++i) {
        lo_x = fminf(lo_x, red[0][i]); hi_x = fmaxf(hi_x, red[1][i]);
        lo_z = fminf(lo_z, red[2][i]); hi_z = fmaxf(hi_z, red[3][i]);
    }
    const bool none = !(lo_x <= hi_x) || !(lo_z <= hi_z);                           // no box claims a cell: every point is dropped
    if (none) { lo_x = hi_x = lo_z = hi_z = 0.f; }
    // near-square cells, gx * gz <= ROI_CELLS; a degenerate extent gives one cell along that axis
    const float wx = hi_x - lo_x, wz = hi_z - lo_z;
    int gx = 1, gz = 1;
    if (wx > 0.f && wz > 0.f) {
        const float cs = sqrtf(wx * wz / (float)(ROI_CELLS - 128));
        gx = max(1, min(ROI_CELLS, (int)(wx / cs) + 1));
        gz = max(1, min(ROI_CELLS / gx, (int)(wz / cs) + 1));
    } else if (wx > 0.f) gx = ROI_CELLS / 4;
    else if (wz > 0.f) gz = ROI_CELLS / 4;
    const float inv_wx = wx > 0.f ? (float)gx / wx : 0.f, inv_wz = wz > 0.f ? (float)gz / wz : 0.f;
    const int cells = gx * gz;
    int cell[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const bool keep = !none && px[u] >= lo_x && px[u] <= hi_x && pz[u] >= lo_z && pz[u] <= hi_z;       // false for NaN
        cell[u] = keep ? roi_coord(pz[u], lo_z, inv_wz, gz) * gx + roi_coord(px[u], lo_x, inv_wx, gx) : -1;
        if (keep) atomicAdd(&hist[cell[u]], 1);
    }
    __syncthreads();
    // exclusive scan of hist[0 .. cells): 4 consecutive cells per thread, wave scan, 16 wave totals
    constexpr int CPT = ROI_CELLS / 1024;
    int v[CPT], run = 0;
#pragma unroll
    for (int u = 0; u < CPT; ++u) { v[u] = run; run += (tid * CPT + u < cells) ? hist[tid * CPT + u] : 0; }
    int inc = run;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int woff = 0, total = 0;
    for (int i = 0; i < 16; ++i) { if (i < w) woff += wsum[i]; total += wsum[i]; }
    const int excl = woff + inc - run;
#pragma unroll
    for (int u = 0; u < CPT; ++u)
        if (tid * CPT + u < cells) { hist[tid * CPT + u] = excl + v[u]; start[tid * CPT + u] = excl + v[u]; }
    if (tid == 0) {
        start[cells] = total;
        RoiBinHeader h; h.xmin = lo_x; h.inv_wx = inv_wx; h.zmin = lo_z; h.inv_wz = inv_wz; h.gx = gx; h.gz = gz; h.kept = total; h.pad = 0;
        *hdr = h;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PPT; ++u)
        if (cell[u] >= 0) sorted[atomicAdd(&hist[cell[u]], 1)] = make_float4(px[u], py[u], pz[u], __int_as_float(i0 + u * 1024 + tid));
}

// the candidate range of grid row r for a box frame: cells [c0, c1] of that row, or c0 > c1 when the row cannot hold an in-box point.
// Conservative by construction: an in-box point's dx, dz lie in the rectangle inflated by `pad` (roi_footprint); the row's z-slab is
// widened by `slack` cells on both sides against the rounding of roi_coord's own product (the first / last row reach to -inf / +inf:
// roi_coord clamps); each of the rectangle's two linear constraints gives, over the slab, a lower / upper bound on dx at an end of
// the slab -- the larger of the lower ones, the smaller of the upper ones -- inside the bounding interval [-ex, ex]; roi_coord is
// monotone, so every x inside [cx + lo, cx + hi] falls into a cell of [c0, c1].
__device__ __forceinline__ void roi_row_range(const BoxFrame &f, const RoiBinHeader &h, bool small, float ex, float pad, int r, int &c0, int &c1) {
    float lo = -ex - pad, hi = ex + pad;
    if (small) {       // (a box that is not `small` -- BEV half-diagonal >= 9.9 m, or non-finite extents -- keeps the bounding interval)
        const float cw = h.inv_wz > 0.f ? 1.0f / h.inv_wz : INFINITY;
        const float slack = 0.01f + 2e-6f * (float)r;            // cells: above the rounding of roi_coord's product for any row index
        float za = r == 0 ? -INFINITY : h.zmin + ((float)r - slack) * cw - f.cz - pad;
        float zb = r == h.gz - 1 ? INFINITY : h.zmin + ((float)r + 1.0f + slack) * cw - f.cz + pad;
        const float c = f.cosa, s = f.sina, hl = f.hl + pad, hw = f.hw + pad;
        const float ez = fabsf(s) * hl + fabsf(c) * hw + pad;   // the slab clipped to the footprint's own z-extent: finite products
        za = fmaxf(za, -ez); zb = fminf(zb, ez);
        if (!(za <= zb)) { c0 = 1; c1 = 0; return; }
        if (fabsf(c) > 1e-3f) {        // |dx c - dz s| <= hl  ->  dx c in [-hl + dz s, hl + dz s]
            const float a0 = (-hl + za * s) / c, a1 = (-hl + zb * s) / c, b0 = (hl + za * s) / c, b1 = (hl + zb * s) / c;
            lo = fmaxf(lo, c > 0.f ? fminf(a0, a1) : fminf(b0, b1));
            hi = fminf(hi, c > 0.f ? fmaxf(b0, b1) : fmaxf(a0, a1));
        }
        if (fabsf(s) > 1e-3f) {        // |dx s + dz c| <= hw  ->  dx s in [-hw - dz c, hw - dz c]
            const float a0 = (-hw - za * c) / s, a1 = (-hw - zb * c) / s, b0 = (hw - za * c) / s, b1 = (hw - zb * c) / s;
            lo = fmaxf(lo, s > 0.f ? fminf(a0, a1) : fminf(b0, b1));
            hi = fminf(hi, s > 0.f ? fmaxf(b0, b1) : fmaxf(a0, a1));
        }
        if (!(lo <= hi)) { c0 = 1; c1 = 0; return; }
    }
    c0 = roi_coord(f.cx + lo - pad, h.xmin, h.inv_wx, h.gx);
    c1 = roi_coord(f.cx + hi + pad, h.xmin, h.inv_wx, h.gx);
}

constexpr int ROI_ROWS_MAX = 256;      // grid rows of one box with a cell range of their own (more: the box takes every kept point of a chunk)

template <int CR>
__global__ __launch_bounds__(256) void roipool3d_binned_kernel(int pts_num, int boxes_num, int feat_len, int S, int shift,
                                                               const float *__restrict__ xyz, const char *__restrict__ ws,
                                                               const float *__restrict__ boxes3d, const float *__restrict__ pts_feature,
                                                               float *__restrict__ pooled, int32_t *__restrict__ empty_flag,
                                                               int32_t *__restrict__ pts_idx, int fill, int batch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int row = 3 + feat_len;
    float *stage = reinterpret_cast<float *>(smem);                                 // 2 * CR * row
    int *sel = reinterpret_cast<int *>(stage + 2 * CR * row);                       // S
    int *G = sel + S;                                                               // S + (1 << shift)
    __shared__ int hist[256], pref[257], rc0[ROI_ROWS_MAX], rc1[ROI_ROWS_MAX], rbeg[ROI_ROWS_MAX], rend[ROI_ROWS_MAX];
    __shared__ int s_K, s_T, s_cnt;
    int b, m;
    roi_scene_group(batch, boxes_num, b, m);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t bm = (size_t)b * boxes_num + m;
    xyz += (size_t)b * pts_num * 3;
    pts_feature += (size_t)b * pts_num * feat_len;
    const char *scene = ws + (size_t)b * roi_bin_scene_stride(pts_num);
    const RoiBinHeader h = *reinterpret_cast<const RoiBinHeader *>(scene + (size_t)ROI_CH * 16);      // the grid is the same in every chunk of a scene
    const int chunks = roi_chunks(pts_num);
    const BoxFrame f = make_frame(boxes3d + bm * 7);
    const bool small = frame_is_small(f);
    float ex, ez, pad;
    roi_footprint(f, small, ex, ez, pad);
    const int r0 = roi_coord(f.cz - ez - 2.0f * pad, h.zmin, h.inv_wz, h.gz), r1 = roi_coord(f.cz + ez + 2.0f * pad, h.zmin, h.inv_wz, h.gz);
    const bool whole = r1 - r0 + 1 > ROI_ROWS_MAX;                // a box over more rows than the table holds: every kept point is a candidate
    const int nrows = whole ? 1 : r1 - r0 + 1;
    hist[tid] = 0;
    if (tid == 0) s_cnt = 0;
    if (!whole)
        for (int r = tid; r < nrows; r += 256) {
            int c0, c1;
            roi_row_range(f, h, small, ex, pad, r0 + r, c0, c1);
            rc0[r] = c0; rc1[r] = c1;
        }
    __syncthreads();
    // the segments (row ranges) of one chunk -> rbeg / rend
    auto segments = [&](int ch) {
        const int *start = reinterpret_cast<const int *>(scene + (size_t)ch * roi_chunk_stride() + (size_t)ROI_CH * 16 + sizeof(RoiBinHeader));
        if (whole) { if (tid == 0) { rbeg[0] = 0; rend[0] = start[h.gx * h.gz]; } }
        else
            for (int r = tid; r < nrows; r += 256) {
                const int c0 = rc0[r], c1 = rc1[r];
                const bool any = c0 <= c1;
                rbeg[r] = any ? start[(r0 + r) * h.gx + c0] : 0;
                rend[r] = any ? start[(r0 + r) * h.gx + c1 + 1] : 0;
            }
    };
    auto sweep = [&](int ch, auto &&hit) {
        const float4 *sorted = reinterpret_cast<const float4 *>(scene + (size_t)ch * roi_chunk_stride());
        for (int r = w; r < nrows; r += 4) {
            const int e = rend[r];
            for (int i = rbeg[r] + lane; i < e; i += 64) {
                const float4 p = sorted[i];
                const bool in = small ? pt_in_frame<true>(f, p.x, p.y, p.z) : pt_in_frame<false>(f, p.x, p.y, p.z);
                if (in) hit(__float_as_int(p.w));
            }
        }
    };
    // ---- pass 1: chunk by chunk (ascending index ranges), in-box candidates count into their index bucket, until S are found
    int used = 0;
    for (int ch = 0; ch < chunks; ++ch) {
        segments(ch);
        __syncthreads();
        int mine = 0;
        sweep(ch, [&](int k) { atomicAdd(&hist[k >> shift], 1); ++mine; });
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
        if (lane == 0 && mine) atomicAdd(&s_cnt, mine);
        __syncthreads();
        used = ch + 1;
        if (s_cnt >= S) break;                                     // workgroup-uniform
    }
    if (w == 0) {          // inclusive scan of the 256 buckets (4 per lane), T = first bucket whose running count reaches S
        const int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
        const int tot = h0 + h1 + h2 + h3;
        int inc = tot;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        const int ex0 = inc - tot;
        pref[4 * lane] = ex0; pref[4 * lane + 1] = ex0 + h0; pref[4 * lane + 2] = ex0 + h0 + h1; pref[4 * lane + 3] = ex0 + h0 + h1 + h2;
        if (lane == 63) { pref[256] = inc; s_K = inc; }
        const uint64_t reach = __builtin_amdgcn_ballot_w64(inc >= S);
        if (reach == 0) { if (lane == 0) s_T = 255; }
        else if (lane == __builtin_ctzll(reach)) {
            int T = 4 * lane + 3;
            if (ex0 + h0 >= S) T = 4 * lane; else if (ex0 + h0 + h1 >= S) T = 4 * lane + 1; else if (ex0 + h0 + h1 + h2 >= S) T = 4 * lane + 2;
            s_T = T;
        }
    }
    __syncthreads();
    const int K = s_K, T = s_T;
    if (K == 0) {      // roipool3d_kernel.cu:147-149,181-183: flag the box, leave its rows untouched
        if (tid == 0) empty_flag[bm] = 1;
        if (pts_idx)
            for (int q = tid; q < S; q += 256) pts_idx[bm * S + q] = 0;
        if (fill) {
            float4v *o = reinterpret_cast<float4v *>(pooled + bm * (size_t)S * row);
            const float4v zero = {0.f, 0.f, 0.f, 0.f};
            for (int q = tid; q < (S * row) >> 2; q += 256) o[q] = zero;
        }
        return;
    }
    if (fill && tid == 0) empty_flag[bm] = 0;
    hist[tid] = pref[tid];                                        // cursors
    __syncthreads();
    // ---- pass 2: the candidates of buckets <= T, placed bucket by bucket (chunks past the bucket's index range hold none)
    const int last_idx = ((T + 1) << shift) - 1;
    for (int ch = 0; ch < used && ch * ROI_CH <= last_idx; ++ch) {
        if (used > 1 || ch > 0) { __syncthreads(); segments(ch); __syncthreads(); }       // (one chunk used: its segments are still in place)
        sweep(ch, [&](int k) { const int bk = k >> shift; if (bk <= T) G[atomicAdd(&hist[bk], 1)] = k; });
    }
    __syncthreads();
    // ---- rank inside the bucket -> the list in index order; then wrap-pad (duplicate_idx = k % cnt, roipool3d_kernel.cu:153-157)
    const int taken = pref[T + 1], cnt = min(K, S);
    for (int p = tid; p < taken; p += 256) {
        const int v = G[p], bk = v >> shift, lo = pref[bk], hi = pref[bk + 1];
        int rank = 0;
        for (int q = lo; q < hi; ++q) rank += G[q] < v ? 1 : 0;
        if (lo + rank < S) sel[lo + rank] = v;
    }
    __syncthreads();
    for (int q = cnt + tid; q < S; q += 256) sel[q] = sel[q % cnt];
    __syncthreads();
    if (pts_idx)
        for (int q = tid; q < S; q += 256) pts_idx[bm * S + q] = sel[q];
    // ---- copy: the S x (3+C) block through LDS in stretches of CR rows (see roipool3d_kernel's copy phase)
    float *out = pooled + bm * (size_t)S * row;
    const int half = tid >> 5, l32 = tid & 31, f4 = feat_len >> 2;
    constexpr int RPH = CR / 8;
    for (int c0 = 0, it = 0; c0 < S; c0 += CR, ++it) {
        float *st = stage + (it & 1) * CR * row;
        const int nr = min(CR, S - c0);
        float4v v[RPH];
        float p3[RPH];
#pragma unroll
        for (int u = 0; u < RPH; ++u) {
            const int src = sel[min(c0 + half * RPH + u, S - 1)];
            if (l32 < f4) v[u] = reinterpret_cast<const float4v *>(pts_feature + (size_t)src * feat_len)[l32];
            if (l32 < 3) p3[u] = xyz[(size_t)src * 3 + l32];
        }
#pragma unroll
        for (int u = 0; u < RPH; ++u) {
            const int r = half * RPH + u;
            if (r < nr) {
                if (l32 < f4) *reinterpret_cast<float4u *>(st + r * row + 3 + 4 * l32) = v[u];
                if (l32 < 3) st[r * row + l32] = p3[u];
            }
        }
        __syncthreads();
        float4v *o4 = reinterpret_cast<float4v *>(out + (size_t)c0 * row);
        const int units = (nr * row) >> 2;
        for (int q = tid; q < units; q += 256)
            __builtin_nontemporal_store(*reinterpret_cast<const float4v *>(st + 4 * q), o4 + q);
    }
}

__global__ __launch_bounds__(256) void pts_in_boxes3d_kernel(int boxes_num, int pts_num,
                                                             const float *__restrict__ pts,
                                                             const float *__restrict__ boxes3d,
                                                             int64_t *__restrict__ flag) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    const BoxFrame f = make_frame(boxes3d + (size_t)i * 7);
    if (j >= pts_num) return;
    const float *p = pts + (size_t)j * 3;
    flag[(size_t)i * pts_num + j] = pt_in_frame(f, p[0], p[1], p[2]) ? 1 : 0;
}

}  // namespace ws3d

// scenes the binned variant takes (bytes of its workspace, 0: not applicable): large enough that the grid pays, indices that fit
// 256 buckets of <= 1024
static size_t roi_binned_bytes(int batch_size, int pts_num) {
    static const int env = getenv("WS3D_ROI_BINNED") ? atoi(getenv("WS3D_ROI_BINNED")) : -1;     // 0: off; N > 0: from N points on (A/B runs)
    const int min_n = env > 0 ? env : 16384;      // c3 (16384 points, 100 boxes): +1 % of the step; c5 (65536, 512): 0.365 -> 0.265 ms
    if (env == 0 || pts_num < min_n || pts_num > 256 * 1024 || batch_size <= 0) return 0;
    return (size_t)batch_size * ws3d::roi_bin_scene_stride(pts_num);
}

extern "C" size_t ws3d_roipool3d_workspace_bytes(int batch_size, int pts_num) { return roi_binned_bytes(batch_size, pts_num); }

static int roipool3d_launch(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                            int sampled_pts_num, const float *xyz, const float *boxes3d,
                            const float *pts_feature, float *pooled_features,
                            int32_t *pooled_empty_flag, int32_t *pts_idx, int fill, ws3d_stream_t stream,
                            void *workspace = nullptr, size_t workspace_bytes = 0) {
    using namespace ws3d;
    if (batch_size < 0 || pts_num < 0 || boxes_num < 0 || feature_in_len < 0 || sampled_pts_num <= 0 ||
        !xyz || !boxes3d || (!pts_feature && feature_in_len > 0) || !pooled_features || !pooled_empty_flag) {
        set_error("ws3d_roipool3d: invalid argument (B=%d N=%d M=%d C=%d S=%d)", batch_size, pts_num,
                  boxes_num, feature_in_len, sampled_pts_num);
        return WS3D_E_INVALID;
    }
    if (batch_size == 0 || boxes_num == 0) return WS3D_OK;
    // boxes per workgroup: sharing the scene scan among 4 boxes pays when that still leaves >= 2
    // workgroups per CU (WS3D_ROI_BG overrides for A/B runs)
    static const int bg_env = getenv("WS3D_ROI_BG") ? atoi(getenv("WS3D_ROI_BG")) : 0;
    static const int cr_env = getenv("WS3D_ROI_STAGE") ? atoi(getenv("WS3D_ROI_STAGE")) : -1;   // 0 / 16 / 32: A/B runs
    int bg = ((long)batch_size * ((boxes_num + 3) / 4) >= 512 && sampled_pts_num <= 1024) ? 4 : 1;
    if (bg_env == 1 || bg_env == 2 || bg_env == 4) bg = bg_env;
    const int row = 3 + feature_in_len;
    // rows staged through LDS by the copy phase (the aligned-store path): what its layout needs, else the direct copy
    int cr = (feature_in_len >= 4 && (feature_in_len & 3) == 0 && feature_in_len <= 128 && ((long)sampled_pts_num * row) % 4 == 0 &&
              ((reinterpret_cast<uintptr_t>(pts_feature) | reinterpret_cast<uintptr_t>(pooled_features)) & 15) == 0) ? 32 : 0;
    if (cr && (cr_env == 0 || cr_env == 16 || cr_env == 32)) cr = cr_env;
    const size_t lists_ints = ((size_t)(bg * 4 + 1) * (size_t)sampled_pts_num + 3) & ~(size_t)3;
    const size_t smem = sizeof(int) * lists_ints + sizeof(float) * 2 * (size_t)cr * row;
    if (smem > 150 * 1024 || batch_size > 65535) {
        set_error("ws3d_roipool3d: sampled_pts_num=%d / batch=%d unsupported", sampled_pts_num, batch_size);
        return WS3D_E_UNSUPPORTED;
    }
    // large scenes with a workspace: bin the scene once, every box tests the cells under its footprint only
    const size_t need = roi_binned_bytes(batch_size, pts_num);
    if (workspace && need > 0 && workspace_bytes >= need && cr > 0 && (sampled_pts_num & 3) == 0 && sampled_pts_num <= 2048 &&
        (long)batch_size * boxes_num < (1L << 31) && (reinterpret_cast<uintptr_t>(workspace) & 15) == 0) {
        int shift = 0;
        while ((256 << shift) < pts_num) ++shift;
        const int crb = 16;
        const size_t lds = sizeof(float) * 2 * (size_t)crb * row + sizeof(int) * ((size_t)2 * sampled_pts_num + ((size_t)1 << shift));
        hipLaunchKernelGGL(roi_bin_kernel, dim3(roi_chunks(pts_num), batch_size), dim3(1024), 0, as_stream(stream), pts_num, boxes_num, xyz, boxes3d,
                           reinterpret_cast<char *>(workspace));
        if (int rc = raise_lds_cap((const void *)roipool3d_binned_kernel<16>, lds, "ws3d_roipool3d")) return rc;
        hipLaunchKernelGGL((roipool3d_binned_kernel<16>), dim3((unsigned)((long)batch_size * boxes_num)), dim3(256), lds, as_stream(stream),
                           pts_num, boxes_num, feature_in_len, sampled_pts_num, shift, xyz, reinterpret_cast<const char *>(workspace), boxes3d,
                           pts_feature, pooled_features, pooled_empty_flag, pts_idx, fill, batch_size);
        return check_launch("ws3d_roipool3d(binned)");
    }
    // large scenes, 4 boxes per workgroup: the variant that overlaps scan and copy inside the workgroup
    static const int pipe_env = getenv("WS3D_ROI_PIPE") ? atoi(getenv("WS3D_ROI_PIPE")) : -1;   // 0: off; 1 / 2: boxes per pass
    const int quarter = (((pts_num + 3) / 4 + 63) / 64) * 64;
    if (bg == 4 && cr > 0 && pipe_env != 0 && quarter <= 65536 && (sampled_pts_num & 3) == 0 && pts_num >= 16 * 1024) {
        const int sg = pipe_env == 1 ? 1 : 2;
        if (cr_env != 16 && cr_env != 32) cr = 16;
        const size_t pm = sizeof(float) * 2 * (size_t)cr * row + sizeof(int) * (size_t)sg * sampled_pts_num +
                          sizeof(uint16_t) * (size_t)sg * 4 * sampled_pts_num;
#define WS3D_ROI_PIPE_LAUNCH(SGV, CRV)                                                                                            \
        {                                                                                                                         \
            if (int rc = raise_lds_cap((const void *)roipool3d_pipe_kernel<SGV, CRV>, pm, "ws3d_roipool3d")) return rc;            \
            hipLaunchKernelGGL((roipool3d_pipe_kernel<SGV, CRV>), dim3((unsigned)(((boxes_num + 3) / 4) * batch_size)), dim3(256), pm, as_stream(stream), \
                               pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature, pooled_features,   \
                               pooled_empty_flag, pts_idx, fill, batch_size);                                                                 \
        }
        if (sg == 1) { if (cr == 16) WS3D_ROI_PIPE_LAUNCH(1, 16) else WS3D_ROI_PIPE_LAUNCH(1, 32) }
        else { if (cr == 16) WS3D_ROI_PIPE_LAUNCH(2, 16) else WS3D_ROI_PIPE_LAUNCH(2, 32) }
#undef WS3D_ROI_PIPE_LAUNCH
        return check_launch("ws3d_roipool3d");
    }
#define WS3D_ROI_LAUNCH(BGV, CRV)                                                                                  \
    {                                                                                                              \
        if (int rc = raise_lds_cap((const void *)roipool3d_kernel<BGV, CRV>, smem, "ws3d_roipool3d")) return rc;   \
        hipLaunchKernelGGL((roipool3d_kernel<BGV, CRV>), dim3((unsigned)(((boxes_num + BGV - 1) / BGV) * batch_size)), dim3(256), smem, \
                           as_stream(stream), pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d,   \
                           pts_feature, pooled_features, pooled_empty_flag, pts_idx, fill, batch_size);                        \
    }
#define WS3D_ROI_LAUNCH_BG(CRV) { if (bg == 4) WS3D_ROI_LAUNCH(4, CRV) else if (bg == 2) WS3D_ROI_LAUNCH(2, CRV) else WS3D_ROI_LAUNCH(1, CRV) }
    if (cr == 32) WS3D_ROI_LAUNCH_BG(32) else if (cr == 16) WS3D_ROI_LAUNCH_BG(16) else WS3D_ROI_LAUNCH_BG(0)
#undef WS3D_ROI_LAUNCH_BG
#undef WS3D_ROI_LAUNCH
    return check_launch("ws3d_roipool3d");
}

extern "C" int ws3d_roipool3d(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                              int sampled_pts_num, const float *xyz, const float *boxes3d,
                              const float *pts_feature, float *pooled_features,
                              int32_t *pooled_empty_flag, int32_t *pts_idx, ws3d_stream_t stream) {
    return roipool3d_launch(batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature,
                            pooled_features, pooled_empty_flag, pts_idx, 0, stream);
}

extern "C" int ws3d_roipool3d_fill(int batch_size, int pts_num, int boxes_num, int feature_in_len,
                                   int sampled_pts_num, const float *xyz, const float *boxes3d,
                                   const float *pts_feature, float *pooled_features,
                                   int32_t *pooled_empty_flag, int32_t *pts_idx, ws3d_stream_t stream) {
    return roipool3d_launch(batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature,
                            pooled_features, pooled_empty_flag, pts_idx, 1, stream);
}

extern "C" int ws3d_roipool3d_ws(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num, const float *xyz,
                                 const float *boxes3d, const float *pts_feature, float *pooled_features, int32_t *pooled_empty_flag,
                                 int32_t *pts_idx, int fill, void *workspace, size_t workspace_bytes, ws3d_stream_t stream) {
    return roipool3d_launch(batch_size, pts_num, boxes_num, feature_in_len, sampled_pts_num, xyz, boxes3d, pts_feature,
                            pooled_features, pooled_empty_flag, pts_idx, fill ? 1 : 0, stream, workspace, workspace_bytes);
}

extern "C" int ws3d_pts_in_boxes3d(int boxes_num, int pts_num, const float *pts, const float *boxes3d,
                                   int64_t *flag, ws3d_stream_t stream) {
    using namespace ws3d;
    if (boxes_num < 0 || pts_num < 0 || !pts || !boxes3d || !flag) {
        set_error("ws3d_pts_in_boxes3d: invalid argument (M=%d N=%d)", boxes_num, pts_num);
        return WS3D_E_INVALID;
    }
    if (boxes_num == 0 || pts_num == 0) return WS3D_OK;
    if (boxes_num > 65535) { set_error("ws3d_pts_in_boxes3d: boxes_num > 65535"); return WS3D_E_UNSUPPORTED; }
    hipLaunchKernelGGL(pts_in_boxes3d_kernel, dim3((pts_num + 255) / 256, boxes_num), dim3(256), 0,
                       as_stream(stream), boxes_num, pts_num, pts, boxes3d, flag);
    return check_launch("ws3d_pts_in_boxes3d");
}
