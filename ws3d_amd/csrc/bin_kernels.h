// The two per-scene binning passes (binning.h) as device functions: one workgroup of 1024 threads bins scene `b` of a (B, n, 3) cloud.
// ws3d_sort_points_grid / ws3d_sort_points_xz launch them as kernels of their own; ws3d_sort_points_jobs (ballquery_group.hip) runs
// several (cloud, flavour) jobs -- the levels of a network, both flavours -- in ONE launch, blockIdx.y = the job.
#pragma once
#include "common.h"
#include "binning.h"

namespace ws3d {

// ---- fine (x, z) grid flavour (binning.h) for the ball query.  One workgroup per scene: bounding box of the finite
// (x, z), a gx x gz grid of near-square cells (gx * gz <= GRID16_CELLS, ~0.5 points per cell), counting sort with 16-bit
// counters packed two per LDS word (a cell holds < 65536 points, so the halves never carry into each other).
static __device__ __forceinline__ void bin_points_grid_body(const int b, int n, const float *__restrict__ xyz, char *__restrict__ ws) {
    // 64 KB + one pad word per 16: thread t scans words 16 t .. 16 t + 15, stored at 17 t + i -- conflict-free (unpadded, the
    // 64 lanes of a wave hit two banks: 16-way conflicts on every access of the scan)
    __shared__ __attribute__((aligned(16))) unsigned hist[GRID16_CELLS / 2 + GRID16_CELLS / 32];
#define HW(wi) ((wi) + ((wi) >> 4))
    __shared__ int wsum[16];
    __shared__ float red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    xyz += (size_t)b * n * 3;
    char *base = ws + (size_t)b * bin_scene_stride(n);
    float4 *sorted = reinterpret_cast<float4 *>(base);
    BinHeader *hdr = reinterpret_cast<BinHeader *>(base + (size_t)n * 16);
    uint16_t *start = reinterpret_cast<uint16_t *>(base + (size_t)n * 16 + sizeof(BinHeader));
    int *params = reinterpret_cast<int *>(base + (size_t)n * 16 + sizeof(BinHeader) + GRID16_PARAMS);

    // the thread's <= 16 points (n <= 16384) are loaded ONCE, 16 independent 12-byte loads in flight, and stay in registers
    // for the three passes (bounding box, histogram, scatter): with a load per pass and point the kernel was a chain of
    // ~48 dependent memory round trips (47 us for one scene)
    typedef float f3v __attribute__((ext_vector_type(3)));
    typedef f3v f3u __attribute__((aligned(4)));
    f3v pt[SORT_MAX_N / 1024];
#pragma unroll
    for (int s = 0; s < SORT_MAX_N / 1024; ++s) pt[s] = *reinterpret_cast<const f3u *>(xyz + (size_t)min(tid + 1024 * s, n - 1) * 3);
    float lo_x = INFINITY, hi_x = -INFINITY, lo_z = INFINITY, hi_z = -INFINITY;
#pragma unroll
    for (int s = 0; s < SORT_MAX_N / 1024; ++s) {
        if (tid + 1024 * s >= n) continue;
        const float x = pt[s].x, z = pt[s].z;
        if (fabsf(x) < INFINITY) { lo_x = fminf(lo_x, x); hi_x = fmaxf(hi_x, x); }
        if (fabsf(z) < INFINITY) { lo_z = fminf(lo_z, z); hi_z = fmaxf(hi_z, z); }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, o)); hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
        lo_z = fminf(lo_z, __shfl_xor(lo_z, o)); hi_z = fmaxf(hi_z, __shfl_xor(hi_z, o));
    }
    if (lane == 0) { red[0][w] = lo_x; red[1][w] = hi_x; red[2][w] = lo_z; red[3][w] = hi_z; }
    for (int i = tid; i < GRID16_CELLS / 2 + GRID16_CELLS / 32; i += 1024) hist[i] = 0u;
    __syncthreads();
    lo_x = red[0][0]; hi_x = red[1][0]; lo_z = red[2][0]; hi_z = red[3][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        lo_x = fminf(lo_x, red[0][i]); hi_x = fmaxf(hi_x, red[1][i]);
        lo_z = fminf(lo_z, red[2][i]); hi_z = fmaxf(hi_z, red[3][i]);
    }
    const float xmin = lo_x <= hi_x ? lo_x : 0.f, wx = lo_x <= hi_x ? hi_x - lo_x : 0.f;
    const float zmin = lo_z <= hi_z ? lo_z : 0.f, wz = lo_z <= hi_z ? hi_z - lo_z : 0.f;
    int gx = 1, gz = 1;
    const int target = max(1, min(GRID16_CELLS, 2 * n));
    if (wx > 0.f && wz > 0.f) {
        const float h = sqrtf(wx * wz / (float)target);
        gx = max(1, min(GRID16_CELLS, (int)ceilf(wx / h)));
        gz = max(1, min(GRID16_CELLS / gx, (int)ceilf(wz / h)));
    } else if (wx > 0.f) {
        gx = target;
    } else if (wz > 0.f) {
        gz = target;
    }
    const int ncell = gx * gz;
    const float inv_wx = wx > 0.f ? (float)gx / wx : 0.f, inv_wz = wz > 0.f ? (float)gz / wz : 0.f;
    auto cell_of = [&](const f3v p) { return grid_coord(p.z, zmin, inv_wz, gz) * gx + grid_coord(p.x, xmin, inv_wx, gx); };
#pragma unroll
    for (int s = 0; s < SORT_MAX_N / 1024; ++s) {
        if (tid + 1024 * s >= n) continue;
        const int c = cell_of(pt[s]);
        atomicAdd(&hist[HW(c >> 1)], 1u << (16 * (c & 1)));
    }
    __syncthreads();
    // exclusive scan over 32768 16-bit counters: 16 words (32 cells) per thread
    unsigned wd[16];
    int v = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { wd[i] = hist[tid * 17 + i]; v += (int)(wd[i] & 0xffffu) + (int)(wd[i] >> 16); }
    const int mine = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int off = 0;
    for (int i = 0; i < w; ++i) off += wsum[i];
    int run = off + v - mine;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c0 = (tid * 16 + i) * 2;
        const unsigned lo = (unsigned)run;
        run += (int)(wd[i] & 0xffffu);
        const unsigned hi = (unsigned)run;
        run += (int)(wd[i] >> 16);
        hist[tid * 17 + i] = lo | (hi << 16);           // running scatter cursors (each < 65536)
        if (c0 <= ncell) start[c0] = (uint16_t)lo;
        if (c0 + 1 <= ncell) start[c0 + 1] = (uint16_t)hi;
    }
    if (tid == 0) {
        hdr->xmin = xmin; hdr->inv_w = inv_wx; hdr->n = n; hdr->pad = -gx;
        params[0] = __float_as_int(zmin); params[1] = __float_as_int(inv_wz); params[2] = gz;
    }
    if (tid == 1023 && ncell == GRID16_CELLS) start[GRID16_CELLS] = (uint16_t)n;   // the sentinel behind a full table
    __syncthreads();
    // every point's slot first (registers), then the sorted records go out through LDS in stretches of 4096 slots -- the
    // counters' memory, free once the slots are known -- as whole 64 KB runs.  Scattered straight from the registers the 16-byte
    // records reach HBM as partial lines that are evicted and re-merged: 402 MB written per 512 scenes for 134 MB of records
    // (rocprofv3 WRITE_SIZE), and the kernel ran at that traffic's speed
    unsigned slot2[SORT_MAX_N / 2048];                            // two 16-bit slots per register (0xffff: no point)
#pragma unroll
    for (int s = 0; s < SORT_MAX_N / 1024; ++s) {
        unsigned sl = 0xffffu;
        if (tid + 1024 * s < n) {
            const int c = cell_of(pt[s]);
            const unsigned old = atomicAdd(&hist[HW(c >> 1)], 1u << (16 * (c & 1)));
            sl = (old >> (16 * (c & 1))) & 0xffffu;
        }
        slot2[s >> 1] = (s & 1) ? (slot2[s >> 1] | (sl << 16)) : sl;
    }
#undef HW
    __syncthreads();
    float4 *stage = reinterpret_cast<float4 *>(hist);             // 4352 records fit, 4096 used
    for (int q0 = 0; q0 < n; q0 += 4096) {
#pragma unroll
        for (int s = 0; s < SORT_MAX_N / 1024; ++s) {
            const unsigned rel = ((slot2[s >> 1] >> (16 * (s & 1))) & 0xffffu) - (unsigned)q0;
            if (rel < 4096u) stage[rel] = make_float4(pt[s].x, pt[s].y, pt[s].z, __int_as_float(tid + 1024 * s));
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = q0 + tid + 1024 * j;
            if (i < n) sorted[i] = stage[tid + 1024 * j];
        }
        __syncthreads();
    }
}

// ---- (x, z) grid flavour of the binned known set (binning.h), for the 3-NN search only ----
// One workgroup per scene: bounding box of the finite (x, z), a gx x gz grid of near-square cells with
// ~2 points each (gx * gz <= BQS_CELLS), LDS histogram, exclusive scan, scatter.
static __device__ __forceinline__ void bin_points_xz_body(const int b, int n, const float *__restrict__ xyz, char *__restrict__ ws) {
    __shared__ int hist[BQS_CELLS];
    __shared__ int wsum[16];
    __shared__ float red[4][16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    xyz += (size_t)b * n * 3;
    char *base = ws + (size_t)b * bin_scene_stride(n);
    float4 *sorted = reinterpret_cast<float4 *>(base);
    BinHeader *hdr = reinterpret_cast<BinHeader *>(base + (size_t)n * 16);
    int *start = reinterpret_cast<int *>(base + (size_t)n * 16 + sizeof(BinHeader));

    float lo_x = INFINITY, hi_x = -INFINITY, lo_z = INFINITY, hi_z = -INFINITY;
    for (int i = tid; i < n; i += 1024) {
        const float x = xyz[(size_t)i * 3], z = xyz[(size_t)i * 3 + 2];
        if (fabsf(x) < INFINITY) { lo_x = fminf(lo_x, x); hi_x = fmaxf(hi_x, x); }
        if (fabsf(z) < INFINITY) { lo_z = fminf(lo_z, z); hi_z = fmaxf(hi_z, z); }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo_x = fminf(lo_x, __shfl_xor(lo_x, o)); hi_x = fmaxf(hi_x, __shfl_xor(hi_x, o));
        lo_z = fminf(lo_z, __shfl_xor(lo_z, o)); hi_z = fmaxf(hi_z, __shfl_xor(hi_z, o));
    }
    if (lane == 0) { red[0][w] = lo_x; red[1][w] = hi_x; red[2][w] = lo_z; red[3][w] = hi_z; }
    for (int i = tid; i < BQS_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    lo_x = red[0][0]; hi_x = red[1][0]; lo_z = red[2][0]; hi_z = red[3][0];
#pragma unroll
    for (int i = 1; i < 16; ++i) {
        lo_x = fminf(lo_x, red[0][i]); hi_x = fmaxf(hi_x, red[1][i]);
        lo_z = fminf(lo_z, red[2][i]); hi_z = fmaxf(hi_z, red[3][i]);
    }
    const float xmin = lo_x <= hi_x ? lo_x : 0.f, wx = lo_x <= hi_x ? hi_x - lo_x : 0.f;
    const float zmin = lo_z <= hi_z ? lo_z : 0.f, wz = lo_z <= hi_z ? hi_z - lo_z : 0.f;
    // near-square cells, about two points each; a degenerate extent gets one cell along that axis
    int gx = 1, gz = 1;
    const int target = max(1, min(BQS_CELLS, n / 2));
    if (wx > 0.f && wz > 0.f) {
        const float h = sqrtf(wx * wz / (float)target);
        gx = max(1, min(BQS_CELLS, (int)ceilf(wx / h)));
        gz = max(1, min(BQS_CELLS / gx, (int)ceilf(wz / h)));
    } else if (wx > 0.f) {
        gx = target;
    } else if (wz > 0.f) {
        gz = target;
    }
    const float inv_wx = wx > 0.f ? (float)gx / wx : 0.f, inv_wz = wz > 0.f ? (float)gz / wz : 0.f;
    auto cell_of = [&](const float *p) { return grid_coord(p[2], zmin, inv_wz, gz) * gx + grid_coord(p[0], xmin, inv_wx, gx); };
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[cell_of(xyz + (size_t)i * 3)], 1);
    __syncthreads();
    const int a0 = hist[2 * tid], a1 = hist[2 * tid + 1];
    int v = a0 + a1;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int off = 0;
    for (int i = 0; i < w; ++i) off += wsum[i];
    const int excl = off + v - (a0 + a1);
    __syncthreads();
    hist[2 * tid] = excl;
    hist[2 * tid + 1] = excl + a0;
    start[2 * tid] = excl;
    start[2 * tid + 1] = excl + a0;
    if (tid == 0) {
        start[BQS_CELLS] = n;
        hdr->xmin = xmin; hdr->inv_w = inv_wx; hdr->n = n; hdr->pad = gx;
        start[GRID_ZMIN] = __float_as_int(zmin); start[GRID_INV_WZ] = __float_as_int(inv_wz); start[GRID_GZ] = gz;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const float *p = xyz + (size_t)i * 3;
        const int pos = atomicAdd(&hist[cell_of(p)], 1);
        sorted[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
    }
}

struct BinJob { const float *xyz; char *ws; int n; int kind; };      // kind 0: fine grid (ball query), 1: (x, z) grid (three_nn)
constexpr int BIN_MAX_JOBS = 8;
struct BinJobs { BinJob j[BIN_MAX_JOBS]; };

}  // namespace ws3d
