// interpolate.hip -- three_nn / three_interpolate (+grad) for gfx950.  Replaces
// pointnet2_cuda.{three_nn_wrapper, three_interpolate_wrapper,
// three_interpolate_grad_wrapper} (interpolate.cpp:14-53, interpolate_gpu.cu:9-160).
//
// three_nn: one lane per unknown point, the known set streams through LDS as float4
// (broadcast reads), a wave-uniform "can anybody still improve?" test skips the
// insertion for most candidates.  Strict '<' and ascending scan order keep the earlier
// index on equal distances, as in the reference.  The reference keeps its running
// bests in double initialised to 1e40 but compares a float d against them; float
// bests initialised to +inf give bit-identical decisions and outputs ((float)1e40 ==
// +inf, inf < inf is false exactly like inf < 1e40).
#include <algorithm>
#include <cmath>

#include <cstdlib>

#include "common.h"
#include "binning.h"
#include "bin_kernels.h"

namespace ws3d {

constexpr int NN_TILE = 1024;

// inverse-distance weights of the FP module (pointnet2_modules.py:139-142) from three_nn's SQUARED distances:
// w_k = (1 / (sqrt(d2_k) + 1e-8)) / sum_j (1 / (sqrt(d2_j) + 1e-8)), every operation a separate correctly rounded fp32 op like
// the torch composition sqrt / add / reciprocal / sum / div
__device__ __forceinline__ void nn_weights3(const float d0, const float d1, const float d2, float *__restrict__ w) {
    const float r0 = 1.0f / (sqrtf(d0) + 1e-8f), r1 = 1.0f / (sqrtf(d1) + 1e-8f), r2 = 1.0f / (sqrtf(d2) + 1e-8f);
    const float norm = (r0 + r1) + r2;
    w[0] = r0 / norm; w[1] = r1 / norm; w[2] = r2 / norm;
}
#ifndef NN_UNROLL
#define NN_UNROLL 4   // candidates per side and trip in the binned 3-NN walk
#endif

__global__ __launch_bounds__(256) void three_nn_kernel(int n, int m,
                                                       const float *__restrict__ unknown,
                                                       const float *__restrict__ known,
                                                       float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx, float *__restrict__ weight) {
    __shared__ float4 tile[NN_TILE];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int pi = blockIdx.x * 256 + tid;
    const bool active = pi < n;
    unknown += (size_t)b * n * 3;
    known += (size_t)b * m * 3;
    float ux = 0.f, uy = 0.f, uz = 0.f;
    if (active) { ux = unknown[pi * 3 + 0]; uy = unknown[pi * 3 + 1]; uz = unknown[pi * 3 + 2]; }

    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    auto insert = [&](const float d, const int k) {
        if (d < b1) {
            b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k;
        } else if (d < b2) {
            b3 = b2; i3 = i2; b2 = d; i2 = k;
        } else if (d < b3) {
            b3 = d; i3 = k;
        }
    };
    for (int base = 0; base < m; base += NN_TILE) {
        const int lim = min(NN_TILE, m - base);
        __syncthreads();
        for (int i = tid; i < lim; i += 256) {
            const float *p = known + (size_t)(base + i) * 3;
            tile[i] = make_float4(p[0], p[1], p[2], 0.f);
        }
        __syncthreads();
        int i = 0;
        for (; i + 4 <= lim; i += 4) {
            const float4 p0 = tile[i], p1 = tile[i + 1], p2 = tile[i + 2], p3 = tile[i + 3];
            const float d0 = sqdist3(ux - p0.x, uy - p0.y, uz - p0.z);
            const float d1 = sqdist3(ux - p1.x, uy - p1.y, uz - p1.z);
            const float d2 = sqdist3(ux - p2.x, uy - p2.y, uz - p2.z);
            const float d3 = sqdist3(ux - p3.x, uy - p3.y, uz - p3.z);
            if (__any((d0 < b3) | (d1 < b3) | (d2 < b3) | (d3 < b3))) {
                insert(d0, base + i);
                insert(d1, base + i + 1);
                insert(d2, base + i + 2);
                insert(d3, base + i + 3);
            }
        }
        for (; i < lim; ++i) {
            const float4 p = tile[i];
            insert(sqdist3(ux - p.x, uy - p.y, uz - p.z), base + i);
        }
    }
    if (active) {
        float *od = dist2 + ((size_t)b * n + pi) * 3;
        int32_t *oi = idx + ((size_t)b * n + pi) * 3;
        od[0] = b1; od[1] = b2; od[2] = b3;
        oi[0] = i1; oi[1] = i2; oi[2] = i3;
        if (weight) nn_weights3(b1, b2, b3, weight + ((size_t)b * n + pi) * 3);
    }
}

// ---- (x, z) grid flavour of the binned known set (binning.h), for the 3-NN search only: bin_kernels.h
__global__ __launch_bounds__(1024) void bin_points_xz_kernel(int n, const float *__restrict__ xyz, char *__restrict__ ws) {
    bin_points_xz_body(blockIdx.x, n, xyz, ws);
}

// Exact 3-NN against an x-binned copy of the known set (binning.h; built once per FP layer by
// ws3d_sort_points_x).  The reference's ascending scan with strict '<' keeps, among equal
// distances, the smaller index in front: its result is the 3 smallest (d2, index) pairs in
// lexicographic order, which is order-independent.  Each lane walks the sorted array outwards from
// its own x cell, left and right, and stops on a side once a LOWER BOUND of every remaining
// squared distance on that side exceeds its third best.  The array is ordered by cell, so a point
// further out than p has a cell index >= cell(p), hence an x within one cell width of p.x or
// beyond: dx > (p.x - ux) - slack with slack = 2 cell widths (1 from the cell order, the rest
// covers the <= 5e-4 cell fp32 error of the cell coordinate and the clamped end cells); and
// d2 = fmaf(dz,dz,fmaf(dx,dx,dy*dy)) >= fl(dx*dx) because rounding is monotone.  At equality the
// point is still examined (index tie-break); a non-finite p.x (clamped into an end cell by the
// binning) never stops the walk.
// LDS = true: the whole binned known set (m <= 4096 points, 64 KB) is staged in LDS first -- the
// walk is bound by the rate of its random 16-byte reads, and LDS serves those an order of
// magnitude faster than the global gather path (0.20 -> see DESIGN.md 5.3).
template <bool LDS>
__device__ __forceinline__ void three_nn_sorted_body(const int bx, const int b, int n, int m, const float *__restrict__ unknown,
                                                     const char *__restrict__ ws,
                                                     float *__restrict__ dist2, int32_t *__restrict__ idx, float *__restrict__ weight,
                                                     const char *__restrict__ wsq) {
    extern __shared__ __attribute__((aligned(16))) char smem_nn[];
    int pi = bx * 512 + threadIdx.x;
    const char *base = ws + (size_t)b * bin_scene_stride(m);
    const float4 *sorted = reinterpret_cast<const float4 *>(base);
    const BinHeader hdr = *reinterpret_cast<const BinHeader *>(base + (size_t)m * 16);
    const int *start = reinterpret_cast<const int *>(base + (size_t)m * 16 + sizeof(BinHeader));
    if (LDS) {
        float4 *stage = reinterpret_cast<float4 *>(smem_nn);
        for (int i = threadIdx.x; i < m; i += 512) stage[i] = sorted[i];
        sorted = stage;
        if (hdr.pad > 0) {          // grid flavour: the cell table is read several times per query
            int *tab = reinterpret_cast<int *>(stage + m);
            for (int i = threadIdx.x; i < BQS_CELLS + 4; i += 512) tab[i] = start[i];
            start = tab;
        }
        __syncthreads();
    }
    if (pi >= n) return;
    float ux, uy, uz;
    if (wsq) {
        // queries in CELL order: lane pi takes the pi-th point of a binned copy of the unknown set (any flavour: n float4
        // {x, y, z, bits(index)} per scene) and writes its row at that index -- the lanes of a wave then walk cells of similar
        // density and end their searches together (each query's result does not depend on which lane runs it)
        const float4 q = reinterpret_cast<const float4 *>(wsq + (size_t)b * bin_scene_stride(n))[pi];
        ux = q.x; uy = q.y; uz = q.z;
        pi = __float_as_int(q.w);
    } else {
        const float *u = unknown + ((size_t)b * n + pi) * 3;
        ux = u[0]; uy = u[1]; uz = u[2];
    }
    const float slack = hdr.inv_w > 0.f ? 2.0f / hdr.inv_w : INFINITY;   // two cell widths, see above

    float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int i1 = 0, i2 = 0, i3 = 0;
    auto before = [](float d, int k, float bd, int bi) { return d < bd || (d == bd && k < bi); };
    auto visit = [&](const float4 p) {
        const float d = sqdist3(ux - p.x, uy - p.y, uz - p.z);
        const int k = __float_as_int(p.w);
        if (before(d, k, b3, i3)) {
            if (before(d, k, b2, i2)) {
                b3 = b2; i3 = i2;
                if (before(d, k, b1, i1)) { b2 = b1; i2 = i1; b1 = d; i1 = k; } else { b2 = d; i2 = k; }
            } else {
                b3 = d; i3 = k;
            }
        }
    };
    if (hdr.pad > 0) {
        // ---- (x, z) grid: square rings of cells around the query's cell ----
        // After the rings 0..r every unvisited known point lies in a cell whose x or z coordinate
        // differs from the query's by more than r cells, so its distance exceeds r cell sizes (minus
        // the fp32 error of the cell coordinates, <= 1e-3 cell): stop once that bound exceeds the third
        // best.  The query's cell is clamped into the grid; for a query outside the bounding box the
        // bound only gets more conservative.  Rows of a ring are contiguous ranges of the cell table.
        const int gx = hdr.pad, gz = start[GRID_GZ];
        const float inv_wx = hdr.inv_w, inv_wz = __int_as_float(start[GRID_INV_WZ]);
        const int cx = grid_coord(ux, hdr.xmin, inv_wx, gx), cz = grid_coord(uz, __int_as_float(start[GRID_ZMIN]), inv_wz, gz);
        float hmin = INFINITY;                        // the smaller cell size of the axes that have more than one cell
        if (gx > 1) hmin = fminf(hmin, 1.0f / inv_wx);
        if (gz > 1) hmin = fminf(hmin, 1.0f / inv_wz);
        auto scan = [&](int z, int x_lo, int x_hi) {  // cells (x_lo..x_hi, z), clipped to the grid
            if (z < 0 || z >= gz) return;
            x_lo = max(x_lo, 0); x_hi = min(x_hi, gx - 1);
            if (x_lo > x_hi) return;
            int i = start[z * gx + x_lo];
            const int e = start[z * gx + x_hi + 1];
            for (; i + 1 < e; i += 2) {               // two independent loads per trip
                const float4 p0 = sorted[i], p1 = sorted[i + 1];
                visit(p0); visit(p1);
            }
            if (i < e) visit(sorted[i]);
        };
        for (int dz = -1; dz <= 1; ++dz) scan(cz + dz, cx - 1, cx + 1);          // rings 0 and 1 as one 3 x 3 block
        const int reach = max(max(cx, gx - 1 - cx), max(cz, gz - 1 - cz));       // last ring that still holds cells
        for (int r = 1; r < reach; ) {
            const float lb = (float)r * hmin * 0.999f;
            if (lb * lb > b3) break;
            ++r;
            scan(cz - r, cx - r, cx + r);
            scan(cz + r, cx - r, cx + r);
            for (int z = cz - r + 1; z <= cz + r - 1; ++z) { scan(z, cx - r, cx - r); scan(z, cx + r, cx + r); }
        }
        float *odg = dist2 + ((size_t)b * n + pi) * 3;
        int32_t *oig = idx + ((size_t)b * n + pi) * 3;
        odg[0] = b1; odg[1] = b2; odg[2] = b3;
        oig[0] = i1; oig[1] = i2; oig[2] = i3;
        if (weight) nn_weights3(b1, b2, b3, weight + ((size_t)b * n + pi) * 3);
        return;
    }
    if (hdr.pad < 0) {
        // fine-grid flavour of the ball query (ws3d_sort_points_grid) handed to three_nn: its table is not ours; the result
        // is order-independent, so a full scan of the binned copy is still exact (correct, just not pruned)
        for (int i = 0; i < m; ++i) visit(sorted[i]);
        float *odf = dist2 + ((size_t)b * n + pi) * 3;
        int32_t *oif = idx + ((size_t)b * n + pi) * 3;
        odf[0] = b1; odf[1] = b2; odf[2] = b3;
        oif[0] = i1; oif[1] = i2; oif[2] = i3;
        if (weight) nn_weights3(b1, b2, b3, weight + ((size_t)b * n + pi) * 3);
        return;
    }
    const int c0 = x_cell(ux, hdr.xmin, hdr.inv_w);
    int R = start[c0], L = R - 1;
    bool go_r = R < m, go_l = L >= 0;
    // 4 points per side and trip: the 8 loads are independent of the tests (addresses only depend
    // on R / L), so the walk pays one memory round trip per 8 candidates instead of per candidate;
    // loads past the stopping point or the array ends are clamped and their results unused
    while (go_r || go_l) {
        float4 pr[NN_UNROLL], pl[NN_UNROLL];
#pragma unroll
        for (int q = 0; q < NN_UNROLL; ++q) {
            pr[q] = sorted[min(R + q, m - 1)];
            pl[q] = sorted[max(L - q, 0)];
        }
#pragma unroll
        for (int q = 0; q < NN_UNROLL; ++q) {
            if (go_r) {
                const float lb = (pr[q].x - ux) - slack;   // every point further right has dx > lb
                if (lb > 0.f && lb * lb > b3 && pr[q].x < INFINITY) { go_r = false; } else { visit(pr[q]); go_r = ++R < m; }
            }
            if (go_l) {
                const float lb = (ux - pl[q].x) - slack;   // every point further left has -dx > lb
                if (lb > 0.f && lb * lb > b3 && pl[q].x > -INFINITY) { go_l = false; } else { visit(pl[q]); go_l = --L >= 0; }
            }
        }
    }
    float *od = dist2 + ((size_t)b * n + pi) * 3;
    int32_t *oi = idx + ((size_t)b * n + pi) * 3;
    od[0] = b1; od[1] = b2; od[2] = b3;
    oi[0] = i1; oi[1] = i2; oi[2] = i3;
    if (weight) nn_weights3(b1, b2, b3, weight + ((size_t)b * n + pi) * 3);
}

template <bool LDS>
__global__ __launch_bounds__(512) void three_nn_sorted_kernel(int n, int m, const float *__restrict__ unknown,
                                                              const char *__restrict__ ws,
                                                              float *__restrict__ dist2, int32_t *__restrict__ idx, float *__restrict__ weight,
                                                              const char *__restrict__ wsq) {
    three_nn_sorted_body<LDS>(blockIdx.x, blockIdx.y, n, m, unknown, ws, dist2, idx, weight, wsq);
}

// several searches (the FP modules of a network: every level's queries against the next level's binned centres) in ONE launch:
// job k owns the x tiles [tiles_end of job k - 1, tiles_end of job k); dynamic LDS sized for the largest known set
struct NnJob { const float *unknown; const char *ws; float *dist2; int32_t *idx; float *weight; const char *wsq; int n, m, tiles_end, pad; };
constexpr int NN_MAX_JOBS = 4;
struct NnJobs { NnJob j[NN_MAX_JOBS]; int njobs; };
__global__ __launch_bounds__(512) void three_nn_sorted_jobs_kernel(const NnJobs jobs) {
    int k = 0, t0 = 0;
    while (k + 1 < jobs.njobs && (int)blockIdx.x >= jobs.j[k].tiles_end) { t0 = jobs.j[k].tiles_end; ++k; }
    const NnJob j = jobs.j[k];
    three_nn_sorted_body<true>((int)blockIdx.x - t0, blockIdx.y, j.n, j.m, j.unknown, j.ws, j.dist2, j.idx, j.weight, j.wsq);
}

constexpr int TI_CCH = 16;

__global__ __launch_bounds__(256) void three_interpolate_kernel(int c, int m, int n,
                                                                const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight,
                                                                float *__restrict__ out) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * TI_CCH;
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= n) return;
    const int32_t *id = idx + ((size_t)b * n + pi) * 3;
    const float *w = weight + ((size_t)b * n + pi) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *p = points + ((size_t)b * c + c0) * m;
    float *o = out + ((size_t)b * c + c0) * n + pi;
    const int cc = min(TI_CCH, c - c0);
    for (int ch = 0; ch < cc; ++ch, p += m, o += n)
        // w0*p0 + w1*p1 + w2*p2 (interpolate_gpu.cu:96) under nvcc's default contraction
        *o = __builtin_fmaf(w2, p[i2], __builtin_fmaf(w0, p[i0], w1 * p[i1]));
}

// n % 4 == 0: 4 consecutive points per lane -- indices and weights arrive as three 16-byte loads
// each, every channel costs 12 gathers and ONE 16-byte store; 2 channels in flight
__global__ __launch_bounds__(256) void three_interpolate_vec4_kernel(int c, int m, int n,
                                                                     const float *__restrict__ points,
                                                                     const int32_t *__restrict__ idx,
                                                                     const float *__restrict__ weight,
                                                                     float *__restrict__ out) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * TI_CCH;
    const int pi = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (pi >= n) return;
    int id[12];
    float w[12];
    {
        const int4 *ip = reinterpret_cast<const int4 *>(idx + ((size_t)b * n + pi) * 3);
        const float4 *wp = reinterpret_cast<const float4 *>(weight + ((size_t)b * n + pi) * 3);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int4 iv = ip[q];
            const float4 wv = wp[q];
            id[4 * q] = iv.x; id[4 * q + 1] = iv.y; id[4 * q + 2] = iv.z; id[4 * q + 3] = iv.w;
            w[4 * q] = wv.x; w[4 * q + 1] = wv.y; w[4 * q + 2] = wv.z; w[4 * q + 3] = wv.w;
        }
    }
    const float *p = points + ((size_t)b * c + c0) * m;
    float *o = out + ((size_t)b * c + c0) * n + pi;
    const int cc = min(TI_CCH, c - c0);
    auto blend = [&](const float *row) {
        float4 r;  // w0*p0 + w1*p1 + w2*p2 under nvcc's default contraction, as in the scalar kernel
        r.x = __builtin_fmaf(w[2], row[id[2]], __builtin_fmaf(w[0], row[id[0]], w[1] * row[id[1]]));
        r.y = __builtin_fmaf(w[5], row[id[5]], __builtin_fmaf(w[3], row[id[3]], w[4] * row[id[4]]));
        r.z = __builtin_fmaf(w[8], row[id[8]], __builtin_fmaf(w[6], row[id[6]], w[7] * row[id[7]]));
        r.w = __builtin_fmaf(w[11], row[id[11]], __builtin_fmaf(w[9], row[id[9]], w[10] * row[id[10]]));
        return r;
    };
    int ch = 0;
    for (; ch + 2 <= cc; ch += 2) {
        const float4 r0 = blend(p + (size_t)ch * m), r1 = blend(p + (size_t)(ch + 1) * m);
        *reinterpret_cast<float4 *>(o + (size_t)ch * n) = r0;
        *reinterpret_cast<float4 *>(o + (size_t)(ch + 1) * n) = r1;
    }
    if (ch < cc) *reinterpret_cast<float4 *>(o + (size_t)ch * n) = blend(p + (size_t)ch * m);
}

// Rows-in-LDS variant: the 3 gathers per output element hit a (b, ch) row of m floats; with the row
// resident in LDS they cost an LDS read instead of a trip through the texture path (random 4-byte
// global gathers run at a fraction of the store bandwidth).  One workgroup = CH rows x all n points
// (4 consecutive points per lane and trip): rows staged once with 16-byte loads, every output
// written with 16-byte stores.  Same fmaf expression as the scalar kernel.
template <int CH, int NT = 512>
__global__ __launch_bounds__(NT) void three_interpolate_lds_kernel(int c, int m, int n,
                                                                    const float *__restrict__ points,
                                                                    const int32_t *__restrict__ idx,
                                                                    const float *__restrict__ weight,
                                                                    float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_ti[];
    float *rows = reinterpret_cast<float *>(smem_ti);  // CH * m
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * CH;
    const int cc = min(CH, c - c0);
    const float *p = points + ((size_t)b * c + c0) * m;
    const int tot = cc * m;   // the cc rows are contiguous in global memory
    if ((m & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        for (int i = threadIdx.x; i < tot / 4; i += NT) reinterpret_cast<float4 *>(rows)[i] = reinterpret_cast<const float4 *>(p)[i];
    } else {
        for (int i = threadIdx.x; i < tot; i += NT) rows[i] = p[i];
    }
    __syncthreads();
    const int per_x = (n / 4 + (int)gridDim.x - 1) / (int)gridDim.x;   // groups of 4 points per x-slice
    const int g_lo = blockIdx.x * per_x, g_hi = min(n / 4, g_lo + per_x);
    for (int g = g_lo + threadIdx.x; g < g_hi; g += NT) {
        const int pi = 4 * g;
        int id[12];
        float w[12];
        const int4 *ip = reinterpret_cast<const int4 *>(idx + ((size_t)b * n + pi) * 3);
        const float4 *wp = reinterpret_cast<const float4 *>(weight + ((size_t)b * n + pi) * 3);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int4 iv = ip[q];
            const float4 wv = wp[q];
            id[4 * q] = iv.x; id[4 * q + 1] = iv.y; id[4 * q + 2] = iv.z; id[4 * q + 3] = iv.w;
            w[4 * q] = wv.x; w[4 * q + 1] = wv.y; w[4 * q + 2] = wv.z; w[4 * q + 3] = wv.w;
        }
        float *o = out + ((size_t)b * c + c0) * n + pi;
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            if (ch < cc) {
                const float *row = rows + ch * m;
                float4 r;
                r.x = __builtin_fmaf(w[2], row[id[2]], __builtin_fmaf(w[0], row[id[0]], w[1] * row[id[1]]));
                r.y = __builtin_fmaf(w[5], row[id[5]], __builtin_fmaf(w[3], row[id[3]], w[4] * row[id[4]]));
                r.z = __builtin_fmaf(w[8], row[id[8]], __builtin_fmaf(w[6], row[id[6]], w[7] * row[id[7]]));
                r.w = __builtin_fmaf(w[11], row[id[11]], __builtin_fmaf(w[9], row[id[9]], w[10] * row[id[10]]));
                *reinterpret_cast<float4 *>(o + (size_t)ch * n) = r;
            }
        }
    }
}

__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    int c, int n, int m, const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, float *__restrict__ grad_points) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * TI_CCH;
    const int pi = blockIdx.x * 256 + threadIdx.x;
    if (pi >= n) return;
    const int32_t *id = idx + ((size_t)b * n + pi) * 3;
    const float *w = weight + ((size_t)b * n + pi) * 3;
    const int i0 = id[0], i1 = id[1], i2 = id[2];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    const float *g = grad_out + ((size_t)b * c + c0) * n + pi;
    float *gp = grad_points + ((size_t)b * c + c0) * m;
    const int cc = min(TI_CCH, c - c0);
    for (int ch = 0; ch < cc; ++ch, g += n, gp += m) {
        const float gv = *g;
        atomicAdd(gp + i0, gv * w0);
        atomicAdd(gp + i1, gv * w1);
        atomicAdd(gp + i2, gv * w2);
    }
}

__device__ __forceinline__ float relu_f(float v) { return v < 0.f ? 0.f : v; }  // NaN stays NaN (torch.relu)

// y[b, o, l] = act(y[b, o, l] + bias[o]) in place, one pass (the SharedMLP GEMM epilogue that
// rocBLAS' strided-batched GEMM does not fuse for a per-row bias): float4 along l.
__global__ __launch_bounds__(256) void bias_act_kernel(int o_ch, long l, int relu, float *__restrict__ y,
                                                       const float *__restrict__ bias) {
    const long row = blockIdx.y;                       // b * o_ch + o
    const float bv = bias[row % o_ch];
    float *p = y + row * l;
    const long l4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? (l >> 2) : 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < l4; i += (long)gridDim.x * 256) {
        float4 v = reinterpret_cast<float4 *>(p)[i];
        v.x += bv; v.y += bv; v.z += bv; v.w += bv;
        if (relu) { v.x = relu_f(v.x); v.y = relu_f(v.y); v.z = relu_f(v.z); v.w = relu_f(v.w); }
        reinterpret_cast<float4 *>(p)[i] = v;
    }
    for (long i = l4 * 4 + (long)blockIdx.x * 256 + threadIdx.x; i < l; i += (long)gridDim.x * 256) {
        float v = p[i] + bv;
        p[i] = relu ? relu_f(v) : v;
    }
}

// out[b, o, m] = act(max_s y[b, o, m, s] + bias[o]): the SA module's pool over nsample fused with
// the last layer's bias + ReLU (they commute with the max exactly: both are monotone
// non-decreasing and so is fp32 rounding).  G = s/4 adjacent lanes own one (b,o,m) row and read
// it as float4 (fully coalesced); the group maximum is folded with DPP-width shuffles.  NaN
// propagates like torch.amax / F.max_pool2d.
__device__ __forceinline__ float nanmax(float a, float b) { return (b > a || b != b) ? b : a; }

template <int G>
__global__ __launch_bounds__(256) void rowmax_bias_act_kernel(int o_ch, long m, long rows, int relu,
                                                              const float *__restrict__ y,
                                                              const float *__restrict__ bias,
                                                              float *__restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / G;
    float v = -INFINITY;
    if (row < rows) {
        const float4 q = reinterpret_cast<const float4 *>(y)[t];
        v = nanmax(nanmax(q.x, q.y), nanmax(q.z, q.w));
    }
#pragma unroll
    for (int d = 1; d < G; d <<= 1) v = nanmax(v, __shfl_xor(v, d, 64));
    if (row < rows && (t % G) == 0) {
        if (bias) v += bias[(row / m) % o_ch];
        out[row] = relu ? relu_f(v) : v;
    }
}

__global__ __launch_bounds__(256) void rowmax_bias_act_generic_kernel(int o_ch, long m, long rows, int s, int relu,
                                                                      const float *__restrict__ y,
                                                                      const float *__restrict__ bias,
                                                                      float *__restrict__ out) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float *p = y + row * s;
    float v = -INFINITY;
    for (int i = 0; i < s; ++i) v = nanmax(v, p[i]);
    if (bias) v += bias[(row / m) % o_ch];
    out[row] = relu ? relu_f(v) : v;
}

// ---- channels-last variants (features (b, points, C) row-major): the layout of the row-major
// SharedMLP GEMM chain (ws3d_amd/fastpath.py).  Lanes run along the channels, so every gather is a
// contiguous row read and every store is coalesced.

// out[b, p, 0:C] = w0*f[i0] + w1*f[i1] + w2*f[i2] (same fmaf expression as three_interpolate_kernel)
// written with row stride out_stride >= C, i.e. straight into the left part of the FP module's
// concatenated [interpolated | skip] buffer.
__global__ __launch_bounds__(256) void three_interpolate_nlc_kernel(int c, int m, int n, long total4,
                                                                    const float *__restrict__ feats,
                                                                    const int32_t *__restrict__ idx,
                                                                    const float *__restrict__ weight,
                                                                    float *__restrict__ out, int out_stride) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int c4n = c >> 2;
    const long row = t / c4n;            // b * n + p
    const int c4 = (int)(t - row * c4n);
    const int b = (int)(row / n);
    const int32_t *id = idx + row * 3;
    const float *w = weight + row * 3;
    const float4 *f = reinterpret_cast<const float4 *>(feats + (size_t)b * m * c);
    const float4 p0 = f[(size_t)id[0] * c4n + c4], p1 = f[(size_t)id[1] * c4n + c4], p2 = f[(size_t)id[2] * c4n + c4];
    const float w0 = w[0], w1 = w[1], w2 = w[2];
    float4 r;
    r.x = __builtin_fmaf(w2, p2.x, __builtin_fmaf(w0, p0.x, w1 * p1.x));
    r.y = __builtin_fmaf(w2, p2.y, __builtin_fmaf(w0, p0.y, w1 * p1.y));
    r.z = __builtin_fmaf(w2, p2.z, __builtin_fmaf(w0, p0.z, w1 * p1.z));
    r.w = __builtin_fmaf(w2, p2.w, __builtin_fmaf(w0, p0.w, w1 * p1.w));
    // 16-byte store that only needs 4-byte alignment: the row stride may be odd (257 = 256 interpolated + 1 skip channel)
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef f4v f4u __attribute__((aligned(4)));
    const f4v rv = {r.x, r.y, r.z, r.w};
    *reinterpret_cast<f4u *>(out + (size_t)row * out_stride + 4 * c4) = rv;
}

// out[r, 0:O] = max over the ns consecutive rows y[r*ns .. r*ns+ns-1, 0:O] (the SA pool on a
// channels-last tensor), written with row stride out_stride (into a slice of the MSG concat buffer).
__global__ __launch_bounds__(256) void rowmax_rows_kernel(long total4, int o_ch, int ns,
                                                          const float *__restrict__ y, float *__restrict__ out,
                                                          int out_stride) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total4) return;
    const int o4n = o_ch >> 2;
    const long r = t / o4n;
    const int o4 = (int)(t - r * o4n);
    const float4 *src = reinterpret_cast<const float4 *>(y + (size_t)r * ns * o_ch) + o4;
    float4 v = src[0];
    for (int i = 1; i < ns; ++i) {
        const float4 q = src[(size_t)i * o4n];
        v.x = nanmax(v.x, q.x); v.y = nanmax(v.y, q.y); v.z = nanmax(v.z, q.z); v.w = nanmax(v.w, q.w);
    }
    *reinterpret_cast<float4 *>(out + (size_t)r * out_stride + 4 * o4) = v;
}

// ---- pooling over nsample with the position of the maximum (training path of the SA module) ----
// The scan rule of a max-pool window, `if (v > best || isnan(v)) take v` from best = -inf at position
// 0 (the reference pools with F.max_pool2d(kernel=[1, nsample]), pointnet2_modules.py:50): first
// position of the maximum, a NaN wins and the LAST NaN is the recorded position.  Summaries of
// adjacent blocks combine with the same rule (later block wins iff its value is greater or NaN),
// so L = nsample/4 lanes scan one float4 each and merge by butterfly -- loads stay fully coalesced.
struct PoolBest { float v; int i; };
__device__ __forceinline__ PoolBest pool_take(PoolBest a, float v, int i) {
    if (v > a.v || v != v) { a.v = v; a.i = i; }
    return a;
}

template <int L>
__global__ __launch_bounds__(256) void pool_nsample_kernel(long rows, const float *__restrict__ x, float *__restrict__ out,
                                                           uint8_t *__restrict__ arg) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / L;
    const int part = (int)(t - row * L);
    const bool live = row < rows;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) q = reinterpret_cast<const float4 *>(x)[t];
    PoolBest best{-INFINITY, 4 * part};
    best = pool_take(best, q.x, 4 * part); best = pool_take(best, q.y, 4 * part + 1);
    best = pool_take(best, q.z, 4 * part + 2); best = pool_take(best, q.w, 4 * part + 3);
    // a block that never took anything reports (-inf, its first position); merged with an earlier
    // block it loses (-inf > x is false), so position 0 survives for an all -inf row like the scan
#pragma unroll
    for (int off = 1; off < L; off <<= 1) {
        PoolBest o;
        o.v = __shfl_xor(best.v, off);
        o.i = __shfl_xor(best.i, off);
        const bool mine_first = (part & off) == 0;
        const PoolBest first = mine_first ? best : o, second = mine_first ? o : best;
        best = (second.v > first.v || second.v != second.v) ? second : first;
    }
    if (live && part == 0) { out[row] = best.v; arg[row] = (uint8_t)best.i; }
}

__global__ __launch_bounds__(256) void pool_nsample_any_kernel(long rows, int ns, const float *__restrict__ x,
                                                               float *__restrict__ out, uint8_t *__restrict__ arg) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    PoolBest best{-INFINITY, 0};
    for (int i = 0; i < ns; ++i) best = pool_take(best, x[row * ns + i], i);
    out[row] = best.v; arg[row] = (uint8_t)best.i;
}

// grad_x[r, s] = (s == arg[r]) ? grad_out[r] : 0, every element written
template <int L>
__global__ __launch_bounds__(256) void pool_nsample_grad_kernel(long rows, const float *__restrict__ grad_out,
                                                                const uint8_t *__restrict__ arg, float *__restrict__ grad_x) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / L;
    if (row >= rows) return;
    const int part = (int)(t - row * L);
    const int a = (int)arg[row] - 4 * part;
    const float g = grad_out[row];
    reinterpret_cast<float4 *>(grad_x)[t] = make_float4(a == 0 ? g : 0.f, a == 1 ? g : 0.f, a == 2 ? g : 0.f, a == 3 ? g : 0.f);
}

__global__ __launch_bounds__(256) void pool_nsample_grad_any_kernel(long rows, int ns, const float *__restrict__ grad_out,
                                                                    const uint8_t *__restrict__ arg, float *__restrict__ grad_x) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const int a = arg[row];
    const float g = grad_out[row];
    for (int i = 0; i < ns; ++i) grad_x[row * ns + i] = i == a ? g : 0.f;
}

// inverse-distance weights of the FP module (pointnet2_modules.py:139-142) from three_nn's SQUARED
// distances: w_k = (1 / (sqrt(d2_k) + 1e-8)) / sum_j (1 / (sqrt(d2_j) + 1e-8)), every operation a
// separate correctly rounded fp32 op like the torch composition sqrt / add / reciprocal / sum / div.
__global__ __launch_bounds__(256) void nn_weights_kernel(long rows, const float *__restrict__ dist2, float *__restrict__ weight) {
    const long r = (long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *d = dist2 + r * 3;
    nn_weights3(d[0], d[1], d[2], weight + r * 3);
}

}  // namespace ws3d

extern "C" int ws3d_bias_act_inplace(int b, int o_ch, long l, int relu, float *y, const float *bias,
                                     ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || o_ch <= 0 || l < 0 || !y || !bias || (long)b * o_ch > 65535L * 16) {
        set_error("ws3d_bias_act_inplace: invalid argument (b=%d o=%d l=%ld)", b, o_ch, l);
        return WS3D_E_INVALID;
    }
    if (b == 0 || l == 0) return WS3D_OK;
    const long rows = (long)b * o_ch;
    if (rows > 65535) { set_error("ws3d_bias_act_inplace: too many rows"); return WS3D_E_UNSUPPORTED; }
    const unsigned gx = (unsigned)std::min<long>(64, (l / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(bias_act_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, as_stream(stream), o_ch, l, relu, y, bias);
    return check_launch("ws3d_bias_act_inplace");
}

extern "C" int ws3d_rowmax_bias_act(int b, int o_ch, long m, int s, int relu, const float *y, const float *bias,
                                    float *out, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || o_ch <= 0 || m < 0 || s <= 0 || !y || !out) {
        set_error("ws3d_rowmax_bias_act: invalid argument (b=%d o=%d m=%ld s=%d)", b, o_ch, m, s);
        return WS3D_E_INVALID;
    }
    const long rows = (long)b * o_ch * m;
    if (rows == 0) return WS3D_OK;
    hipStream_t st = as_stream(stream);
    const bool vec = (s % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
    const int g = vec ? s / 4 : 0;
    const long threads = vec ? rows * g : rows;
    const long blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffL) { set_error("ws3d_rowmax_bias_act: tensor too large"); return WS3D_E_UNSUPPORTED; }
#define WS3D_RM(G) hipLaunchKernelGGL((rowmax_bias_act_kernel<G>), dim3((unsigned)blocks), dim3(256), 0, st, o_ch, m, rows, relu, y, bias, out)
    switch (g) {
        case 1: WS3D_RM(1); break;
        case 2: WS3D_RM(2); break;
        case 4: WS3D_RM(4); break;
        case 8: WS3D_RM(8); break;
        case 16: WS3D_RM(16); break;
        case 32: WS3D_RM(32); break;
        case 64: WS3D_RM(64); break;
        default:
            hipLaunchKernelGGL(rowmax_bias_act_generic_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st,
                               o_ch, m, rows, s, relu, y, bias, out);
    }
#undef WS3D_RM
    return check_launch("ws3d_rowmax_bias_act");
}

extern "C" int ws3d_three_interpolate_nlc(int b, int c, int m, int n, const float *feats_nlc, const int32_t *idx,
                                          const float *weight, float *out_nlc, int out_stride, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || c <= 0 || m <= 0 || n < 0 || !feats_nlc || !idx || !weight || !out_nlc || out_stride < c ||
        (c & 3) || (reinterpret_cast<uintptr_t>(feats_nlc) & 15) || (reinterpret_cast<uintptr_t>(out_nlc) & 3)) {
        set_error("ws3d_three_interpolate_nlc: invalid argument (b=%d c=%d m=%d n=%d stride=%d; c %% 4, 16-byte feature base)",
                  b, c, m, n, out_stride);
        return WS3D_E_INVALID;
    }
    const long total4 = (long)b * n * (c / 4);
    if (total4 == 0) return WS3D_OK;
    if ((total4 + 255) / 256 > 0x7fffffffL) { set_error("ws3d_three_interpolate_nlc: too large"); return WS3D_E_UNSUPPORTED; }
    hipLaunchKernelGGL(three_interpolate_nlc_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream),
                       c, m, n, total4, feats_nlc, idx, weight, out_nlc, out_stride);
    return check_launch("ws3d_three_interpolate_nlc");
}

extern "C" int ws3d_rowmax_rows(long rows_out, int ns, int o_ch, const float *y, float *out, int out_stride,
                                ws3d_stream_t stream) {
    using namespace ws3d;
    const uintptr_t al = reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out);
    if (rows_out < 0 || ns <= 0 || o_ch <= 0 || !y || !out || out_stride < o_ch || (o_ch & 3) || (out_stride & 3) || (al & 15)) {
        set_error("ws3d_rowmax_rows: invalid argument (rows=%ld ns=%d o=%d stride=%d)", rows_out, ns, o_ch, out_stride);
        return WS3D_E_INVALID;
    }
    const long total4 = rows_out * (o_ch / 4);
    if (total4 == 0) return WS3D_OK;
    if ((total4 + 255) / 256 > 0x7fffffffL) { set_error("ws3d_rowmax_rows: too large"); return WS3D_E_UNSUPPORTED; }
    hipLaunchKernelGGL(rowmax_rows_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, as_stream(stream), total4,
                       o_ch, ns, y, out, out_stride);
    return check_launch("ws3d_rowmax_rows");
}

static int pool_lanes(int ns, const void *p) {   // lanes per row of the float4 kernels, 0 = generic kernel
    if ((ns & 3) || (reinterpret_cast<uintptr_t>(p) & 15)) return 0;
    const int l = ns >> 2;
    return (l == 1 || l == 2 || l == 4 || l == 8 || l == 16) ? l : 0;
}

extern "C" int ws3d_pool_nsample(long rows, int nsample, const float *x, float *out, uint8_t *arg, ws3d_stream_t stream) {
    using namespace ws3d;
    if (rows < 0 || nsample <= 0 || nsample > 255 || !x || !out || !arg) {
        set_error("ws3d_pool_nsample: invalid argument (rows=%ld nsample=%d)", rows, nsample);
        return WS3D_E_INVALID;
    }
    if (rows == 0) return WS3D_OK;
    const int l = pool_lanes(nsample, x);
    const long threads = l ? rows * l : rows;
    if ((threads + 255) / 256 > 0x7fffffffL) { set_error("ws3d_pool_nsample: too large"); return WS3D_E_UNSUPPORTED; }
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t st = as_stream(stream);
    switch (l) {
    case 1: hipLaunchKernelGGL(pool_nsample_kernel<1>, grid, block, 0, st, rows, x, out, arg); break;
    case 2: hipLaunchKernelGGL(pool_nsample_kernel<2>, grid, block, 0, st, rows, x, out, arg); break;
    case 4: hipLaunchKernelGGL(pool_nsample_kernel<4>, grid, block, 0, st, rows, x, out, arg); break;
    case 8: hipLaunchKernelGGL(pool_nsample_kernel<8>, grid, block, 0, st, rows, x, out, arg); break;
    case 16: hipLaunchKernelGGL(pool_nsample_kernel<16>, grid, block, 0, st, rows, x, out, arg); break;
    default: hipLaunchKernelGGL(pool_nsample_any_kernel, grid, block, 0, st, rows, nsample, x, out, arg);
    }
    return check_launch("ws3d_pool_nsample");
}

extern "C" int ws3d_pool_nsample_grad(long rows, int nsample, const float *grad_out, const uint8_t *arg, float *grad_x,
                                      ws3d_stream_t stream) {
    using namespace ws3d;
    if (rows < 0 || nsample <= 0 || nsample > 255 || !grad_out || !arg || !grad_x) {
        set_error("ws3d_pool_nsample_grad: invalid argument (rows=%ld nsample=%d)", rows, nsample);
        return WS3D_E_INVALID;
    }
    if (rows == 0) return WS3D_OK;
    const int l = pool_lanes(nsample, grad_x);
    const long threads = l ? rows * l : rows;
    if ((threads + 255) / 256 > 0x7fffffffL) { set_error("ws3d_pool_nsample_grad: too large"); return WS3D_E_UNSUPPORTED; }
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipStream_t st = as_stream(stream);
    switch (l) {
    case 1: hipLaunchKernelGGL(pool_nsample_grad_kernel<1>, grid, block, 0, st, rows, grad_out, arg, grad_x); break;
    case 2: hipLaunchKernelGGL(pool_nsample_grad_kernel<2>, grid, block, 0, st, rows, grad_out, arg, grad_x); break;
    case 4: hipLaunchKernelGGL(pool_nsample_grad_kernel<4>, grid, block, 0, st, rows, grad_out, arg, grad_x); break;
    case 8: hipLaunchKernelGGL(pool_nsample_grad_kernel<8>, grid, block, 0, st, rows, grad_out, arg, grad_x); break;
    case 16: hipLaunchKernelGGL(pool_nsample_grad_kernel<16>, grid, block, 0, st, rows, grad_out, arg, grad_x); break;
    default: hipLaunchKernelGGL(pool_nsample_grad_any_kernel, grid, block, 0, st, rows, nsample, grad_out, arg, grad_x);
    }
    return check_launch("ws3d_pool_nsample_grad");
}

extern "C" int ws3d_three_nn_weights(long rows, const float *dist2, float *weight, ws3d_stream_t stream) {
    using namespace ws3d;
    if (rows < 0 || !dist2 || !weight) { set_error("ws3d_three_nn_weights: invalid argument"); return WS3D_E_INVALID; }
    if (rows == 0) return WS3D_OK;
    if ((rows + 255) / 256 > 0x7fffffffL) { set_error("ws3d_three_nn_weights: too large"); return WS3D_E_UNSUPPORTED; }
    hipLaunchKernelGGL(nn_weights_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, as_stream(stream), rows, dist2, weight);
    return check_launch("ws3d_three_nn_weights");
}

extern "C" int ws3d_sort_points_xz(int b, int n, const float *xyz, void *sorted, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || n > SORT_MAX_N || !xyz || !sorted) {
        set_error("ws3d_sort_points_xz: invalid argument (b=%d n=%d, n <= %d)", b, n, SORT_MAX_N);
        return WS3D_E_INVALID;
    }
    if (b == 0) return WS3D_OK;
    hipLaunchKernelGGL(bin_points_xz_kernel, dim3(b), dim3(1024), 0, as_stream(stream), n, xyz, reinterpret_cast<char *>(sorted));
    return check_launch("ws3d_sort_points_xz");
}

static int three_nn_launch(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, float *weight,
                           const void *sorted_known, ws3d_stream_t stream, const void *sorted_unknown = nullptr);

extern "C" int ws3d_three_nn(int b, int n, int m, const float *unknown, const float *known,
                             float *dist2, int32_t *idx, const void *sorted_known, ws3d_stream_t stream) {
    return three_nn_launch(b, n, m, unknown, known, dist2, idx, nullptr, sorted_known, stream);
}

extern "C" int ws3d_three_nn_w(int b, int n, int m, const float *unknown, const float *known,
                               float *dist2, int32_t *idx, float *weight, const void *sorted_known, ws3d_stream_t stream) {
    if (!weight) { ws3d::set_error("ws3d_three_nn_w: weight is NULL"); return WS3D_E_INVALID; }
    return three_nn_launch(b, n, m, unknown, known, dist2, idx, weight, sorted_known, stream);
}

extern "C" int ws3d_three_nn_wq(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, float *weight,
                                const void *sorted_known, const void *sorted_unknown, ws3d_stream_t stream) {
    if (!weight) { ws3d::set_error("ws3d_three_nn_wq: weight is NULL"); return WS3D_E_INVALID; }
    if (sorted_unknown && (n <= 0 || n > ws3d::SORT_MAX_N)) {
        ws3d::set_error("ws3d_three_nn_wq: sorted_unknown needs 0 < n <= %d (n=%d)", ws3d::SORT_MAX_N, n);
        return WS3D_E_INVALID;
    }
    return three_nn_launch(b, n, m, unknown, known, dist2, idx, weight, sorted_known, stream, sorted_unknown);
}

static int three_nn_launch(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx, float *weight,
                           const void *sorted_known, ws3d_stream_t stream, const void *sorted_unknown) {
    using namespace ws3d;
    if (b < 0 || n < 0 || m < 0 || !unknown || (!known && m > 0) || !dist2 || !idx) {
        set_error("ws3d_three_nn: invalid argument (b=%d n=%d m=%d)", b, n, m);
        return WS3D_E_INVALID;
    }
    if (b == 0 || n == 0) return WS3D_OK;
    if (sorted_known && m >= 3 && m <= SORT_MAX_N) {
        const size_t lds = (size_t)m * sizeof(float4) + (size_t)(BQS_CELLS + 4) * sizeof(int);
        if ((size_t)m * sizeof(float4) <= 64 * 1024) {
            if (int rc = raise_lds_cap((const void *)three_nn_sorted_kernel<true>, 80 * 1024, "ws3d_three_nn")) return rc;
            hipLaunchKernelGGL(three_nn_sorted_kernel<true>, dim3((n + 511) / 512, b), dim3(512), lds, as_stream(stream),
                               n, m, unknown, reinterpret_cast<const char *>(sorted_known), dist2, idx, weight,
                               reinterpret_cast<const char *>(sorted_unknown));
        }
        else
            hipLaunchKernelGGL(three_nn_sorted_kernel<false>, dim3((n + 511) / 512, b), dim3(512), 0, as_stream(stream),
                               n, m, unknown, reinterpret_cast<const char *>(sorted_known), dist2, idx, weight,
                               reinterpret_cast<const char *>(sorted_unknown));
        return check_launch("ws3d_three_nn(sorted)");
    }
    hipLaunchKernelGGL(three_nn_kernel, dim3((n + 255) / 256, b), dim3(256), 0, as_stream(stream), n, m,
                       unknown, known, dist2, idx, weight);
    return check_launch("ws3d_three_nn");
}

extern "C" int ws3d_three_nn_jobs(int b, int njobs, const int *n, const int *m, const float *const *unknown, const void *const *sorted_known,
                                  float *const *dist2, int32_t *const *idx, float *const *weight, const void *const *sorted_unknown,
                                  ws3d_stream_t stream) {
    using namespace ws3d;
    if (b == 0 || njobs == 0) return WS3D_OK;
    if (b < 0 || b > 65535 || njobs < 0 || njobs > NN_MAX_JOBS || !n || !m || !unknown || !sorted_known || !dist2 || !idx || !weight) {
        set_error("ws3d_three_nn_jobs: invalid argument (b=%d njobs=%d, at most %d jobs)", b, njobs, NN_MAX_JOBS);
        return WS3D_E_INVALID;
    }
    NnJobs jobs{};
    size_t lds = 0;
    int tiles = 0;
    for (int k = 0; k < njobs; ++k) {
        if (n[k] <= 0 || m[k] < 3 || (size_t)m[k] * sizeof(float4) > 64 * 1024 || !unknown[k] || !sorted_known[k] || !dist2[k] || !idx[k] || !weight[k] ||
            (sorted_unknown && sorted_unknown[k] && n[k] > SORT_MAX_N)) {
            set_error("ws3d_three_nn_jobs: job %d invalid (n=%d m=%d; 3 <= m <= 4096 binned known points, a binned query copy needs n <= %d)", k, n[k], m[k],
                      SORT_MAX_N);
            return WS3D_E_INVALID;
        }
        tiles += (n[k] + 511) / 512;
        jobs.j[k] = NnJob{unknown[k], reinterpret_cast<const char *>(sorted_known[k]), dist2[k], idx[k], weight[k],
                          reinterpret_cast<const char *>(sorted_unknown ? sorted_unknown[k] : nullptr), n[k], m[k], tiles, 0};
        lds = std::max(lds, (size_t)m[k] * sizeof(float4) + (size_t)(BQS_CELLS + 4) * sizeof(int));
    }
    jobs.njobs = njobs;
    if (int rc = raise_lds_cap((const void *)three_nn_sorted_jobs_kernel, 80 * 1024, "ws3d_three_nn_jobs")) return rc;
    hipLaunchKernelGGL(three_nn_sorted_jobs_kernel, dim3((unsigned)tiles, b), dim3(512), lds, as_stream(stream), jobs);
    return check_launch("ws3d_three_nn_jobs");
}

extern "C" int ws3d_three_interpolate(int b, int c, int m, int n, const float *points,
                                      const int32_t *idx, const float *weight, float *out,
                                      ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || c < 0 || m <= 0 || n < 0 || !points || !idx || !weight || !out) {
        set_error("ws3d_three_interpolate: invalid argument (b=%d c=%d m=%d n=%d)", b, c, m, n);
        return WS3D_E_INVALID;
    }
    if (b == 0 || c == 0 || n == 0) return WS3D_OK;
    const uintptr_t al = reinterpret_cast<uintptr_t>(idx) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(out);
    static const bool wide_ok = getenv("WS3D_TI_NO_WIDE") == nullptr;
    if (wide_ok && (n & 3) == 0 && (al & 15) == 0 && c >= 16 && (size_t)m * 8 * 4 <= 128 * 1024 && n >= 4096) {
        // round 6: 8 rows per workgroup of 16 waves -- the 6 x 16 bytes of indices and weights a lane reads per trip are shared by 8
        // channels instead of 4 (they are re-read by every channel group: at 4 channels that stream is 1.5 x the output), the rows are
        // staged once per scene and group unless the launch would leave CUs empty
        constexpr int CH = 8;
        const int rows_wg = (c + CH - 1) / CH;
        int gx = 1;
        while ((long)gx * rows_wg * b < 256 && (n / 4) / (gx * 2) >= 1024) gx *= 2;
        const size_t lds = (size_t)CH * m * sizeof(float);
        if (int rc = raise_lds_cap((const void *)three_interpolate_lds_kernel<CH, 1024>, lds, "ws3d_three_interpolate")) return rc;
        hipLaunchKernelGGL((three_interpolate_lds_kernel<CH, 1024>), dim3(gx, rows_wg, b), dim3(1024), lds, as_stream(stream), c, m, n, points, idx, weight, out);
    } else if ((n & 3) == 0 && (al & 15) == 0 && (size_t)m * 4 * 4 <= 64 * 1024 && n >= 1024) {
        // rows in LDS: 4 channels per workgroup; split n over x only as far as needed to fill the chip
        constexpr int CH = 4;
        const int rows_wg = (c + CH - 1) / CH;
        int gx = 1;
        while ((long)gx * rows_wg * b < 1024 && (n / 4) / (gx * 2) >= 512) gx *= 2;
        hipLaunchKernelGGL((three_interpolate_lds_kernel<CH>), dim3(gx, rows_wg, b), dim3(512),
                           (size_t)CH * m * sizeof(float), as_stream(stream), c, m, n, points, idx, weight, out);
    } else if ((n & 3) == 0 && (al & 15) == 0)
        hipLaunchKernelGGL(three_interpolate_vec4_kernel, dim3((n / 4 + 255) / 256, (c + TI_CCH - 1) / TI_CCH, b),
                           dim3(256), 0, as_stream(stream), c, m, n, points, idx, weight, out);
    else
        hipLaunchKernelGGL(three_interpolate_kernel, dim3((n + 255) / 256, (c + TI_CCH - 1) / TI_CCH, b),
                           dim3(256), 0, as_stream(stream), c, m, n, points, idx, weight, out);
    return check_launch("ws3d_three_interpolate");
}

extern "C" int ws3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out,
                                           const int32_t *idx, const float *weight,
                                           float *grad_points, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || c < 0 || m <= 0 || n < 0 || !grad_out || !idx || !weight || !grad_points) {
        set_error("ws3d_three_interpolate_grad: invalid argument (b=%d c=%d n=%d m=%d)", b, c, n, m);
        return WS3D_E_INVALID;
    }
    if (b == 0 || c == 0 || n == 0) return WS3D_OK;
    hipLaunchKernelGGL(three_interpolate_grad_kernel, dim3((n + 255) / 256, (c + TI_CCH - 1) / TI_CCH, b),
                       dim3(256), 0, as_stream(stream), c, n, m, grad_out, idx, weight, grad_points);
    return check_launch("ws3d_three_interpolate_grad");
}
