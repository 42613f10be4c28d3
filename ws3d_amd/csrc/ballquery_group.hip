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
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "common.h"
#include "binning.h"
#include "bin_kernels.h"

namespace ws3d {

// Which flavour a binned buffer holds is written in its device-side header; the HOST picks the kernel, so the sort entry
// points also note it per buffer address (the last sort into an address wins; unknown addresses run the kernel that
// reads the header and handles both flavours).  Consulted at launch / graph-capture time, never synchronises.
static std::mutex g_flavour_mu;
static std::unordered_map<const void *, int> g_flavour;   // 1 = fine (x, z) grid, 0 = x slabs
static void note_flavour(const void *p, int f) {
    std::lock_guard<std::mutex> lk(g_flavour_mu);
    if (g_flavour.size() > 8192) g_flavour.clear();
    g_flavour[p] = f;
}
static bool grid_flavour(const void *p) {
    std::lock_guard<std::mutex> lk(g_flavour_mu);
    const auto it = g_flavour.find(p);
    return it != g_flavour.end() && it->second == 1;
}

// 16-byte store; streaming (non-temporal) when the grouped tensor is far larger than the last-level
// cache (stage-2 shapes: 3.4 GB per launch, 0.98 -> 0.75 ms), plain otherwise so that the SharedMLP
// GEMM that follows still finds a small tensor in L2 / MALL
__device__ __forceinline__ void st4(float *p, const float4 v, const bool stream) {
#ifdef WS3D_BQ_NO_NT      // A/B build (scripts/r06/bq_variants.sh): plain stores everywhere
    *reinterpret_cast<float4 *>(p) = v;
    return;
#endif
    if (stream) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        f4v t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));
    } else {
        *reinterpret_cast<float4 *>(p) = v;
    }
}
constexpr size_t STREAM_STORE_BYTES = (size_t)512 << 20;

// Shared epilogue: NC centres' padded neighbour rows (LDS) -> idx tensor and/or the fused
// (B, 3+C, M, ns) grouped tensor, coalesced along (m, s).
template <typename IDX, bool FUSED, int NT, int NC>
__device__ __forceinline__ void bq_emit(int b, int tid, int m0, int n, int m, int c_feat, int nsample,
                                        int use_xyz, const float *__restrict__ xyz /* scene base */,
                                        const float *__restrict__ features, int32_t *__restrict__ idx_out,
                                        float *__restrict__ out, const IDX *rows, int rstride,
                                        const int *cnt_s, const float4 *cen, int nbatch = -1, float4 *pad4 = nullptr /* NC float4 of LDS or NULL */,
                                        const bool pad_ready = false /* pad4 holds every centre's centred first hit already: no barrier in here */) {
    if (nbatch < 0) nbatch = (int)gridDim.y;      // scenes of the launch (the 1-D grid kernel passes it)
    const int total_e = NC * nsample;
    // round 6: entry e -> (centre, sample) by a shift where nsample is a power of two (every list length of the network; a run-time
    // integer division is ~20 VALU instructions per entry), and the lists leave the chip four entries per lane as 16-byte stores
    const int ns_sh = (nsample & (nsample - 1)) == 0 ? (int)__builtin_ctz((unsigned)nsample) : -1;
    auto centre_of = [&](const int e) { return ns_sh >= 0 ? e >> ns_sh : e / nsample; };
    const bool idx_quads = (nsample & 3) == 0 && (reinterpret_cast<uintptr_t>(idx_out) & 15) == 0;
    if (!FUSED) {
        // ball_query contract: rows without any hit are left untouched (ball_query_gpu.cu:29-44); use_xyz bit 2 (the _fill entry
        // point): such rows are written as zeros -- what the reference's callers get from their zero-initialised idx tensor
        int32_t *o = idx_out + ((size_t)b * m + m0) * nsample;
        const bool fill = (use_xyz & 4) != 0;
        if (idx_quads) {
            for (int q = tid; q < total_e / 4; q += NT) {
                const int e = 4 * q, c = centre_of(e);
                if (m0 + c >= m) continue;
                const IDX *r = rows + (size_t)c * rstride + (e - c * nsample);
                if (cnt_s[c] > 0) *reinterpret_cast<int4 *>(o + e) = make_int4((int)r[0], (int)r[1], (int)r[2], (int)r[3]);
                else if (fill) *reinterpret_cast<int4 *>(o + e) = make_int4(0, 0, 0, 0);
            }
            return;
        }
        for (int e = tid; e < total_e; e += NT) {
            const int c = centre_of(e), s = e - c * nsample;
            if (m0 + c < m) {
                if (cnt_s[c] > 0) o[e] = (int32_t)rows[(size_t)c * rstride + s];
                else if (fill) o[e] = 0;
            }
        }
        return;
    }
    // (the 3 + 1 channel shape below writes the lists beside the grouped rows: it holds the four indices of a store in registers anyway)
    const bool lists_in_fast_path = !(use_xyz & 2) && c_feat == 1 && (use_xyz & 1) && gridDim.z == 1 && idx_quads &&
                                    (reinterpret_cast<uintptr_t>(out) & 15) == 0;
    if (idx_out && blockIdx.z == 0 && !lists_in_fast_path) {
        int32_t *o = idx_out + ((size_t)b * m + m0) * nsample;
        if (idx_quads) {
            for (int q = tid; q < total_e / 4; q += NT) {
                const int e = 4 * q, c = centre_of(e);
                if (m0 + c >= m) continue;
                const IDX *r = rows + (size_t)c * rstride + (e - c * nsample);
                *reinterpret_cast<int4 *>(o + e) = make_int4((int)r[0], (int)r[1], (int)r[2], (int)r[3]);
            }
        } else {
            for (int e = tid; e < total_e; e += NT) {
                const int c = centre_of(e), s = e - c * nsample;
                if (m0 + c < m) o[e] = (int32_t)rows[(size_t)c * rstride + s];
            }
        }
    }
    if (use_xyz & 2) {
        // channels-LAST output (b, m, nsample, 3 + C) from channels-last features (b, n, C): one
        // contiguous row per (centre, sample) -- the layout the row-major SharedMLP GEMMs consume
        // without a transpose (ws3d_query_and_group_nlc).  32 lanes move one feature row with
        // aligned 16-byte loads and 4-byte-aligned 16-byte stores; lanes 0-2 add the centred xyz.
        const int cx3 = (use_xyz & 1) ? 3 : 0;
        const int row = cx3 + c_feat;
        float *ob = out + ((size_t)b * m + m0) * nsample * row;
        const float *fb = features ? features + (size_t)b * n * c_feat : nullptr;
        // gridDim.z workgroups share a tile of centres (each repeats the cheap search) and emit
        // disjoint slices of its (centre, sample) rows: wide rows at few centres still fill the chip
        const int e_chunk = ((total_e + (int)gridDim.z - 1) / (int)gridDim.z + 1) & ~1;
        const int e_lo = (int)blockIdx.z * e_chunk, e_hi = min(total_e, e_lo + e_chunk);
        if (c_feat == 1 && cx3 == 3) {
            for (int e = e_lo + tid; e < e_hi; e += NT) {
                const int c = e / nsample, s = e - c * nsample;
                if (m0 + c >= m) continue;
                const int id = (int)rows[(size_t)c * rstride + s];
                const float4 ce = cen[c];
                const float *p = xyz + (size_t)id * 3;
                *reinterpret_cast<float4 *>(ob + (size_t)e * 4) = make_float4(p[0] - ce.x, p[1] - ce.y, p[2] - ce.z, fb[id]);
            }
        } else if ((c_feat & 3) == 0 && c_feat > 0 && (reinterpret_cast<uintptr_t>(fb) & 15) == 0) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            typedef f4v f4u __attribute__((aligned(4)));
            const int l32 = tid & 31, f4 = c_feat >> 2;
            for (int e0 = e_lo + (tid >> 5) * 2; e0 < e_hi; e0 += (NT >> 5) * 2) {
                int id[2];
                bool ok[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = min(e0 + q, total_e - 1);
                    const int c = e / nsample;
                    ok[q] = e0 + q < e_hi && m0 + c < m;
                    id[q] = ok[q] ? (int)rows[(size_t)c * rstride + (e - c * nsample)] : 0;   // rows of absent centres are uninitialised
                }
                for (int c4 = l32; c4 < f4; c4 += 32) {
                    f4v v[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) v[q] = reinterpret_cast<const f4v *>(fb + (size_t)id[q] * c_feat)[c4];
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (ok[q]) *reinterpret_cast<f4u *>(ob + (size_t)(e0 + q) * row + cx3 + 4 * c4) = v[q];
                }
                if (cx3 && l32 < 3) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (ok[q]) {
                            const int c = (e0 + q) / nsample;
                            const float cv = l32 == 0 ? cen[c].x : (l32 == 1 ? cen[c].y : cen[c].z);
                            ob[(size_t)(e0 + q) * row + l32] = xyz[(size_t)id[q] * 3 + l32] - cv;
                        }
                }
            }
        } else {
            for (int e = e_lo + tid; e < e_hi; e += NT) {
                const int c = e / nsample, s = e - c * nsample;
                if (m0 + c >= m) continue;
                const int id = (int)rows[(size_t)c * rstride + s];
                float *o = ob + (size_t)e * row;
                if (cx3) {
                    const float4 ce = cen[c];
                    const float *p = xyz + (size_t)id * 3;
                    o[0] = p[0] - ce.x; o[1] = p[1] - ce.y; o[2] = p[2] - ce.z;
                }
                for (int ch = 0; ch < c_feat; ++ch) o[cx3 + ch] = fb[(size_t)id * c_feat + ch];
            }
        }
        return;
    }
    const int c_xyz = use_xyz ? 3 : 0;
    const int c_out = c_xyz + c_feat;
    const size_t plane = (size_t)m * nsample;
    float *ob = out + (size_t)b * c_out * plane + (size_t)m0 * nsample;
    const float *fb = features ? features + (size_t)b * c_feat * n : nullptr;
    // gridDim.z workgroups share a tile of centres: each repeats the (cheap) search and emits
    // its own slice of the feature channels, so wide layers (C = 256..512) fill the chip
    const int chunk = (c_feat + (int)gridDim.z - 1) / (int)gridDim.z;
    const int ch_lo = (int)blockIdx.z * chunk, ch_hi = min(c_feat, ch_lo + chunk);
    if ((nsample & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
        const bool stream = (size_t)nbatch * c_out * plane * sizeof(float) > STREAM_STORE_BYTES;
        // 4 consecutive samples of one centre per lane: 16-byte stores along (m, s), 4 independent
        // gathers per channel, two channels in flight
        if (c_feat == 1 && use_xyz && gridDim.z == 1) {
            // the c2 / first-SA-level shape (3 xyz + 1 feature channel): two groups of 4 samples per trip, all 16 gathers
            // issued before the first store -- the emit is latency-bound on its gathers.  Round 6: the eight list entries of a trip
            // are read from LDS unconditionally (a lane without work reads row 0: its `ok ? r[u] : 0` was eight exec-masked branches,
            // each with its own LDS round trip before the gather could issue), the gathers and stores take 32-bit offsets from scalar
            // bases, the streaming / plain store choice is made once per launch instead of once per store, and the lists go out here
            // (ws3d_query_and_group writes them too) as one 16-byte store per lane and group.
            typedef float f3v __attribute__((ext_vector_type(3)));
            typedef f3v f3u __attribute__((aligned(4)));
            const bool lists = idx_out && lists_in_fast_path;
            int32_t *ib = lists ? idx_out + ((size_t)b * m + m0) * nsample : nullptr;
            const char *xb = reinterpret_cast<const char *>(xyz), *fbb = reinterpret_cast<const char *>(fb);
            char *o0 = reinterpret_cast<char *>(ob), *o1 = reinterpret_cast<char *>(ob + plane), *o2 = reinterpret_cast<char *>(ob + 2 * plane),
                 *o3 = reinterpret_cast<char *>(ob + 3 * plane);
            // The PADDING of a list (entries >= the centre's hit count repeat its first hit: 50-62 of the 64 entries of an r = 0.1 m
            // ball on a KITTI-density scan) needs no gather of its own: the first hit of every centre is fetched ONCE, centred, parked
            // in LDS (pad4), and a group of four entries that lies wholly in the padding stores that value four times.  Measured
            // (scripts/r06/bq_variants.sh): the store pattern alone runs at 6.8 TB/s, the emit with a 12-byte and a 4-byte gather
            // per entry at half of that (0.83 ms with the search compiled out), with the gathers of the real entries only 0.64, without
            // any gather 0.52 (profiles/r06_bq_emit_anatomy.txt).
            const bool skip_pad = pad4 != nullptr;
            if (skip_pad && !pad_ready) {
                if (tid < NC) {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (m0 + tid < m) {
                        const unsigned id0 = (unsigned)rows[(size_t)tid * rstride];          // (a centre without a hit: its row is all zeros -> point 0, as below)
                        const f3v p0 = *reinterpret_cast<const f3u *>(xb + id0 * 12u);
                        const float4 ce = cen[tid];
                        v = make_float4(p0.x - ce.x, p0.y - ce.y, p0.z - ce.z, *reinterpret_cast<const float *>(fbb + id0 * 4u));
                    }
                    pad4[tid] = v;
                }
                __syncthreads();
            }
            auto trips = [&](auto stream_tag) {
                constexpr bool ST = decltype(stream_tag)::value;
                const int nq = total_e / 4;
                for (int q0 = tid; q0 < nq; q0 += 2 * NT) {
                    unsigned eo[2];
                    int cc[2], id[2][4];
                    bool ok[2], pad[2];
                    f3v p[2][4];
                    float f[2][4];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int q = q0 + h * NT;
                        const int e = 4 * min(q, nq - 1);
                        cc[h] = centre_of(e);
                        ok[h] = q < nq && m0 + cc[h] < m;
                        eo[h] = (unsigned)e * 4u;
                        const int s0 = e - cc[h] * nsample;
#ifdef WS3D_BQG_ALL_PAD        // ablation: no gather at all (every group of four entries takes the parked first hit; values are wrong)
                        pad[h] = skip_pad;
#else
                        pad[h] = skip_pad && s0 >= cnt_s[cc[h]];
#endif
                        const IDX *r = rows + (ok[h] ? cc[h] * rstride + s0 : 0);     // (rows of absent centres are uninitialised: row 0 instead)
#ifdef WS3D_BQG_NO_IDS         // ablation: the list entries are not read (values are wrong)
#pragma unroll
                        for (int u = 0; u < 4; ++u) id[h][u] = e + u;
#else
#pragma unroll
                        for (int u = 0; u < 4; ++u) id[h][u] = (int)r[u];
#endif
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int u = 0; u < 4; ++u) { p[h][u] = f3v{0.f, 0.f, 0.f}; f[h][u] = 0.f; }
                        if (!pad[h]) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
#ifdef WS3D_BQG_PACKED_ABL    // ablation (scripts/ablate_bq.sh): what ONE aligned 16-byte gather per sample would cost (values are wrong)
                                const float4 v = reinterpret_cast<const float4 *>(xyz)[id[h][u] * 3 / 4];
                                p[h][u] = f3v{v.x, v.y, v.z};
                                f[h][u] = v.w;
#else
                                p[h][u] = *reinterpret_cast<const f3u *>(xb + (unsigned)id[h][u] * 12u);
                                f[h][u] = *reinterpret_cast<const float *>(fbb + (unsigned)id[h][u] * 4u);
#endif
                            }
                        }
                    }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (!ok[h]) continue;
                        if (lists) *reinterpret_cast<int4 *>(reinterpret_cast<char *>(ib) + eo[h]) = make_int4(id[h][0], id[h][1], id[h][2], id[h][3]);
                        float4 vx, vy, vz, vf;
                        if (pad[h]) {
                            const float4 pv = pad4[cc[h]];
                            vx = make_float4(pv.x, pv.x, pv.x, pv.x); vy = make_float4(pv.y, pv.y, pv.y, pv.y);
                            vz = make_float4(pv.z, pv.z, pv.z, pv.z); vf = make_float4(pv.w, pv.w, pv.w, pv.w);
                        } else {
                            const float4 ce = cen[cc[h]];
                            vx = make_float4(p[h][0].x - ce.x, p[h][1].x - ce.x, p[h][2].x - ce.x, p[h][3].x - ce.x);
                            vy = make_float4(p[h][0].y - ce.y, p[h][1].y - ce.y, p[h][2].y - ce.y, p[h][3].y - ce.y);
                            vz = make_float4(p[h][0].z - ce.z, p[h][1].z - ce.z, p[h][2].z - ce.z, p[h][3].z - ce.z);
                            vf = make_float4(f[h][0], f[h][1], f[h][2], f[h][3]);
                        }
                        st4(reinterpret_cast<float *>(o0 + eo[h]), vx, ST);
                        st4(reinterpret_cast<float *>(o1 + eo[h]), vy, ST);
                        st4(reinterpret_cast<float *>(o2 + eo[h]), vz, ST);
                        st4(reinterpret_cast<float *>(o3 + eo[h]), vf, ST);
                    }
                }
            };
            if (stream) trips(std::true_type{}); else trips(std::false_type{});
            return;
        }
        for (int q = tid; q < total_e / 4; q += NT) {
            const int e = 4 * q;
            const int c = centre_of(e), s = e - c * nsample;
            if (m0 + c >= m) continue;
            const IDX *r = rows + (size_t)c * rstride + s;
            const int i0 = (int)r[0], i1 = (int)r[1], i2 = (int)r[2], i3 = (int)r[3];
            if (use_xyz && blockIdx.z == 0) {
                const float4 ce = cen[c];
                // ONE 12-byte gather per neighbour (global_load_dwordx3, 4-byte aligned) instead of three dword gathers: the
                // emit is bound by the rate of uncoalesced L2 accesses (~1 lane per clock and CU), not by bytes
                typedef float f3v __attribute__((ext_vector_type(3)));
                typedef f3v f3u __attribute__((aligned(4)));
                const f3v p0 = *reinterpret_cast<const f3u *>(xyz + (size_t)i0 * 3), p1 = *reinterpret_cast<const f3u *>(xyz + (size_t)i1 * 3),
                          p2 = *reinterpret_cast<const f3u *>(xyz + (size_t)i2 * 3), p3 = *reinterpret_cast<const f3u *>(xyz + (size_t)i3 * 3);
                // grouped_xyz -= new_xyz (pointnet2_utils.py:252)
                *reinterpret_cast<float4 *>(ob + e) = make_float4(p0.x - ce.x, p1.x - ce.x, p2.x - ce.x, p3.x - ce.x);
                *reinterpret_cast<float4 *>(ob + plane + e) = make_float4(p0.y - ce.y, p1.y - ce.y, p2.y - ce.y, p3.y - ce.y);
                *reinterpret_cast<float4 *>(ob + 2 * plane + e) = make_float4(p0.z - ce.z, p1.z - ce.z, p2.z - ce.z, p3.z - ce.z);
            }
            int ch = ch_lo;
            for (; ch + 4 <= ch_hi; ch += 4) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float *f = fb + (size_t)(ch + u) * n;
                    v[u] = make_float4(f[i0], f[i1], f[i2], f[i3]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) st4(ob + (size_t)(c_xyz + ch + u) * plane + e, v[u], stream);
            }
            for (; ch < ch_hi; ++ch) {
                const float *f0 = fb + (size_t)ch * n;
                st4(ob + (size_t)(c_xyz + ch) * plane + e, make_float4(f0[i0], f0[i1], f0[i2], f0[i3]), stream);
            }
        }
        return;
    }
    for (int e = tid; e < total_e; e += NT) {
        const int c = e / nsample, s = e - c * nsample;
        if (m0 + c >= m) continue;
        const int id = (int)rows[(size_t)c * rstride + s];
        if (use_xyz && blockIdx.z == 0) {
            const float4 ce = cen[c];
            const float *p = xyz + (size_t)id * 3;
            ob[e] = p[0] - ce.x;                 // grouped_xyz -= new_xyz (pointnet2_utils.py:252)
            ob[plane + e] = p[1] - ce.y;
            ob[2 * plane + e] = p[2] - ce.z;
        }
        for (int ch = ch_lo; ch < ch_hi; ++ch) ob[(size_t)(c_xyz + ch) * plane + e] = fb[(size_t)ch * n + id];
    }
}

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

    bq_emit<IDX, FUSED, 64 * BQ_NW, 64>(b, tid, m0, n, m, c_feat, nsample, use_xyz, xyz, features, idx_out, out,
                                         rows, rstride, cnt_s, cen);
}

// ----------------------------------------------------------------------------------------
// x-binned ball query.  A per-scene copy of the points counting-sorted into BQS_CELLS uniform
// x cells (float4 {x,y,z,index} + a cell-start table) turns the O(N) scan per centre into a scan
// of the cells overlapping |x - cx| < r only (r=0.1 in an 80 m scene: ~1/400 of the points).
// The slab arrives in arbitrary order, so each lane keeps the nsample SMALLEST indices seen
// (insertion into its sorted LDS row) -- exactly the reference's "first nsample by index" set;
// distance test, padding and no-hit rules are unchanged, so the result is bit-identical to the
// brute-force scan (and independent of the order inside a cell).  Slabs longer than
// BQS_MAX_SLAB (pathological density) fall back to the ordered full scan for that lane.
constexpr int BQS_MAX_SLAB = 3072;
#ifndef BQS_UNROLL
#define BQS_UNROLL 8
#endif
// one workgroup per scene: min/max of x, LDS histogram, exclusive scan, scatter
__global__ __launch_bounds__(1024) void bin_points_x_kernel(int n, const float *__restrict__ xyz,
                                                            char *__restrict__ ws) {
    __shared__ int hist[BQS_CELLS];
    __shared__ int wsum[16];
    __shared__ float red_min[16], red_max[16];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    xyz += (size_t)b * n * 3;
    char *base = ws + (size_t)b * bin_scene_stride(n);
    float4 *sorted = reinterpret_cast<float4 *>(base);
    BinHeader *hdr = reinterpret_cast<BinHeader *>(base + (size_t)n * 16);
    int *start = reinterpret_cast<int *>(base + (size_t)n * 16 + sizeof(BinHeader));

    float mn = INFINITY, mx = -INFINITY;
    for (int i = tid; i < n; i += 1024) {
        const float x = xyz[(size_t)i * 3];
        if (fabsf(x) < INFINITY) { mn = fminf(mn, x); mx = fmaxf(mx, x); }  // finite only
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    if (lane == 0) { red_min[w] = mn; red_max[w] = mx; }
    for (int i = tid; i < BQS_CELLS; i += 1024) hist[i] = 0;
    __syncthreads();
    mn = red_min[0]; mx = red_max[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) { mn = fminf(mn, red_min[i]); mx = fmaxf(mx, red_max[i]); }
    const float xmin = mn <= mx ? mn : 0.f;
    const float width = mn <= mx ? (mx - mn) : 0.f;
    const float inv_w = width > 0.f ? (float)BQS_CELLS / width : 0.f;
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[x_cell(xyz[(size_t)i * 3], xmin, inv_w)], 1);
    __syncthreads();
    // exclusive scan of 2048 counters: 2 per thread, wave scan + cross-wave offsets
    const int a0 = hist[2 * tid], a1 = hist[2 * tid + 1];
    int v = a0 + a1;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    int off = 0;
    for (int i = 0; i < w; ++i) off += wsum[i];
    const int excl = off + v - (a0 + a1);
    __syncthreads();
    hist[2 * tid] = excl;            // becomes the running scatter cursor
    hist[2 * tid + 1] = excl + a0;
    start[2 * tid] = excl;
    start[2 * tid + 1] = excl + a0;
    if (tid == 0) { start[BQS_CELLS] = n; hdr->xmin = xmin; hdr->inv_w = inv_w; hdr->n = n; hdr->pad = 0; }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const float *p = xyz + (size_t)i * 3;
        const int pos = atomicAdd(&hist[x_cell(p[0], xmin, inv_w)], 1);
        sorted[pos] = make_float4(p[0], p[1], p[2], __int_as_float(i));
    }
}

// ---- fine (x, z) grid flavour (binning.h) for the ball query: bin_kernels.h
__global__ __launch_bounds__(1024) void bin_points_grid_kernel(int n, const float *__restrict__ xyz, char *__restrict__ ws) {
    bin_points_grid_body(blockIdx.x, n, xyz, ws);
}

// several (cloud, flavour) binning jobs in one launch: grid (scenes, jobs)
__global__ __launch_bounds__(1024) void bin_points_jobs_kernel(const BinJobs jobs) {
    const BinJob j = jobs.j[blockIdx.y];
    if (j.kind == 0) bin_points_grid_body(blockIdx.x, j.n, j.xyz, j.ws);
    else bin_points_xz_body(blockIdx.x, j.n, j.xyz, j.ws);
}

// One workgroup = 64 centres (one per lane) x 4 waves; wave j scans the j-th quarter of every
// centre's slab and keeps its own "nsample smallest indices" row; wave 0 then 4-way merges.
template <bool FUSED>
__global__ __launch_bounds__(256) void ball_query_sorted_kernel(int n, int m, int c_feat, float radius,
                                                                int nsample, int use_xyz,
                                                                const float *__restrict__ xyz,
                                                                const char *__restrict__ ws,
                                                                const float *__restrict__ new_xyz,
                                                                const float *__restrict__ features,
                                                                int32_t *__restrict__ idx_out,
                                                                float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *cen = reinterpret_cast<float4 *>(smem);                    // 64
    int *cnt_s = reinterpret_cast<int *>(cen + 64);                    // 4 * 64 (partial counts), then [0,64) = final
    uint16_t *rows = reinterpret_cast<uint16_t *>(cnt_s + 256);        // final rows: 64 * rstride
    const int rstride = nsample + 1;
    uint16_t *part = rows + 64 * rstride;                              // partial rows: 4 * 64 * rstride

    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.x * 64;
    const int mi = m0 + lane;
    const bool active = mi < m;
    xyz += (size_t)b * n * 3;
    const char *base = ws + (size_t)b * bin_scene_stride(n);
    const float4 *sorted = reinterpret_cast<const float4 *>(base);
    const BinHeader hdr = *reinterpret_cast<const BinHeader *>(base + (size_t)n * 16);
    const int *start = reinterpret_cast<const int *>(base + (size_t)n * 16 + sizeof(BinHeader));
    new_xyz += (size_t)b * m * 3;
    float cx = 0.f, cy = 0.f, cz = 0.f;
    if (active) { cx = new_xyz[mi * 3 + 0]; cy = new_xyz[mi * 3 + 1]; cz = new_xyz[mi * 3 + 2]; }
    if (w == 0) cen[lane] = make_float4(cx, cy, cz, 0.f);
    __syncthreads();

    const float radius2 = radius * radius;
    const float rabs = fabsf(radius);
    uint16_t *row = part + (size_t)(w * 64 + lane) * rstride;
    int cnt = 0;
    auto insert = [&](const int id) {  // keep the nsample smallest indices, ascending
        if (cnt == nsample) {
            if (id >= (int)row[nsample - 1]) return;
            --cnt;
        }
        int pos = cnt;
        while (pos > 0 && (int)row[pos - 1] > id) { row[pos] = row[pos - 1]; --pos; }
        row[pos] = (uint16_t)id;
        ++cnt;
    };
#ifdef WS3D_BQS_NO_SEARCH
    if (active && w == 0) { for (int q = 0; q < 20; ++q) insert((mi * 7 + q * 13) % n); }
    if (false) {
#else
    if (active && cx == cx) {
#endif
        auto scan = [&](int k, const int ke) {
            // BQS_UNROLL entries per trip: the loads are independent of the tests, so the walk pays one memory round trip
            // per BQS_UNROLL candidates (clamped loads past the end are ignored)
            for (; k < ke; k += BQS_UNROLL) {
                float4 p[BQS_UNROLL];
#pragma unroll
                for (int u = 0; u < BQS_UNROLL; ++u) p[u] = sorted[min(k + u, ke - 1)];
#pragma unroll
                for (int u = 0; u < BQS_UNROLL; ++u) {
                    const float dx = cx - p[u].x;
                    if (k + u < ke && fabsf(dx) < rabs) {
                        const float d2 = sqdist3(dx, cy - p[u].y, cz - p[u].z);
                        if (d2 < radius2) insert(__float_as_int(p[u].w));
                    }
                }
            }
        };
        bool full_scan = false;
        if (hdr.pad < 0) {
            // fine (x, z) grid: a hit has |fl(cx - px)| < r and |fl(cz - pz)| < r (each squared term alone is <= d2 under
            // monotone rounding), i.e. px within r (1 + 2^-23) of cx; the bounds are widened by 6e-7 (|c| + r) -- ten times
            // the rounding of c -+ r -- and grid_coord is monotone, so no cell holding a hit is skipped.  Wave j takes
            // the grid rows j, j + 4, ...: one contiguous range of cells per row.
            const uint16_t *start16 = reinterpret_cast<const uint16_t *>(start);
            const int *params = reinterpret_cast<const int *>(reinterpret_cast<const char *>(start) + GRID16_PARAMS);
            const int gx = -hdr.pad, gz = params[2];
            const float zmin = __int_as_float(params[0]), inv_wz = __int_as_float(params[1]);
            const float sx = (fabsf(cx) + rabs) * 6e-7f, sz = (fabsf(cz) + rabs) * 6e-7f;
            const int ix0 = grid_coord((cx - rabs) - sx, hdr.xmin, hdr.inv_w, gx), ix1 = grid_coord((cx + rabs) + sx, hdr.xmin, hdr.inv_w, gx);
            const int iz0 = grid_coord((cz - rabs) - sz, zmin, inv_wz, gz), iz1 = grid_coord((cz + rabs) + sz, zmin, inv_wz, gz);
            int total = 0;
            for (int iz = iz0; iz <= iz1 && total <= BQS_MAX_SLAB; ++iz)
                total += (int)start16[iz * gx + ix1 + 1] - (int)start16[iz * gx + ix0];
            if (total <= BQS_MAX_SLAB) {
                for (int iz = iz0 + w; iz <= iz1; iz += 4) scan((int)start16[iz * gx + ix0], (int)start16[iz * gx + ix1 + 1]);
            } else {
                full_scan = true;
            }
        } else {
            // x slabs: cells overlapping |x - cx| < r: x_cell is monotone, one extra cell each side absorbs the
            // rounding of cx -+ r (cell width >> 1 ulp of x)
            const int c_lo = max(0, x_cell(cx - rabs, hdr.xmin, hdr.inv_w) - 1);
            const int c_hi = min(BQS_CELLS - 1, x_cell(cx + rabs, hdr.xmin, hdr.inv_w) + 1);
            const int k0 = start[c_lo], kend = start[c_hi + 1];
            if (kend - k0 <= BQS_MAX_SLAB) {
                const int q = (kend - k0 + 3) >> 2;               // this wave's quarter of the slab
                const int k = min(kend, k0 + w * q);
                scan(k, min(kend, k + q));
            } else {
                full_scan = true;
            }
        }
        if (full_scan && w == 0) {
            // pathological density: ordered full scan with early exit (the reference's own loop)
            for (int q = 0; q < n && cnt < nsample; ++q) {
                const float d2 = sqdist3(cx - xyz[q * 3 + 0], cy - xyz[q * 3 + 1], cz - xyz[q * 3 + 2]);
                if (d2 < radius2) { row[cnt] = (uint16_t)q; ++cnt; }
            }
        }
    }
    cnt_s[w * 64 + lane] = cnt;
    __syncthreads();
    if (w == 0) {
        // 4-way merge of the ascending partial rows: the nsample smallest indices overall
        const uint16_t *r0 = part + (size_t)(0 * 64 + lane) * rstride, *r1 = part + (size_t)(1 * 64 + lane) * rstride;
        const uint16_t *r2 = part + (size_t)(2 * 64 + lane) * rstride, *r3 = part + (size_t)(3 * 64 + lane) * rstride;
        const int c0 = cnt_s[lane], c1 = cnt_s[64 + lane], c2 = cnt_s[128 + lane], c3 = cnt_s[192 + lane];
        int i0 = 0, i1 = 0, i2 = 0, i3 = 0, total = 0;
        uint16_t *dst = rows + (size_t)lane * rstride;
        while (total < nsample) {
            const int v0 = i0 < c0 ? (int)r0[i0] : 0x7fffffff, v1 = i1 < c1 ? (int)r1[i1] : 0x7fffffff;
            const int v2 = i2 < c2 ? (int)r2[i2] : 0x7fffffff, v3 = i3 < c3 ? (int)r3[i3] : 0x7fffffff;
            const int mn = min(min(v0, v1), min(v2, v3));
            if (mn == 0x7fffffff) break;
            dst[total++] = (uint16_t)mn;
            i0 += (v0 == mn); i1 += (v1 == mn); i2 += (v2 == mn); i3 += (v3 == mn);
        }
        const uint16_t first = total > 0 ? dst[0] : (uint16_t)0;
        for (int s = total; s < nsample; ++s) dst[s] = first;
    }
    __syncthreads();
    if (w == 0) {
        // final count AFTER every lane has read the partial counts (cnt_s[0..63] is reused)
        const int total = min(nsample, cnt_s[lane] + cnt_s[64 + lane] + cnt_s[128 + lane] + cnt_s[192 + lane]);
        cnt_s[lane] = active ? total : 0;
    }
    __syncthreads();
#ifndef WS3D_BQS_NO_EMIT
    bq_emit<uint16_t, FUSED, 256, 64>(b, tid, m0, n, m, c_feat, nsample, use_xyz, xyz, features, idx_out, out, rows,
                                      rstride, cnt_s, cen);
#else
    if (tid == 0 && out) out[((size_t)b * m + m0)] = (float)rows[0] + (float)cnt_s[0];
#endif
}

// ---- fine-grid flavour: 64 centres per workgroup, wave 0 searches (one lane per centre: the grid leaves ~5-30 candidates,
// so the four-wave split and its merge are not worth their 33 KB of partial rows), then all four waves emit.  10 KB of LDS
// per workgroup instead of 44 KB: 8 workgroups = 32 waves per CU instead of 12 -- the emit is latency-bound on its gathers,
// and resident waves are what keeps loads in flight.  1-D grid, XCD-aware: workgroup g runs on XCD g % 8 (observed dispatch
// order), so scene = (g / 8 / tiles) * 8 + g % 8 puts all tiles of a scene on ONE XCD -- its points and features are pulled
// into one L2 instead of eight.
// Anatomy at the c2 shape (scripts/ablate_bq.sh, 512 scenes): search alone 0.16 ms, emit alone 0.61 ms (4.4 TB/s of stores), together
// 0.78 ms.  A variant in which a workgroup takes 4 tiles and wave 0 searches tile t + 1 while waves 1-3 emit tile t was built and
// measured: 0.90 ms -- the emit is bound by the issue capacity of its waves, not by load latency, so three emitting waves
// instead of four cost more (emit alone 0.87 ms) than the hidden search saves; not kept.
template <bool FUSED>
__global__ __launch_bounds__(256) void ball_query_grid_kernel(int nb, int n, int m, int c_feat, float radius, int nsample, int use_xyz,
                                                              const float *__restrict__ xyz, const char *__restrict__ ws,
                                                              const float *__restrict__ new_xyz, const float *__restrict__ features,
                                                              int32_t *__restrict__ idx_out, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *cen = reinterpret_cast<float4 *>(smem);                    // 64
    int *cnt_s = reinterpret_cast<int *>(cen + 64);                    // 64
    uint16_t *rows = reinterpret_cast<uint16_t *>(cnt_s + 64);         // 64 * rstride
    const int rstride = nsample + 1;
    const int tiles = (m + 63) / 64;
    int b, tile;
    if ((nb & 7) == 0) {
        const int g = blockIdx.x, j = g >> 3;
        b = (j / tiles) * 8 + (g & 7);
        tile = j - (j / tiles) * tiles;
    } else {
        b = blockIdx.x / tiles;
        tile = blockIdx.x - b * tiles;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = tile * 64;
    const int mi = m0 + lane;
    const bool active = mi < m;
    xyz += (size_t)b * n * 3;
    const char *base = ws + (size_t)b * bin_scene_stride(n);
    const float4 *sorted = reinterpret_cast<const float4 *>(base);
    const BinHeader hdr = *reinterpret_cast<const BinHeader *>(base + (size_t)n * 16);
    const uint16_t *start16 = reinterpret_cast<const uint16_t *>(base + (size_t)n * 16 + sizeof(BinHeader));
    const int *params = reinterpret_cast<const int *>(base + (size_t)n * 16 + sizeof(BinHeader) + GRID16_PARAMS);
    new_xyz += (size_t)b * m * 3;
    if (w == 0) {
        float cx = 0.f, cy = 0.f, cz = 0.f;
        if (active) { cx = new_xyz[mi * 3 + 0]; cy = new_xyz[mi * 3 + 1]; cz = new_xyz[mi * 3 + 2]; }
        cen[lane] = make_float4(cx, cy, cz, 0.f);
        const float radius2 = radius * radius;
        const float rabs = fabsf(radius);
        uint16_t *row = rows + (size_t)lane * rstride;
        int cnt = 0;
        auto insert = [&](const int id) {  // keep the nsample smallest indices, ascending
            if (cnt == nsample) {
                if (id >= (int)row[nsample - 1]) return;
                --cnt;
            }
            int pos = cnt;
            while (pos > 0 && (int)row[pos - 1] > id) { row[pos] = row[pos - 1]; --pos; }
            row[pos] = (uint16_t)id;
            ++cnt;
        };
#ifdef WS3D_BQG_NO_SEARCH   // ablation (scripts/ablate_bq.sh): every centre "finds" 40 neighbours
        if (active) { for (int q = 0; q < 40 && q < nsample; ++q) row[q] = (uint16_t)((mi * 7 + q * 13) % n); cnt = min(40, nsample); }
        if (false) {
#else
        if (active && cx == cx && hdr.pad >= 0) {
            // not a fine-grid buffer (see ball_query_grid_coop_kernel): the reference's ordered scan
            for (int q = 0; q < n && cnt < nsample; ++q) {
                const float d2 = sqdist3(cx - xyz[q * 3 + 0], cy - xyz[q * 3 + 1], cz - xyz[q * 3 + 2]);
                if (d2 < radius2) { row[cnt] = (uint16_t)q; ++cnt; }
            }
        } else if (active && cx == cx) {
#endif
            // see ball_query_sorted_kernel for the bounds argument (hits lie within r (1 + 2^-23) of the centre on each axis)
            const int gx = -hdr.pad, gz = params[2];
            const float zmin = __int_as_float(params[0]), inv_wz = __int_as_float(params[1]);
            const float sx = (fabsf(cx) + rabs) * 6e-7f, sz = (fabsf(cz) + rabs) * 6e-7f;
            const int ix0 = grid_coord((cx - rabs) - sx, hdr.xmin, hdr.inv_w, gx), ix1 = grid_coord((cx + rabs) + sx, hdr.xmin, hdr.inv_w, gx);
            const int iz0 = grid_coord((cz - rabs) - sz, zmin, inv_wz, gz), iz1 = grid_coord((cz + rabs) + sz, zmin, inv_wz, gz);
            int total = 0;
            for (int iz = iz0; iz <= iz1 && total <= BQS_MAX_SLAB; ++iz)
                total += (int)start16[iz * gx + ix1 + 1] - (int)start16[iz * gx + ix0];
            if (total <= BQS_MAX_SLAB) {
                for (int iz = iz0; iz <= iz1; ++iz) {
                    int k = (int)start16[iz * gx + ix0];
                    const int ke = (int)start16[iz * gx + ix1 + 1];
                    for (; k < ke; k += 4) {
                        float4 p[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) p[u] = sorted[min(k + u, ke - 1)];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float dx = cx - p[u].x;
                            if (k + u < ke && fabsf(dx) < rabs) {
                                const float d2 = sqdist3(dx, cy - p[u].y, cz - p[u].z);
                                if (d2 < radius2) insert(__float_as_int(p[u].w));
                            }
                        }
                    }
                }
            } else {
                // pathological density: ordered full scan with early exit (the reference's own loop)
                for (int q = 0; q < n && cnt < nsample; ++q) {
                    const float d2 = sqdist3(cx - xyz[q * 3 + 0], cy - xyz[q * 3 + 1], cz - xyz[q * 3 + 2]);
                    if (d2 < radius2) { row[cnt] = (uint16_t)q; ++cnt; }
                }
            }
        }
        const uint16_t first = cnt > 0 ? row[0] : (uint16_t)0;
        for (int s2 = cnt; s2 < nsample; ++s2) row[s2] = first;
        cnt_s[lane] = active ? cnt : 0;
    }
    __syncthreads();
#ifdef WS3D_BQG_NO_EMIT      // ablation: one word per workgroup keeps the search alive
    if (tid == 0 && idx_out) idx_out[((size_t)b * m + m0) * nsample] = (int)rows[0] + cnt_s[0];
#else
    bq_emit<uint16_t, FUSED, 256, 64>(b, tid, m0, n, m, c_feat, nsample, use_xyz, xyz, features, idx_out, out, rows, rstride, cnt_s, cen, nb);
#endif
}

// ---- fine-grid flavour, one WAVE per centre (round 3).  The kernel above gives every centre one lane: fine while a list holds a
// handful of candidates (the sparse `lidar` generator: 1-3 hits), but on a scan with KITTI's density (synth.hdl64_cloud) the
// r = 0.5 m ball of a near-range centre holds 100-300 candidates and ~80 hits, the lane walks them one by one, keeps its 32 smallest
// indices by insertion into an LDS row (a chain of dependent LDS round trips), and the wave waits for its slowest lane while
// three of the workgroup's four waves idle: 736 us for level 1 of a batch of 8 (55 us on the sparse clouds).
// Here a wave takes 16 of the workgroup's 64 centres in turn and works on ONE centre with all 64 lanes:
//   1. ranges   lane (centre i, row q) reads the cell-start table for grid row iz0 + q of centre i: the first four rows of all
//               16 centres in one round trip (more rows -- r / cell > 3 -- are fetched per centre, four at a time);
//   2. test     the rows' candidate ranges are laid end to end and dealt out 64 at a time: one distance test per lane, the hits
//               are appended to an LDS list through ballot + mbcnt (order irrelevant).  The FIRST 64 candidates of all 16 centres
//               are loaded up front (16 loads in flight) and tested before the per-centre loop starts: a centre in a sparse
//               neighbourhood (<= 64 candidates, most centres of an FPS-sampled scan) costs no memory round trip of its own;
//   3. select   more than nsample hits: the nsample SMALLEST indices are the contract ("first nsample by index").  Their
//               threshold T -- the nsample-th smallest -- is found bit by bit (log2 n rounds of "how many hits lie below T | bit":
//               a compare, a ballot and an s_bcnt1 per 64 hits), then the hits <= T are compacted in place;
//   4. order    <= nsample survivors: every lane ranks its own by counting the smaller ones (broadcast LDS reads) and stores it
//               at its rank; the row is padded with its first entry as before.
// No lane ever waits for another lane's longer list, the work per centre is O(candidates / 64 + log n * hits / 64 + nsample).
// More than BQC_HCAP hits or BQC_MAX_CAND candidates (pathological density): the wave scans the scene in INDEX order, 64 points
// per step, and stops at nsample hits -- the reference's own loop (ball_query_gpu.cu:29-44), 64 wide.
// Same candidate bounds, same distance expression and operand order as above: bit-identical lists.
#ifndef BQC_WIDE_BELOW
#define BQC_WIDE_BELOW 2048   // tiles of 64 centres below which a launch uses 16 waves x 4 centres per tile (scripts/ubench/bq_wide_threshold.sh)
#endif
#ifndef BQC_LARGE_NW
#define BQC_LARGE_NW 4      // waves per tile of the launches that fill the chip by themselves
#endif
#ifndef BQC_HCAP
#define BQC_HCAP 1024                 // hits kept per centre (uint16 in LDS, per wave)
#endif
#ifndef BQC_FUSED_WPE
#define BQC_FUSED_WPE 1               // waves per SIMD the fused 4-wave kernel is compiled for (A/B: scripts/r06/bq_variants2.sh)
#endif
constexpr int BQC_MAX_CAND = 8192;    // candidates tested per centre before the ordered scan takes over

// NW waves per workgroup share a tile of 64 centres, CPW = 64 / NW each: 4 x 16 when the launch fills the chip by itself (the c2 block: 32768
// workgroups), 16 x 4 when it does not (a batch of 8 at level 1: 512 tiles -- sixteen centres in turn per wave left 2 waves per SIMD and a
// kernel as long as one wave's sixteen searches).
// blockIdx.y == 1 (ws3d_ball_query_pairs2, round 5): the SECOND scale of a set-abstraction level -- same points, centres and binned copy,
// its own radius / nsample / outputs -- so that both searches of a level are ONE launch (a launch costs the 20-deep pipeline ~2.3 us).
// The search of ONE tile of 64 centres by the NW waves of a workgroup (steps 1-4 of the comment above): fills rows[c][0 .. nsample) with the
// padded lists, cnt_s[c] with the hit counts (0 for a centre beyond m) and cen[c] with the centres.  `base` = the scene's binned copy,
// xyz / new_xyz = the scene's points / centres.  (A function of its own since round 6: a persistent search / emit pipeline was built on it and not kept, profiles/r06_bq_emit_anatomy.txt.)
template <int NW>
__device__ __forceinline__ void bqc_search_tile(const int lane, const int w, const int m0, const int n, const int m, const float radius, const int nsample,
                                                const float *__restrict__ xyz, const char *__restrict__ base, const float *__restrict__ new_xyz,
                                                float4 *cen, int *cnt_s, uint16_t *rows, const int rstride, uint16_t *hits_all, uint16_t *stage_all) {
    constexpr int CPW = 64 / NW;
    // grid rows fetched per centre and batch.  A wave of <= 8 centres has the lanes for EIGHT, and the r >= 0.5 m searches of the network span
    // 5-9 rows of the fine grid (with four per batch their centres take a second table fetch in the dense path).  Measured (round 6,
    // -DBQC_ROWS_PER_BATCH=8, scripts/r06/bq_c3_ab.sh): the eight searches of an hdl64 batch 236 -> 229 us in all (level 2, r = 1.0: 32.6 -> 27.7;
    // level 1, r = 0.5: 69.7 -> 69.1), of a lidar batch 163 -> 175 (the longer range cascade on every candidate): four stays the default.
#ifdef BQC_ROWS_PER_BATCH
    constexpr int R = BQC_ROWS_PER_BATCH;
#else
    constexpr int R = 4;
#endif
    static_assert(R * CPW <= 64 && (R == 4 || R == 8), "one lane per (centre, row) of a batch");
#ifdef WS3D_BQC_NO_SEARCH2   // ablation (scripts/r06/bq_variants2.sh): NOTHING is searched -- every centre "finds" six neighbours
    for (int ci = 0; ci < CPW; ++ci) {
        const int c = CPW * w + ci;
        uint16_t *row = rows + (size_t)c * rstride;
        const uint16_t first = (uint16_t)(((m0 + c) * 7) % n);
        for (int s2 = lane; s2 < nsample; s2 += 64) row[s2] = s2 < 6 ? (uint16_t)(((m0 + c) * 7 + s2 * 13) % n) : first;
        if (lane == 0) {
            cnt_s[c] = (m0 + c < m) ? 6 : 0;
            cen[c] = (m0 + c < m) ? make_float4(new_xyz[(m0 + c) * 3], new_xyz[(m0 + c) * 3 + 1], new_xyz[(m0 + c) * 3 + 2], 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    return;
#endif
    const float4 *sorted = reinterpret_cast<const float4 *>(base);
    const BinHeader hdr = *reinterpret_cast<const BinHeader *>(base + (size_t)n * 16);
    const uint16_t *start16 = reinterpret_cast<const uint16_t *>(base + (size_t)n * 16 + sizeof(BinHeader));
    const int *params = reinterpret_cast<const int *>(base + (size_t)n * 16 + sizeof(BinHeader) + GRID16_PARAMS);
    uint16_t *hits = hits_all + w * BQC_HCAP;
    uint16_t *stage = stage_all + w * (CPW * 64);
    const float radius2 = radius * radius;
    const float rabs = fabsf(radius);
    // the host picks this kernel from what the sort entry points noted per buffer ADDRESS; a buffer of another flavour that
    // landed on a noted address (a copy, a recycled allocation) is recognised by its header and served by the ordered scan:
    // slow, never wrong
    const bool not_grid = hdr.pad >= 0;
    const int gx = not_grid ? 1 : -hdr.pad, gz = not_grid ? 1 : params[2];
    const float zmin = __int_as_float(params[0]), inv_wz = __int_as_float(params[1]);
    const int id_bits = 32 - __builtin_clz(max(n - 1, 1));            // ids < n

    // ---- 1. this lane's (centre, row) of the wave's CPW centres x first R grid rows (lanes R CPW .. 63 idle)
    const int ci_l = lane / R, q_l = lane % R;
    const int mi_l = ci_l < CPW ? m0 + CPW * w + ci_l : m;
    float lcx = 0.f, lcy = 0.f, lcz = 0.f;
    if (mi_l < m) { lcx = new_xyz[mi_l * 3 + 0]; lcy = new_xyz[mi_l * 3 + 1]; lcz = new_xyz[mi_l * 3 + 2]; }
    if (q_l == 0 && ci_l < CPW) cen[CPW * w + ci_l] = make_float4(lcx, lcy, lcz, 0.f);
    int l_ix0 = 0, l_ix1 = 0, l_iz0 = 0, l_nrows = 0, l_k0 = 0, l_ke = 0;
    if (mi_l < m && lcx == lcx) {
        // see ball_query_sorted_kernel for the bounds argument (hits lie within r (1 + 2^-23) of the centre on each axis)
        const float sx = (fabsf(lcx) + rabs) * 6e-7f, sz = (fabsf(lcz) + rabs) * 6e-7f;
        l_ix0 = grid_coord((lcx - rabs) - sx, hdr.xmin, hdr.inv_w, gx);
        l_ix1 = grid_coord((lcx + rabs) + sx, hdr.xmin, hdr.inv_w, gx);
        l_iz0 = grid_coord((lcz - rabs) - sz, zmin, inv_wz, gz);
        l_nrows = grid_coord((lcz + rabs) + sz, zmin, inv_wz, gz) - l_iz0 + 1;
        if (not_grid) {
            l_nrows = 1;
        } else if (q_l < l_nrows) {
            l_k0 = (int)start16[(l_iz0 + q_l) * gx + l_ix0];
            l_ke = (int)start16[(l_iz0 + q_l) * gx + l_ix1 + 1];
        }
    }
    // the R ranges of a centre laid end to end (pe[q] = end of range q in that sequence): position j -> index into `sorted`
    auto locate = [&](const int j, const int (&k0)[R], const int (&pe)[R]) {
        int r = k0[0] + j;
#pragma unroll
        for (int q = 1; q < R; ++q) r = j >= pe[q - 1] ? k0[q] + (j - pe[q - 1]) : r;
        return r;
    };
    // ---- 2a. the first 64 candidates of all CPW centres: CPW independent loads in flight, then the tests; the hits of centre i go
    //          to stage[i][..] (this is the whole search of a centre in a sparse neighbourhood -- no memory round trip per centre)
    int l_h0 = 0;                                                       // lane i (< 16): hits among centre i's first 64 candidates
    {
        float4 pre[CPW];
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            int k0[R], pe[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                k0[q] = __builtin_amdgcn_readlane(l_k0, R * ci + q);
                pe[q] = (q ? pe[q - 1] : 0) + (__builtin_amdgcn_readlane(l_ke, R * ci + q) - k0[q]);
            }
            const int L = pe[R - 1];
            pre[ci] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lane < L) pre[ci] = sorted[locate(lane, k0, pe)];
        }
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            int L4 = 0;
#pragma unroll
            for (int q = 0; q < R; ++q) L4 += __builtin_amdgcn_readlane(l_ke, R * ci + q) - __builtin_amdgcn_readlane(l_k0, R * ci + q);
            const float cx = readlane_f(lcx, R * ci), cy = readlane_f(lcy, R * ci), cz = readlane_f(lcz, R * ci);
            const float dx = cx - pre[ci].x;
            const bool hit = lane < L4 && fabsf(dx) < rabs && sqdist3(dx, cy - pre[ci].y, cz - pre[ci].z) < radius2;
            const uint64_t mask = __ballot(hit);
            if (hit) stage[ci * 64 + mbcnt(mask)] = (uint16_t)__float_as_int(pre[ci].w);
            if (lane == ci) l_h0 = __popcll(mask);
        }
    }
    // ---- 2c. (round 6) a centre whose whole search was step 2a -- at most 64 candidates in at most four grid rows: most centres of a
    // furthest-point-sampled scan -- is finished here TOGETHER with the wave's other such centres, four lanes per centre: rank of every hit among its
    // centre's hits, the row stored in rank order, padded, counted.  One centre at a time (the loop below) each of these steps is an LDS round
    // trip the wave waits for, CPW times over: the search of the c2 block ran at the speed of that chain (0.44 ms with 24 waves per CU, 1.0 ms
    // with 8: time ~ 1 / occupancy).  Same hits, same order: bit-identical lists.
    // (16 waves x 4 centres, the launches of a batch of 8: four centres per wave leave the step 16 of 64 lanes, and on the dense r = 0.5 m
    // searches of level 1 few centres qualify -- measured 66 -> 73 us there, so those launches keep the loop alone)
    uint64_t simple_bits = 0;
#ifndef WS3D_BQC_NO_BATCH
    if constexpr (CPW >= 16 && R == 4) {
        const int len_l = l_ke - l_k0;                                  // (0 beyond the centre's rows and for lanes without a centre)
        int L_l = len_l + __shfl_xor(len_l, 1);
        L_l += __shfl_xor(L_l, 2);
        const int H_l = __shfl(l_h0, ci_l & (CPW - 1));
        const bool simple_l = ci_l < CPW && !not_grid && l_nrows <= 4 && L_l <= 64;
        simple_bits = __ballot(simple_l && q_l == 0);
        if (simple_bits) {
            const int c = CPW * w + (ci_l & (CPW - 1));
            const uint16_t *hl = stage_all + w * (CPW * 64) + (ci_l & (CPW - 1)) * 64;
            uint16_t *row = rows + (size_t)c * rstride;
            const int Hs = simple_l ? H_l : 0;
            int Hmax = Hs;
            for (int o = 32; o > 0; o >>= 1) Hmax = max(Hmax, __shfl_xor(Hmax, o));
            Hmax = __builtin_amdgcn_readfirstlane(Hmax);
            for (int j0 = 0; j0 < Hmax; j0 += 4) {
                const int j = j0 + q_l;
                const bool act = j < Hs;
                const int v = act ? (int)hl[j] : 0x7fffffff;
                int rank = 0;
                for (int i = 0; i < Hmax; i += 4) {                     // (a centre's stage row holds 64 entries: reads past its count are masked, never out of bounds)
                    const int o0 = (int)hl[i], o1 = (int)hl[i + 1], o2 = (int)hl[i + 2], o3 = (int)hl[i + 3];
                    rank += (int)(i < Hs && o0 < v) + (int)(i + 1 < Hs && o1 < v) + (int)(i + 2 < Hs && o2 < v) + (int)(i + 3 < Hs && o3 < v);
                }
                if (act && rank < nsample) row[rank] = (uint16_t)v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (simple_l) {
                const int cnt = min(Hs, nsample);
                const uint16_t first = cnt > 0 ? row[0] : (uint16_t)0;
                for (int s2 = cnt + q_l; s2 < nsample; s2 += 4) row[s2] = first;
                if (q_l == 0) cnt_s[c] = (m0 + c < m) ? cnt : 0;
            }
        }
    }
#endif
    for (int ci = 0; ci < CPW; ++ci) {
        if ((simple_bits >> (R * ci)) & 1) continue;
        const int c = CPW * w + ci;                                    // centre slot of the workgroup
        uint16_t *row = rows + (size_t)c * rstride;
        const int src = R * ci;
#ifdef WS3D_BQC_NO_SEARCH   // ablation (scripts/r06/bq_variants.sh): every centre "finds" six neighbours, nothing is searched
        const int nrows = 0;
        int cnt = 0;
        if (m0 + c < m) { if (lane < 6) row[lane] = (uint16_t)(((m0 + c) * 7 + lane * 13) % n); cnt = 6; }
#else
        const int nrows = __builtin_amdgcn_readlane(l_nrows, src);
        int cnt = 0;
#endif
        if (nrows > 0) {
            int k0[R], pe[R];
#pragma unroll
            for (int q = 0; q < R; ++q) {
                k0[q] = __builtin_amdgcn_readlane(l_k0, src + q);
                pe[q] = (q ? pe[q - 1] : 0) + (__builtin_amdgcn_readlane(l_ke, src + q) - k0[q]);
            }
            int L = pe[R - 1];
            int H = __builtin_amdgcn_readlane(l_h0, ci);
            const uint16_t *hl = stage + ci * 64;                       // this centre's hit list
            bool ordered = not_grid;
            if (L > 64 || nrows > R || not_grid) {
                // ---- 2b. a dense neighbourhood: the rest of the candidates, 64 per step, appended to the wave's long list
                const float cx = readlane_f(lcx, src), cy = readlane_f(lcy, src), cz = readlane_f(lcz, src);
                const int ix0 = __builtin_amdgcn_readlane(l_ix0, src), ix1 = __builtin_amdgcn_readlane(l_ix1, src);
                const int iz0 = __builtin_amdgcn_readlane(l_iz0, src);
                if (lane < H) hits[lane] = stage[ci * 64 + lane];
                hl = hits;
                int tested = 0;
                for (int rb = 0; rb < nrows && !ordered; rb += R) {
                    if (rb > 0) {
                        int a = 0, e = 0;
                        if (lane < R && rb + lane < nrows) {
                            a = (int)start16[(iz0 + rb + lane) * gx + ix0];
                            e = (int)start16[(iz0 + rb + lane) * gx + ix1 + 1];
                        }
#pragma unroll
                        for (int q = 0; q < R; ++q) {
                            k0[q] = __builtin_amdgcn_readlane(a, q);
                            pe[q] = (q ? pe[q - 1] : 0) + (__builtin_amdgcn_readlane(e, q) - k0[q]);
                        }
                        L = pe[R - 1];
                    }
                    tested += L;
                    if (tested > BQC_MAX_CAND) { ordered = true; break; }
                    for (int j0 = rb == 0 ? 64 : 0; j0 < L; j0 += 64) {
                        const int j = j0 + lane;
                        const float4 p = sorted[j < L ? locate(j, k0, pe) : k0[0]];
                        const float dx = cx - p.x;
                        const bool hit = j < L && fabsf(dx) < rabs && sqdist3(dx, cy - p.y, cz - p.z) < radius2;
                        const uint64_t mask = __ballot(hit);
                        const int pos = H + mbcnt(mask);
                        if (hit && pos < BQC_HCAP) hits[pos] = (uint16_t)__float_as_int(p.w);
                        H += __popcll(mask);
                    }
                    if (H > BQC_HCAP) ordered = true;
                }
                if (ordered) {
                    // pathological density: the reference's ordered scan with early exit, 64 points per step
                    for (int q0 = 0; q0 < n && cnt < nsample; q0 += 64) {
                        const int q = q0 + lane;
                        bool hit = false;
                        if (q < n) hit = sqdist3(cx - xyz[q * 3 + 0], cy - xyz[q * 3 + 1], cz - xyz[q * 3 + 2]) < radius2;
                        const uint64_t mask = __ballot(hit);
                        const int pos = cnt + mbcnt(mask);
                        if (hit && pos < nsample) row[pos] = (uint16_t)q;
                        cnt += __popcll(mask);
                    }
                    cnt = min(cnt, nsample);
                } else if (H > nsample) {
                    // ---- 3. threshold: T = the nsample-th smallest index among the hits
                    int T = 0;
                    for (int bit = id_bits - 1; bit >= 0; --bit) {
                        const int trial = T | (1 << bit);
                        int below = 0;
                        for (int u = 0; u < H; u += 64) {
                            const int i = u + lane;
                            below += __popcll(__ballot(i < H && (int)hits[i] < trial));
                        }
                        if (below < nsample) T = trial;                 // fewer than nsample hits below `trial`: the threshold is >= trial
                    }
                    int kept = 0;
                    for (int u = 0; u < H; u += 64) {
                        const int i = u + lane;
                        const int v = i < H ? (int)hits[i] : 0x7fffffff;
                        const bool keep = v <= T;
                        const uint64_t mask = __ballot(keep);
                        if (keep) hits[kept + mbcnt(mask)] = (uint16_t)v; // in place: a survivor moves to the left of every unread entry
                        kept += __popcll(mask);
                    }
                    H = kept;                                           // == nsample (indices are distinct)
                }
            } else if (H > nsample) {
                // <= 64 candidates, more hits than the list takes (nsample < 64): the lane's own rank decides
                const int v = lane < H ? (int)hl[lane] : 0x7fffffff;
                int rank = 0;
                for (int i = 0; i < H; ++i) rank += (int)hl[i] < v;
                if (lane < H && rank < nsample) row[rank] = (uint16_t)v;
                cnt = nsample;
                ordered = true;                                         // (the row is written)
            }
            if (!ordered) {
                // ---- 4. order: rank of each survivor = number of smaller survivors
                // (round 6, measured and not kept: the select on two hits per lane in registers and the ranking through v_readlane instead of
                // LDS broadcast reads -- no difference on the c2 block or the searches of a c3 batch, profiles/r06_bq_emit_anatomy.txt)
                cnt = H;
                if (lane < cnt) {
                    const int v = (int)hl[lane];
                    int rank = 0;
                    for (int i = 0; i < cnt; ++i) rank += (int)hl[i] < v;
                    row[rank] = (uint16_t)v;
                }
            }
        }
        // pad with the first (smallest) entry; a centre without a hit: zeros (only the fused emit reads them)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint16_t first = cnt > 0 ? row[0] : (uint16_t)0;
        for (int s2 = cnt + lane; s2 < nsample; s2 += 64) row[s2] = first;
        if (lane == 0) cnt_s[c] = (m0 + c < m) ? cnt : 0;
    }
}

struct BqScale2 { float radius; int nsample; int32_t *idx_out, *rowc, *rowsrc, *total; };
template <bool FUSED, int NW>
__global__ __launch_bounds__(64 * NW, (FUSED && NW == 4) ? BQC_FUSED_WPE : 1) void ball_query_grid_coop_kernel(int nb, int n, int m, int c_feat, float radius0, int nsample0, int use_xyz,
                                                                   const float *__restrict__ xyz, const char *__restrict__ ws,
                                                                   const float *__restrict__ new_xyz, const float *__restrict__ features,
                                                                   int32_t *__restrict__ idx_out0, float *__restrict__ out,
                                                                   int32_t *__restrict__ rowc0, int32_t *__restrict__ rowsrc0,
                                                                   int32_t *__restrict__ total0, BqScale2 second) {
    const bool sc2 = !FUSED && blockIdx.y == 1;
    const float radius = sc2 ? second.radius : radius0;
    const int nsample = sc2 ? second.nsample : nsample0;
    int32_t *__restrict__ idx_out = sc2 ? second.idx_out : idx_out0;
    int32_t *__restrict__ rowc = sc2 ? second.rowc : rowc0;
    int32_t *__restrict__ rowsrc = sc2 ? second.rowsrc : rowsrc0;
    int32_t *__restrict__ total = sc2 ? second.total : total0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4 *cen = reinterpret_cast<float4 *>(smem);                    // 64
    int *cnt_s = reinterpret_cast<int *>(cen + 64);                    // 64
    uint16_t *rows = reinterpret_cast<uint16_t *>(cnt_s + 64);         // 64 * rstride
    const int rstride = nsample + 1;
    uint16_t *hits_all = rows + ((64 * rstride + 7) & ~7);            // NW waves x BQC_HCAP
    uint16_t *stage_all = hits_all + NW * BQC_HCAP;                    // NW waves x CPW centres x 64: the hits of each centre's first 64 candidates
    const int tiles = (m + 63) / 64;
    int b, tile;
    if ((nb & 7) == 0) {
        const int g = blockIdx.x, j = g >> 3;
        b = (j / tiles) * 8 + (g & 7);
        tile = j - (j / tiles) * tiles;
    } else {
        b = blockIdx.x / tiles;
        tile = blockIdx.x - b * tiles;
    }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = tile * 64;
    xyz += (size_t)b * n * 3;
    const char *base = ws + (size_t)b * bin_scene_stride(n);
    new_xyz += (size_t)b * m * 3;
    bqc_search_tile<NW>(lane, w, m0, n, m, radius, nsample, xyz, base, new_xyz, cen, cnt_s, rows, rstride, hits_all, stage_all);
    __syncthreads();
    if (!FUSED && rowc) {
        // ---- the compact (centre, source) pairs of these 64 lists (gemm_pool.hip pair_compact_kernel, here without a launch of
        // its own): a centre contributes its distinct hits -- one pair (centre, point 0) when it has none, like its all-zero list --
        // and the workgroup reserves its stretch of compact rows with ONE atomic add on *total (zero on entry)
        int *pfx = reinterpret_cast<int *>(hits_all);                  // 65 ints: the hit lists are dead
        if (w == 0) {
            const int k = (m0 + lane < m) ? max(cnt_s[lane], 1) : 0;
            int v = k;
            for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(v, o); if (lane >= o) v += t2; }
            int base_row = 0;
            if (lane == 63) base_row = atomicAdd(total, v);
            base_row = __builtin_amdgcn_readlane(base_row, 63);
            pfx[lane] = base_row + v - k;
            if (lane == 63) pfx[64] = base_row + v;
        }
        __syncthreads();
        if (tid < 256) {
            const int c = tid >> 2;
            const int first = pfx[c], k = pfx[c + 1] - first;
            const int32_t cm = (int32_t)((size_t)b * m + m0 + c);
            for (int s2 = tid & 3; s2 < k; s2 += 4) {
                rowc[first + s2] = cm;
                rowsrc[first + s2] = (int32_t)rows[(size_t)c * rstride + s2];
            }
        }
    }
#ifdef WS3D_BQC_NO_EMIT      // ablation: one word per workgroup keeps the search alive
    if (tid == 0 && idx_out) idx_out[((size_t)b * m + m0) * nsample] = (int)rows[0] + cnt_s[0];
#else
    // (the hit lists are dead behind the barrier: their first KB parks the centred first hit of every centre for the padding of the 3 + 1 channel emit.
    // Fetched by the searching waves into a KB of its own instead -- no barrier inside the emit -- the workgroup's LDS crosses the allocation
    // granule that leaves six workgroups per CU: 0.85 -> 0.92 ms.)
    bq_emit<uint16_t, FUSED, 64 * NW, 64>(b, tid, m0, n, m, c_feat, nsample, use_xyz, xyz, features, idx_out, out, rows, rstride, cnt_s, cen, nb,
                                          FUSED ? reinterpret_cast<float4 *>(hits_all) : nullptr);
#endif
}


static size_t bq_smem(int nsample, size_t idx_bytes) {
    return sizeof(float4) * (BQ_NW * BQ_TILE + 64) + sizeof(int) * BQ_NW * 64 +
           idx_bytes * (size_t)BQ_NW * 64 * (nsample + 1);
}

template <bool FUSED>
static int bq_launch(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                     const float *xyz, const float *new_xyz, const float *features, int32_t *idx,
                     float *out, const void *sorted, hipStream_t st, const char *what, int nlc = 0,
                     int32_t *rowc = nullptr, int32_t *rowsrc = nullptr, int32_t *total = nullptr) {
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
    // channel split (fused only): enough workgroups to fill 256 CUs, at least 8 channels each
    int gz = 1;
    use_xyz = FUSED ? ((use_xyz ? 1 : 0) | (nlc ? 2 : 0))   // bit 1 selects the channels-last epilogue (row-wise, no channel split)
                    : (use_xyz & 4);                        // index-only kernels: bit 2 = write zero rows for centres without a hit
    if (FUSED && c >= 16 && !nlc) {
        const long tiles = (long)b * ((m + 63) / 64);
        while (gz < 64 && tiles * gz < 1024 && c / (gz * 2) >= 8) gz *= 2;
    }
    if (FUSED && nlc && c >= 16) {   // row slices: at least 64 rows per workgroup
        const long tiles = (long)b * ((m + 63) / 64);
        while (gz < 64 && tiles * gz < 1024 && (64L * nsample) / (gz * 2) >= 64) gz *= 2;
    }
    if (sorted && n <= SORT_MAX_N && b <= 65535 && grid_flavour(sorted)) {
        // one wave per centre; 16 waves x 4 centres per tile when the tiles alone do not fill the chip, else 4 x 16
        const bool wide = (long)b * ((m + 63) / 64) * gz < BQC_WIDE_BELOW;
        const int nw = wide ? 16 : BQC_LARGE_NW;
        size_t smem_c = sizeof(float4) * 64 + sizeof(int) * 64 + sizeof(uint16_t) * (size_t)(((64 * (nsample + 1) + 7) & ~7) + nw * BQC_HCAP + 64 * 64);
#ifdef WS3D_BQC_EXTRA_LDS     // occupancy probe (scripts/r06/bq_variants2.sh): unused LDS bytes per workgroup
        smem_c += WS3D_BQC_EXTRA_LDS;
#endif
        if (nsample <= 64 && smem_c <= 64 * 1024) {      // longer lists: one lane per centre (below)
            if (wide)
                hipLaunchKernelGGL((ball_query_grid_coop_kernel<FUSED, 16>), dim3((unsigned)(b * ((m + 63) / 64)), 1, gz), dim3(1024), smem_c, st, b, n, m, c,
                                   radius, nsample, use_xyz, xyz, reinterpret_cast<const char *>(sorted), new_xyz, features, idx, out, rowc, rowsrc, total, BqScale2{});
            else
                hipLaunchKernelGGL((ball_query_grid_coop_kernel<FUSED, BQC_LARGE_NW>), dim3((unsigned)(b * ((m + 63) / 64)), 1, gz), dim3(64 * BQC_LARGE_NW), smem_c, st, b, n, m, c,
                                   radius, nsample, use_xyz, xyz, reinterpret_cast<const char *>(sorted), new_xyz, features, idx, out, rowc, rowsrc, total, BqScale2{});
            return check_launch(what);
        }
        if (rowc) { set_error("%s: nsample %d is not covered by the kernel that emits the pairs", what, nsample); return WS3D_E_UNSUPPORTED; }
        const size_t smem_g = sizeof(float4) * 64 + sizeof(int) * 64 + sizeof(uint16_t) * (size_t)64 * (nsample + 1);
        if (smem_g <= 64 * 1024) {
            hipLaunchKernelGGL((ball_query_grid_kernel<FUSED>), dim3((unsigned)(b * ((m + 63) / 64)), 1, gz), dim3(256), smem_g, st, b, n, m, c,
                               radius, nsample, use_xyz, xyz, reinterpret_cast<const char *>(sorted), new_xyz, features, idx, out);
            return check_launch(what);
        }
    }
    if (rowc) { set_error("%s: the pairs come out of the fine-grid kernel only (sort_points_grid buffer, n <= %d)", what, SORT_MAX_N); return WS3D_E_UNSUPPORTED; }
    if (sorted && n <= SORT_MAX_N && b <= 65535) {
        const size_t smem_s = sizeof(float4) * 64 + sizeof(int) * 256 +
                              sizeof(uint16_t) * (size_t)5 * 64 * (nsample + 1);
        if (smem_s <= 150 * 1024) {
            if (int rc = raise_lds_cap((const void *)ball_query_sorted_kernel<FUSED>, smem_s, what)) return rc;
            hipLaunchKernelGGL((ball_query_sorted_kernel<FUSED>), dim3((m + 63) / 64, b, gz), dim3(256), smem_s, st, n,
                               m, c, radius, nsample, use_xyz, xyz, reinterpret_cast<const char *>(sorted),
                               new_xyz, features, idx, out);
            return check_launch(what);
        }
    }
    const bool small_idx = n <= 65536;  // neighbour lists held as uint16 in LDS
    const size_t smem = bq_smem(nsample, small_idx ? 2 : 4);
    if (smem > 160 * 1024 || b > 65535) {
        set_error("%s: nsample=%d needs %zu B of LDS (> 160 KiB) or batch > 65535", what, nsample, smem);
        return WS3D_E_UNSUPPORTED;
    }
    dim3 grid((m + 63) / 64, b, gz);
    if (small_idx) {
        if (int rc = raise_lds_cap((const void *)ball_query_kernel<uint16_t, FUSED>, smem, what)) return rc;
        hipLaunchKernelGGL((ball_query_kernel<uint16_t, FUSED>), grid, dim3(64 * BQ_NW), smem, st, n, m, c,
                           radius, nsample, use_xyz, xyz, new_xyz, features, idx, out);
    } else {
        if (int rc = raise_lds_cap((const void *)ball_query_kernel<int32_t, FUSED>, smem, what)) return rc;
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

// plane % 4 == 0: 4 consecutive (m, s) slots per lane -- one 16-byte index load, 4 gathers per
// channel, 16-byte stores; 4 channels in flight
__global__ __launch_bounds__(256) void group_points_vec4_kernel(int c, int n, int plane,
                                                                const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                float *__restrict__ out, int stream) {
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * GRP_CCH;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= plane) return;
    const int4 id = *reinterpret_cast<const int4 *>(idx + (size_t)b * plane + e);
    const float *src = points + ((size_t)b * c + c0) * n;
    float *dst = out + ((size_t)b * c + c0) * plane + e;
    const int cc = min(GRP_CCH, c - c0);
    int ch = 0;
    for (; ch + 4 <= cc; ch += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float *f = src + (size_t)(ch + u) * n;
            v[u] = make_float4(f[id.x], f[id.y], f[id.z], f[id.w]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) st4(dst + (size_t)(ch + u) * plane, v[u], stream);
    }
    for (; ch < cc; ++ch) {
        const float *f = src + (size_t)ch * n;
        st4(dst + (size_t)ch * plane, make_float4(f[id.x], f[id.y], f[id.z], f[id.w]), stream);
    }
}

// The channel rows of the gather staged in LDS (round 6): the kernel above fetches every output word with a 4-byte gather out of
// L1 / L2 -- about one lane per clock and CU, 2.6 TB/s of output whatever the batch (bench.py --workload ops: 0.32-0.34 of 8 TB/s at SA2's
// 96 channels x 1024 x 48) -- while the source of a channel group is only GRL_CCH x n floats.  Here a workgroup copies its GRL_CCH
// channel rows of ONE scene into LDS with coalesced 16-byte loads (64 KB at n = 4096) and then walks the whole (m, s) plane: one
// 16-byte index load per lane and trip, four LDS reads and one 16-byte store per channel.  grid (channel groups, scenes).
constexpr int GRL_CCH = 4, GRL_THREADS = 1024;
__global__ __launch_bounds__(GRL_THREADS) void group_points_lds_kernel(int c, int n, int plane, const float *__restrict__ points,
                                                                       const int32_t *__restrict__ idx, float *__restrict__ out, int stream) {
    extern __shared__ __attribute__((aligned(16))) float grl_rows[];          // [cc][n]
    const int b = blockIdx.y, c0 = blockIdx.x * GRL_CCH, tid = threadIdx.x;
    const int cc = min(GRL_CCH, c - c0);
    const float *src = points + ((size_t)b * c + c0) * n;
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        for (int i = tid; i < cc * n / 4; i += GRL_THREADS) reinterpret_cast<float4 *>(grl_rows)[i] = reinterpret_cast<const float4 *>(src)[i];
    } else {
        for (int i = tid; i < cc * n; i += GRL_THREADS) grl_rows[i] = src[i];
    }
    __syncthreads();
    const int32_t *ib = idx + (size_t)b * plane;
    float *dst = out + ((size_t)b * c + c0) * plane;
    for (int e = tid * 4; e < plane; e += GRL_THREADS * 4) {
        const int4 id = *reinterpret_cast<const int4 *>(ib + e);
        float4 v[GRL_CCH];
#pragma unroll
        for (int u = 0; u < GRL_CCH; ++u) {
            const float *f = grl_rows + (u < cc ? u : 0) * n;
            v[u] = make_float4(f[id.x], f[id.y], f[id.z], f[id.w]);
        }
#pragma unroll
        for (int u = 0; u < GRL_CCH; ++u)
            if (u < cc) st4(dst + (size_t)u * plane + e, v[u], stream);
    }
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
    if (!grad && (plane & 3) == 0 && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(idx)) & 15) == 0) {
        const int stream = (size_t)b * c * plane * sizeof(float) > STREAM_STORE_BYTES;
        // LDS-staged rows where a channel group's source fits (<= 64 KB: n <= 4096) and the plane is long enough to pay for the staging
        // (every output word needs its index in range: the reference's kernel does not check either, group_points_gpu.cu:47-66)
        static const bool no_lds = getenv("WS3D_GROUP_NO_LDS") != nullptr;
        if (!no_lds && c >= GRL_CCH && (size_t)n * GRL_CCH * sizeof(float) <= 64 * 1024 && plane >= 2L * n) {
            const size_t lds = (size_t)n * GRL_CCH * sizeof(float);
            hipLaunchKernelGGL(group_points_lds_kernel, dim3((c + GRL_CCH - 1) / GRL_CCH, b), dim3(GRL_THREADS), lds, st, c, n, (int)plane, src, idx, dst, stream);
            return check_launch(what);
        }
        grid.x = (unsigned)((plane / 4 + 255) / 256);
        hipLaunchKernelGGL(group_points_vec4_kernel, grid, dim3(256), 0, st, c, n, (int)plane, src, idx, dst, stream);
        return check_launch(what);
    }
    if (grad)
        hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(256), 0, st, c, n, (int)plane, src, idx, dst);
    else
        hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, st, c, n, (int)plane, src, idx, dst);
    return check_launch(what);
}

}  // namespace ws3d

extern "C" size_t ws3d_sorted_points_bytes(int b, int n) {
    if (b <= 0 || n <= 0 || n > ws3d::SORT_MAX_N) return 0;
    return (size_t)b * ws3d::bin_scene_stride(n);
}

extern "C" int ws3d_sort_points_x(int b, int n, const float *xyz, void *sorted, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || n > SORT_MAX_N || !xyz || !sorted) {
        set_error("ws3d_sort_points_x: invalid argument (b=%d n=%d; n must be <= %d)", b, n, SORT_MAX_N);
        return WS3D_E_INVALID;
    }
    if (b == 0) return WS3D_OK;
    hipLaunchKernelGGL(bin_points_x_kernel, dim3(b), dim3(1024), 0, as_stream(stream), n, xyz,
                       reinterpret_cast<char *>(sorted));
    note_flavour(sorted, 0);
    return check_launch("ws3d_sort_points_x");
}

extern "C" int ws3d_sort_points_grid(int b, int n, const float *xyz, void *sorted, ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || n > SORT_MAX_N || !xyz || !sorted) {
        set_error("ws3d_sort_points_grid: invalid argument (b=%d n=%d, n <= %d)", b, n, SORT_MAX_N);
        return WS3D_E_INVALID;
    }
    if (b == 0) return WS3D_OK;
    hipLaunchKernelGGL(bin_points_grid_kernel, dim3(b), dim3(1024), 0, as_stream(stream), n, xyz, reinterpret_cast<char *>(sorted));
    note_flavour(sorted, 1);
    return check_launch("ws3d_sort_points_grid");
}

extern "C" int ws3d_sort_points_jobs(int b, int njobs, const int *n, const int *kind, const float *const *xyz, void *const *sorted,
                                     ws3d_stream_t stream) {
    using namespace ws3d;
    if (b == 0 || njobs == 0) return WS3D_OK;
    if (b < 0 || njobs < 0 || njobs > BIN_MAX_JOBS || !n || !kind || !xyz || !sorted) {
        set_error("ws3d_sort_points_jobs: invalid argument (b=%d njobs=%d, at most %d jobs)", b, njobs, BIN_MAX_JOBS);
        return WS3D_E_INVALID;
    }
    BinJobs jobs{};
    for (int i = 0; i < njobs; ++i) {
        if (n[i] <= 0 || n[i] > SORT_MAX_N || (kind[i] != 0 && kind[i] != 1) || !xyz[i] || !sorted[i]) {
            set_error("ws3d_sort_points_jobs: job %d invalid (n=%d kind=%d; 0 < n <= %d, kind 0 = grid, 1 = xz)", i, n[i], kind[i], SORT_MAX_N);
            return WS3D_E_INVALID;
        }
        jobs.j[i] = BinJob{xyz[i], reinterpret_cast<char *>(sorted[i]), n[i], kind[i]};
    }
    for (int i = 0; i < njobs; ++i)
        note_flavour(sorted[i], kind[i] == 0 ? 1 : 0);        // (as ws3d_sort_points_grid: the ball-query entries pick the fine-grid kernels by buffer address)
    hipLaunchKernelGGL(bin_points_jobs_kernel, dim3(b, njobs), dim3(1024), 0, as_stream(stream), jobs);
    return check_launch("ws3d_sort_points_jobs");
}

extern "C" int ws3d_ball_query(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                               const float *xyz, int32_t *idx, const void *sorted, ws3d_stream_t stream) {
    return ws3d::bq_launch<false>(b, n, m, 0, radius, nsample, 0, xyz, new_xyz, nullptr, idx, nullptr, sorted,
                                  ws3d::as_stream(stream), "ws3d_ball_query");
}

extern "C" int ws3d_ball_query_fill(int b, int n, int m, float radius, int nsample, const float *new_xyz,
                                    const float *xyz, int32_t *idx, const void *sorted, ws3d_stream_t stream) {
    return ws3d::bq_launch<false>(b, n, m, 0, radius, nsample, 4, xyz, new_xyz, nullptr, idx, nullptr, sorted,
                                  ws3d::as_stream(stream), "ws3d_ball_query_fill");
}

extern "C" int ws3d_ball_query_pairs(int b, int n, int m, float radius, int nsample, const float *new_xyz, const float *xyz, int32_t *idx,
                                     const void *sorted_grid, int32_t *rowc, int32_t *rowsrc, int32_t *total, ws3d_stream_t stream) {
    using namespace ws3d;
    if (!rowc || !rowsrc || !total || !sorted_grid) { set_error("ws3d_ball_query_pairs: NULL argument"); return WS3D_E_INVALID; }
    return bq_launch<false>(b, n, m, 0, radius, nsample, 4, xyz, new_xyz, nullptr, idx, nullptr, sorted_grid, as_stream(stream),
                            "ws3d_ball_query_pairs", 0, rowc, rowsrc, total);
}

extern "C" int ws3d_ball_query_pairs2(int b, int n, int m, const float *new_xyz, const float *xyz, const void *sorted_grid,
                                      float radius0, int nsample0, int32_t *idx0, int32_t *rowc0, int32_t *rowsrc0, int32_t *total0,
                                      float radius1, int nsample1, int32_t *idx1, int32_t *rowc1, int32_t *rowsrc1, int32_t *total1,
                                      ws3d_stream_t stream) {
    using namespace ws3d;
    if (b < 0 || n <= 0 || m < 0 || nsample0 <= 0 || nsample1 <= 0 || !xyz || !new_xyz || !sorted_grid || !idx0 || !rowc0 || !rowsrc0 || !total0 ||
        !idx1 || !rowc1 || !rowsrc1 || !total1) {
        set_error("ws3d_ball_query_pairs2: invalid argument (b=%d n=%d m=%d nsample=%d/%d)", b, n, m, nsample0, nsample1);
        return WS3D_E_INVALID;
    }
    if (b == 0 || m == 0) return WS3D_OK;
    const int ns = nsample0 > nsample1 ? nsample0 : nsample1;
    const long tiles = (long)b * ((m + 63) / 64);
    const bool wide = tiles * 2 < BQC_WIDE_BELOW;
    // ws3d_tune key 5 (round 6): 8 waves x 8 centres per tile instead of 16 x 4 in the launches that do not fill the chip -- a longer launch
    // (level 2, r = 1.0: 26 -> 36 us alone) that holds half the waves: about +1 % on the 20-deep step (profiles/r06_bq_emit_anatomy.txt, section 10).
    // Stage1Pipeline sets it while it captures at depth >= 4; a lone caller keeps 16.  Same lists, same pair sets.
    constexpr int TUNE_BQ_WIDE_NW = 5;                         // (ws3d_tune key 5; the keys 0-4 are named in common.h)
    static_assert(TUNE_BQ_WIDE_NW < TUNE_COUNT, "ws3d_tune key");
    const bool wide8 = wide && g_tune[TUNE_BQ_WIDE_NW] == 8;
    const int nw = wide ? (wide8 ? 8 : 16) : BQC_LARGE_NW;
    const size_t smem_c = sizeof(float4) * 64 + sizeof(int) * 64 + sizeof(uint16_t) * (size_t)(((64 * (ns + 1) + 7) & ~7) + nw * BQC_HCAP + 64 * 64);
    if (n > SORT_MAX_N || b > 65535 || tiles > 0x7fffffffL || ns > 64 || smem_c > 64 * 1024 || !grid_flavour(sorted_grid)) {
        set_error("ws3d_ball_query_pairs2: not covered (n=%d nsample=%d/%d, or the binned copy is not a fine-grid one): use ws3d_ball_query_pairs per scale", n, nsample0, nsample1);
        return WS3D_E_UNSUPPORTED;
    }
    const BqScale2 second{radius1, nsample1, idx1, rowc1, rowsrc1, total1};
    hipStream_t st = as_stream(stream);
    if (wide8)
        hipLaunchKernelGGL((ball_query_grid_coop_kernel<false, 8>), dim3((unsigned)tiles, 2, 1), dim3(512), smem_c, st, b, n, m, 0, radius0, nsample0, 4, xyz,
                           reinterpret_cast<const char *>(sorted_grid), new_xyz, (const float *)nullptr, idx0, (float *)nullptr, rowc0, rowsrc0, total0, second);
    else if (wide)
        hipLaunchKernelGGL((ball_query_grid_coop_kernel<false, 16>), dim3((unsigned)tiles, 2, 1), dim3(1024), smem_c, st, b, n, m, 0, radius0, nsample0, 4, xyz,
                           reinterpret_cast<const char *>(sorted_grid), new_xyz, (const float *)nullptr, idx0, (float *)nullptr, rowc0, rowsrc0, total0, second);
    else
        hipLaunchKernelGGL((ball_query_grid_coop_kernel<false, BQC_LARGE_NW>), dim3((unsigned)tiles, 2, 1), dim3(64 * BQC_LARGE_NW), smem_c, st, b, n, m, 0, radius0,
                           nsample0, 4, xyz, reinterpret_cast<const char *>(sorted_grid), new_xyz, (const float *)nullptr, idx0, (float *)nullptr, rowc0, rowsrc0,
                           total0, second);
    return check_launch("ws3d_ball_query_pairs2");
}

extern "C" int ws3d_query_and_group(int b, int n, int m, int c, float radius, int nsample,
                                    int use_xyz, const float *xyz, const float *new_xyz,
                                    const float *features, int32_t *idx_out, float *out,
                                    const void *sorted, ws3d_stream_t stream) {
    return ws3d::bq_launch<true>(b, n, m, features ? c : 0, radius, nsample, use_xyz, xyz, new_xyz,
                                 features, idx_out, out, sorted, ws3d::as_stream(stream), "ws3d_query_and_group");
}

extern "C" int ws3d_query_and_group_nlc(int b, int n, int m, int c, float radius, int nsample, int use_xyz,
                                        const float *xyz, const float *new_xyz, const float *features_nlc,
                                        int32_t *idx_out, float *out_nlc, const void *sorted, ws3d_stream_t stream) {
    return ws3d::bq_launch<true>(b, n, m, features_nlc ? c : 0, radius, nsample, use_xyz, xyz, new_xyz, features_nlc,
                                 idx_out, out_nlc, sorted, ws3d::as_stream(stream), "ws3d_query_and_group_nlc", 1);
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
