// fps.hip -- furthest point sampling for gfx950 (replaces pointnet2_cuda.
// furthest_point_sampling_wrapper, sampling.cpp:36-46 / sampling_gpu.cu:93-253).
//
// Design.  FPS is (m-1) strictly dependent argmax steps; the
// step is ALU/latency-bound, not HBM-bound, so the whole scene lives ON-CHIP for the
// entire kernel: one workgroup per scene, every thread keeps PPT points (x,y,z) and
// their running min-distance in VGPRs (16384 points = 16 per thread of a 1024-thread
// workgroup = 256 KB of the CU's 512 KB register file).  HBM is touched once on entry
// (n*12 B) and once per step for the 4-byte index (+12 B of coordinates when the
// fused gather output is requested).
//
// Measured anatomy of one step at N=16384 (us): distance/min/argmax VALU 0.55, wave argmax
// 0.31, cross-wave exchange 0.43, winner-coordinate extraction 0.07 (ablation builds,
// DESIGN.md section 5.1).  A variant that replaces the wave argmax + records by ONE 64-bit
// LDS atomic max per 16-lane row was correct but slower: ds_max_u64 costs ~600 cycles on
// gfx950 (scripts/ubench/lat.hip).
//
// One step = (1) PPT x {3 sub, mul, 2 fma, min, cmp, 2 select} per lane,
//            (2) wave argmax: 4 DPP max steps + 4 readlanes + ballot/ctz (no LDS),
//            (3) one 20-byte LDS record per wave, ONE s_barrier (records are double
//                buffered by step parity), every wave reduces the <=16 records with a
//                16-lane DPP row reduction and picks up the winner's coordinates with
//                readlanes -- no dependent global load anywhere in the loop.
// The reference needs 1 pass over global memory + 11 __syncthreads per step.
//
// Tie-breaks are part of the contract (bit-exact indices).  In the reference, thread
// tid of a bs-thread block (bs = opt_n_threads(n)) owns k = tid, tid+bs, ...; strict
// '>' keeps the smallest k inside a thread and the shared-memory tree keeps the lower
// slot on ties, so among exact ties the winner is the candidate with the smallest
// bit-reversed (k mod bs), then the smallest k.  Here thread u owns the contiguous
// range [u*PPT, (u+1)*PPT) of positions p = bitrev(k mod bs) * S + k / bs
// (S = ceil(n/bs)), so that plain "lowest slot, lowest lane, lowest wave wins" IS the
// reference's tie order -- for any workgroup size, independent of bs.
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef WS3D_FPS_PACKED
#define WS3D_FPS_PACKED 1
#endif
#ifndef WS3D_FPS_CHAINS
#define WS3D_FPS_CHAINS 0  // neutral (1.20 vs 1.17 us/step): the sweep is issue-bound, cmp/cndmask cost 2 slots each
#endif
#ifndef WS3D_FPS_GMAX
#define WS3D_FPS_GMAX 0  // group maxima in the sweep + slot search afterwards: sweep -260 clk, search +370 clk => 1.23 vs 1.17 us/step
#endif
#ifndef WS3D_FPS_TREE
#define WS3D_FPS_TREE 0  // measured 1.9x SLOWER: v_cmp->SGPR-pair->v_cndmask chains stall (scripts/ubench/lat.hip)
#endif

#ifdef WS3D_FPS_PROF
__device__ long long g_fps_prof[16 * 8];
#define PROF_DECL long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long prof_last = clock64();
#define PROF(i) { const long long now_ = clock64(); prof_acc[i] += now_ - prof_last; prof_last = now_; }
#define PROF_STORE if (lane == 0 && blockIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) g_fps_prof[w * 8 + i_] = prof_acc[i_]; }
#else
#define PROF_DECL
#define PROF(i)
#define PROF_STORE
#endif

namespace ws3d {

__device__ __forceinline__ int bitrev_bits(int v, int bits) {
    return bits == 0 ? 0 : (int)(__builtin_bitreverse32((uint32_t)v) >> (32 - bits));
}

template <int N> struct VecF { typedef float type __attribute__((ext_vector_type(N))); };
template <> struct VecF<1> { typedef float type; };
template <int N> using vecf = typename VecF<N>::type;
template <int N> __device__ __forceinline__ float vec_get(const vecf<N> &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float vec_get<1>(const vecf<1> &v, int) { return v; }
template <int N> __device__ __forceinline__ void vec_set(vecf<N> &v, int i, float x) { v[i] = x; }
template <> __device__ __forceinline__ void vec_set<1>(vecf<1> &v, int, float x) { v = x; }

// pairwise tournament with compile-time widths only (fully static register indexing)
template <int W>
__device__ __forceinline__ void tourney(const float (&v)[W], const int (&i)[W], float &best, int &bslot) {
    if constexpr (W == 1) {
        best = v[0];
        bslot = i[0];
    } else {
        float v2[W / 2];
        int i2[W / 2];
#pragma unroll
        for (int s = 0; s < W / 2; ++s) {
            const bool gt = v[2 * s + 1] > v[2 * s];   // lower slot wins ties
            v2[s] = gt ? v[2 * s + 1] : v[2 * s];
            i2[s] = gt ? i[2 * s + 1] : i[2 * s];
        }
        tourney<W / 2>(v2, i2, best, bslot);
    }
}

template <int PPT, int NT>
__global__ __launch_bounds__(NT) void fps_reg_kernel(const float *__restrict__ xyz,
                                                     float *__restrict__ temp,
                                                     int32_t *__restrict__ idx,
                                                     float *__restrict__ new_xyz, int n, int m,
                                                     int bs, int log2bs, int S) {
    constexpr int NW = NT / WS3D_WAVE;
    __shared__ float4 s_cand[2][16];
    __shared__ int s_k[2][16];

    const int b = blockIdx.x;
    xyz += (size_t)b * n * 3;
    idx += (size_t)b * m;
    if (temp) temp += (size_t)b * n;
    if (new_xyz) new_xyz += (size_t)b * m * 3;

    const int u = threadIdx.x;
    const int lane = u & 63;
    const int w = u >> 6;

    // ext-vector storage: a wave-uniform dynamic index (the winner's slot) then lowers to
    // VGPR-indexed moves instead of a 16-way compare/branch chain.
    vecf<PPT> px, py, pz;
    float t[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int p = u * PPT + s;
        const int rb = p / S, sl = p - rb * S;
        const int k = bitrev_bits(rb, log2bs) + sl * bs;
        const bool valid = (rb < bs) && (k < n);
        vec_set<PPT>(px, s, valid ? xyz[k * 3 + 0] : 0.f);
        vec_set<PPT>(py, s, valid ? xyz[k * 3 + 1] : 0.f);
        vec_set<PPT>(pz, s, valid ? xyz[k * 3 + 2] : 0.f);
        t[s] = valid ? (temp ? temp[k] : 1e10f) : -1.0f;  // -1 never beats a real candidate (>= 0)
    }

    int old = 0;
    float ox = xyz[0], oy = xyz[1], oz = xyz[2];
    if (u == 0) {
        idx[0] = 0;
        if (new_xyz) { new_xyz[0] = ox; new_xyz[1] = oy; new_xyz[2] = oz; }
    }

    PROF_DECL
    for (int j = 1; j < m; ++j) {
        PROF(0)
        float best = -1.0f;
        int bslot = 0;
        constexpr bool GMAX = PPT >= 8 && NT > 64 && WS3D_FPS_GMAX;
        constexpr int GS = PPT >= 4 ? PPT / 4 : 1;   // slots per group
        float gm[4] = {-1.0f, -1.0f, -1.0f, -1.0f};
        if constexpr (GMAX) {
            // The running (best, slot) pair costs a v_cmp and two v_cndmask per point -- 6 issue
            // slots of the 13 a point costs (compares and selects issue at half rate here).  Only
            // the VALUE is needed until the workgroup winner is known: keep 4 group maxima per
            // lane (one v_max per point) and look the slot up afterwards, in the winning lane,
            // lowest slot first -- the same (slot, lane, wave) tie order as before.
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const float d = sqdist3(vec_get<PPT>(px, s) - ox, vec_get<PPT>(py, s) - oy, vec_get<PPT>(pz, s) - oz);
                const float d2 = min_f32(d, t[s]);  // == fminf: t[s] is never NaN
                t[s] = d2;
                gm[s / GS] = max_f32(gm[s / GS], d2);
            }
            best = max_f32(max_f32(gm[0], gm[1]), max_f32(gm[2], gm[3]));
        } else if constexpr (PPT >= 2 && NT == 64 && WS3D_FPS_PACKED && WS3D_DIST_MODE == 0) {  // pays only for the single-wave shapes (measured)
            // two points per instruction for the distance (v_pk_add/mul/fma_f32: same IEEE
            // results per element as the scalar forms)
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 o_x = {ox, ox}, o_y = {oy, oy}, o_z = {oz, oz};
#pragma unroll
            for (int s = 0; s < PPT; s += 2) {
                const f2 dx = f2{px[s], px[s + 1]} - o_x, dy = f2{py[s], py[s + 1]} - o_y, dz = f2{pz[s], pz[s + 1]} - o_z;
                const f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dx, dx, dy * dy));
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float d2 = min_f32(d[q], t[s + q]);
                    t[s + q] = d2;
                    const bool gt = d2 > best;
                    bslot = gt ? s + q : bslot;
                    best = gt ? d2 : best;
                }
            }
        } else if constexpr (PPT >= 4 && WS3D_FPS_TREE) {
            // all distances first (independent), then a pairwise tournament over the slots: the
            // serial best/bslot recurrence (2 dependent ops per slot) left the SIMD idle ~40 % of
            // the sweep at 2 waves/SIMD.  Lower slot wins ties at every level (strict '>').
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const float d = sqdist3(vec_get<PPT>(px, s) - ox, vec_get<PPT>(py, s) - oy, vec_get<PPT>(pz, s) - oz);
                t[s] = min_f32(d, t[s]);  // == fminf: t[s] is never NaN
            }
            float tv[PPT / 2];
            int ti[PPT / 2];
#pragma unroll
            for (int s = 0; s < PPT / 2; ++s) {
                const bool gt = t[2 * s + 1] > t[2 * s];
                tv[s] = gt ? t[2 * s + 1] : t[2 * s];
                ti[s] = gt ? 2 * s + 1 : 2 * s;
            }
            tourney<PPT / 2>(tv, ti, best, bslot);
        } else if constexpr (PPT >= 4 && WS3D_FPS_CHAINS && WS3D_DIST_MODE == 0) {
            // CH independent argmax chains over contiguous slot ranges, advanced in lock step so
            // that CH distance computations (each a 7-deep dependent chain) and CH compare/select
            // recurrences are in flight at once; the single-chain form left one wave with no
            // ILP (the scheduler reused three temporaries for all 32 points).  Chains are merged
            // in range order with strict '>', i.e. the lowest slot still wins exact ties.
            constexpr int CH = 4, LEN = PPT / CH;
            float cb[CH];
            int cs[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) { cb[c] = -1.0f; cs[c] = c * LEN; }
#pragma unroll
            for (int i = 0; i < LEN; ++i) {
                float dx[CH], dy[CH], dz[CH], d[CH];
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    dx[c] = vec_get<PPT>(px, c * LEN + i) - ox;
                    dy[c] = vec_get<PPT>(py, c * LEN + i) - oy;
                    dz[c] = vec_get<PPT>(pz, c * LEN + i) - oz;
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) d[c] = dy[c] * dy[c];
#pragma unroll
                for (int c = 0; c < CH; ++c) d[c] = __builtin_fmaf(dx[c], dx[c], d[c]);
#pragma unroll
                for (int c = 0; c < CH; ++c) d[c] = __builtin_fmaf(dz[c], dz[c], d[c]);   // == sqdist3
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    d[c] = min_f32(d[c], t[c * LEN + i]);  // == fminf: t is never NaN
                    t[c * LEN + i] = d[c];
                }
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const bool gt = d[c] > cb[c];
                    cs[c] = gt ? c * LEN + i : cs[c];
                    cb[c] = gt ? d[c] : cb[c];
                }
            }
            best = cb[0];
            bslot = cs[0];
#pragma unroll
            for (int c = 1; c < CH; ++c) {
                const bool gt = cb[c] > best;
                bslot = gt ? cs[c] : bslot;
                best = gt ? cb[c] : best;
            }
        } else {
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const float d = sqdist3(vec_get<PPT>(px, s) - ox, vec_get<PPT>(py, s) - oy, vec_get<PPT>(pz, s) - oz);
                const float d2 = min_f32(d, t[s]);  // == fminf: t[s] is never NaN
                t[s] = d2;
                const bool gt = d2 > best;
                bslot = gt ? s : bslot;
                best = gt ? d2 : best;
            }
        }
        PROF(1)
        // ---- wave argmax: lowest lane among the lanes holding the wave maximum
        const float wmax = wave_max(best);
        const uint64_t eq = __ballot(best == wmax);
        const int wl = (int)__builtin_ctzll(eq);
        int wslot;
        if constexpr (GMAX) {
            int gsel = 3;
            gsel = gm[2] == wmax ? 2 : gsel;
            gsel = gm[1] == wmax ? 1 : gsel;
            gsel = gm[0] == wmax ? 0 : gsel;
            const int gw = __builtin_amdgcn_readlane(gsel, wl);   // the winning lane's lowest group holding wmax
            int ls = GS - 1;
            auto find = [&](auto G) {
#pragma unroll
                for (int q = GS - 2; q >= 0; --q) ls = t[decltype(G)::value * GS + q] == wmax ? q : ls;
            };
            switch (gw) {   // wave-uniform: one block of GS statically indexed compares
                case 0: find(std::integral_constant<int, 0>{}); break;
                case 1: find(std::integral_constant<int, 1>{}); break;
                case 2: find(std::integral_constant<int, 2>{}); break;
                default: find(std::integral_constant<int, 3>{}); break;
            }
            wslot = gw * GS + __builtin_amdgcn_readlane(ls, wl);
        } else {
            wslot = __builtin_amdgcn_readlane(bslot, wl);
        }
        PROF(2)
        // VGPR-indexed moves (s_set_gpr_idx): measured faster than a tree of scalar branches over
        // statically indexed registers (1.17 vs 1.48 us/step) -- taken branches are expensive here
        const float cx = readlane_f(vec_get<PPT>(px, wslot), wl);
        const float cy = readlane_f(vec_get<PPT>(py, wslot), wl);
        const float cz = readlane_f(vec_get<PPT>(pz, wslot), wl);
        const int kw = (w * 64 + wl) * PPT + wslot;  // tie-order POSITION; -> point index after the loop
        PROF(3)

        if constexpr (NW == 1) {
            ox = cx; oy = cy; oz = cz; old = kw;
        } else {
            const int buf = j & 1;
            if (lane == 0) {
                s_cand[buf][w] = make_float4(wmax, cx, cy, cz);
                s_k[buf][w] = kw;
            }
            PROF(4)
            lds_barrier();
            PROF(5)
            const int e = lane & 15;
            const float4 c = s_cand[buf][e];
            const int kc = s_k[buf][e];
            const float v = e < NW ? c.x : -2.0f;
            const float vm = row16_max(v);
            const uint64_t eq2 = __ballot(v == vm);
            const int sel = (int)__builtin_ctzll(eq2);  // lowest wave among ties
            ox = readlane_f(c.y, sel);
            oy = readlane_f(c.z, sel);
            oz = readlane_f(c.w, sel);
            old = __builtin_amdgcn_readlane(kc, sel);
            PROF(6)
        }
        if (u == 0) {
            idx[j] = old;
            if (new_xyz) { new_xyz[j * 3 + 0] = ox; new_xyz[j * 3 + 1] = oy; new_xyz[j * 3 + 2] = oz; }
        }
    }

    PROF_STORE
    // positions -> point indices, off the critical path (thread 0 wrote idx[]; same lane reads)
    __syncthreads();
    for (int j = 1 + u; j < m; j += NT) {
        const int p = idx[j];
        const int rb = p / S, sl = p - rb * S;
        idx[j] = bitrev_bits(rb, log2bs) + sl * bs;
    }

    if (temp) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int p = u * PPT + s;
            const int rb = p / S, sl = p - rb * S;
            const int k = bitrev_bits(rb, log2bs) + sl * bs;
            if ((rb < bs) && (k < n)) temp[k] = t[s];
        }
    }
}

// ---- two scenes per CU (throughput mode).  fps_reg_kernel<32,512> needs 234 VGPRs: one workgroup
// per CU, two waves per SIMD that run in lock-step, so the SIMD idles through every reduction
// chain (~45 % of a step).  Here z lives in LDS (64 KB per scene, read-only in the loop, 16-byte
// conflict-free reads) and x, y and the running min-distance stay in VGPRs: <= 128 VGPRs, i.e. two
// workgroups = two independent scenes per CU whose sweeps fill each other's chains (measured:
// 0.98 us per scene-step at 512 scenes vs 1.17; a start-up phase offset between the two scenes
// of a CU changes nothing).  Same arithmetic, same position order, same tie rules as fps_reg_kernel.
template <int PPT, int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4)))
void fps_zlds_kernel(const float *__restrict__ xyz, float *__restrict__ temp, int32_t *__restrict__ idx,
                     float *__restrict__ new_xyz, int n, int m, int bs, int log2bs, int S) {
    constexpr int NW = NT / WS3D_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem_z[];
    float4 *zs4 = reinterpret_cast<float4 *>(smem_z);   // [PPT/4][NT] float4: slots 4c..4c+3 of thread u at zs4[c*NT+u]
    __shared__ float4 s_cand[2][16];
    __shared__ int s_k[2][16];

    const int b = blockIdx.x;
    xyz += (size_t)b * n * 3;
    idx += (size_t)b * m;
    if (temp) temp += (size_t)b * n;
    if (new_xyz) new_xyz += (size_t)b * m * 3;
    const int u = threadIdx.x, lane = u & 63, w = u >> 6;

    vecf<PPT> px, py;
    float t[PPT];
#pragma unroll
    for (int c = 0; c < PPT / 4; ++c) {
        float z4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int s = 4 * c + q;
            const int p = u * PPT + s;
            const int rb = p / S, sl = p - rb * S;
            const int k = bitrev_bits(rb, log2bs) + sl * bs;
            const bool valid = (rb < bs) && (k < n);
            vec_set<PPT>(px, s, valid ? xyz[k * 3 + 0] : 0.f);
            vec_set<PPT>(py, s, valid ? xyz[k * 3 + 1] : 0.f);
            z4[q] = valid ? xyz[k * 3 + 2] : 0.f;
            t[s] = valid ? (temp ? temp[k] : 1e10f) : -1.0f;
        }
        zs4[c * NT + u] = make_float4(z4[0], z4[1], z4[2], z4[3]);
    }
    int old = 0;
    float ox = xyz[0], oy = xyz[1], oz = xyz[2];
    if (u == 0) {
        idx[0] = 0;
        if (new_xyz) { new_xyz[0] = ox; new_xyz[1] = oy; new_xyz[2] = oz; }
    }
    __syncthreads();
    const float *zs = reinterpret_cast<const float *>(smem_z);
    for (int j = 1; j < m; ++j) {
        // Two scenes share the CU, so the SIMD is issue-bound and only the instruction COUNT of a
        // step matters (the chain's latency is covered by the other scene): the sweep keeps 4 group
        // maxima per lane (one v_max per point instead of v_cmp + 2 v_cndmask) and the winner's
        // slot is looked up afterwards -- lowest slot first, same (slot, lane, wave) tie order.
        constexpr int GS = PPT / 4;
        float gm[4] = {-1.0f, -1.0f, -1.0f, -1.0f};
        float4 zn = zs4[u];                          // software pipeline: chunk c+1 is in flight while c is consumed
#pragma unroll
        for (int c = 0; c < PPT / 4; ++c) {
            const float4 z4 = zn;
            if (c + 1 < PPT / 4) zn = zs4[(c + 1) * NT + u];
            const float zz[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int s = 4 * c + q;
                const float d = sqdist3(vec_get<PPT>(px, s) - ox, vec_get<PPT>(py, s) - oy, zz[q] - oz);
                const float d2 = min_f32(d, t[s]);
                t[s] = d2;
                gm[s / GS] = max_f32(gm[s / GS], d2);
            }
            __builtin_amdgcn_sched_barrier(0);   // keep two z chunks live at most: 128-VGPR budget
        }
        const float best = max_f32(max_f32(gm[0], gm[1]), max_f32(gm[2], gm[3]));
        const float wmax = wave_max(best);
        const uint64_t eq = __ballot(best == wmax);
        const int wl = (int)__builtin_ctzll(eq);
        int gsel = 3;
        gsel = gm[2] == wmax ? 2 : gsel;
        gsel = gm[1] == wmax ? 1 : gsel;
        gsel = gm[0] == wmax ? 0 : gsel;
        const int gw = __builtin_amdgcn_readlane(gsel, wl);
        int ls = GS - 1;
        auto find = [&](auto G) {
#pragma unroll
            for (int q = GS - 2; q >= 0; --q) ls = t[decltype(G)::value * GS + q] == wmax ? q : ls;
        };
        switch (gw) {   // wave-uniform
            case 0: find(std::integral_constant<int, 0>{}); break;
            case 1: find(std::integral_constant<int, 1>{}); break;
            case 2: find(std::integral_constant<int, 2>{}); break;
            default: find(std::integral_constant<int, 3>{}); break;
        }
        const int wslot = gw * GS + __builtin_amdgcn_readlane(ls, wl);
        const float cz = zs[(((wslot >> 2) * NT) + (w * 64 + wl)) * 4 + (wslot & 3)];   // wave-uniform address: broadcast
        const float cx = readlane_f(vec_get<PPT>(px, wslot), wl);
        const float cy = readlane_f(vec_get<PPT>(py, wslot), wl);
        const int kw = (w * 64 + wl) * PPT + wslot;
        const int buf = j & 1;
        if (lane == 0) {
            s_cand[buf][w] = make_float4(wmax, cx, cy, cz);
            s_k[buf][w] = kw;
        }
        lds_barrier();
        const int e = lane & 15;
        const float4 c4 = s_cand[buf][e];
        const int kc = s_k[buf][e];
        const float v = e < NW ? c4.x : -2.0f;
        const float vm = row16_max(v);
        const uint64_t eq2 = __ballot(v == vm);
        const int sel = (int)__builtin_ctzll(eq2);
        ox = readlane_f(c4.y, sel);
        oy = readlane_f(c4.z, sel);
        oz = readlane_f(c4.w, sel);
        old = __builtin_amdgcn_readlane(kc, sel);
        if (u == 0) {
            idx[j] = old;
            if (new_xyz) { new_xyz[j * 3 + 0] = ox; new_xyz[j * 3 + 1] = oy; new_xyz[j * 3 + 2] = oz; }
        }
    }
    __syncthreads();
    for (int j = 1 + u; j < m; j += NT) {
        const int p = idx[j];
        const int rb = p / S, sl = p - rb * S;
        idx[j] = bitrev_bits(rb, log2bs) + sl * bs;
    }
    if (temp) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int p = u * PPT + s;
            const int rb = p / S, sl = p - rb * S;
            const int k = bitrev_bits(rb, log2bs) + sl * bs;
            if ((rb < bs) && (k < n)) temp[k] = t[s];
        }
    }
}

// ---- streaming fallback for scenes that do not fit the register file (n > 16384).
// Natural strided ownership (thread tid owns k = tid, tid+1024, ...: coalesced),
// min-distance in the caller's temp buffer, explicit tie key = bitrev10(tid).
__device__ __forceinline__ int row16_min_i(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, DPP_QUAD_XOR2, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_HALF_MIRROR, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, DPP_ROW_MIRROR, 0xF, 0xF, false));
    return v;
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = row16_min_i(v);
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return min(min(a, b), min(c, d));
}

__global__ __launch_bounds__(1024) void fps_stream_kernel(const float *__restrict__ xyz,
                                                          float *__restrict__ temp,
                                                          int32_t *__restrict__ idx,
                                                          float *__restrict__ new_xyz, int n,
                                                          int m) {
    __shared__ float4 s_cand[2][16];
    __shared__ int2 s_kk[2][16];  // {k, key}
    const int b = blockIdx.x;
    xyz += (size_t)b * n * 3;
    temp += (size_t)b * n;
    idx += (size_t)b * m;
    if (new_xyz) new_xyz += (size_t)b * m * 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int key = bitrev_bits(tid, 10);

    int old = 0;
    float ox = xyz[0], oy = xyz[1], oz = xyz[2];
    if (tid == 0) {
        idx[0] = 0;
        if (new_xyz) { new_xyz[0] = ox; new_xyz[1] = oy; new_xyz[2] = oz; }
    }
    for (int j = 1; j < m; ++j) {
        float best = -1.0f;
        int besti = 0;
        for (int k = tid; k < n; k += 1024) {
            const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
            const float d = sqdist3(x - ox, y - oy, z - oz);
            const float d2 = fminf(d, temp[k]);
            temp[k] = d2;
            const bool gt = d2 > best;
            besti = gt ? k : besti;
            best = gt ? d2 : best;
        }
        const float wmax = wave_max(best);
        const int mykey = (best == wmax) ? key : (1 << 20);
        const int wkey = wave_min_i(mykey);
        const uint64_t eq = __ballot(mykey == wkey);
        const int wl = (int)__builtin_ctzll(eq);
        const int kw = __builtin_amdgcn_readlane(besti, wl);
        const int buf = j & 1;
        if (lane == 0) {
            const float cx = xyz[kw * 3 + 0], cy = xyz[kw * 3 + 1], cz = xyz[kw * 3 + 2];
            s_cand[buf][w] = make_float4(wmax, cx, cy, cz);
            s_kk[buf][w] = make_int2(kw, wkey);
        }
        __syncthreads();
        const int e = lane & 15;
        const float4 c = s_cand[buf][e];
        const int2 kk = s_kk[buf][e];
        const float vm = row16_max(c.x);
        const int k2 = (c.x == vm) ? kk.y : (1 << 20);
        const int kmin = row16_min_i(k2);
        const uint64_t eq2 = __ballot(k2 == kmin);
        const int sel = (int)__builtin_ctzll(eq2);
        ox = readlane_f(c.y, sel);
        oy = readlane_f(c.z, sel);
        oz = readlane_f(c.w, sel);
        old = __builtin_amdgcn_readlane(kk.x, sel);
        if (tid == 0) {
            idx[j] = old;
            if (new_xyz) { new_xyz[j * 3 + 0] = ox; new_xyz[j * 3 + 1] = oy; new_xyz[j * 3 + 2] = oz; }
        }
    }
}

// ---- 16384 < n <= 65536 (the dense-scan shapes, SURVEY a1: 65536 -> 16384).  The scene no longer fits the register
// file, but its running min-distance does: SPT <= 64 values per lane of a 1024-thread workgroup stay in VGPRs for the whole
// kernel, so a step only STREAMS xyz (12 B per point, coalesced dwordx3, L2-resident: 786 KB per scene) instead of also
// reading and writing the 4-byte min-distance of every point in global memory as fps_stream_kernel (and the reference,
// sampling_gpu.cu:124-138) do -- 12 instead of 20 bytes per point and step, and no store->load dependence between steps.
// Same strided ownership (k = tid + 1024 s) and tie key as fps_stream_kernel.
template <int SPT>
__global__ __launch_bounds__(512) void fps_big_kernel(const float *__restrict__ xyz, float *__restrict__ temp,
                                                      int32_t *__restrict__ idx, float *__restrict__ new_xyz, int n, int m) {
    // 512 threads x SPT <= 128 slots (256-VGPR budget: 1024 threads would leave 128 and spill).  Thread u owns
    // k = u + 512 s, i.e. the points of the reference's threads u (even s) and u + 512 (odd s); their tie keys are
    // bitrev10(u) and bitrev10(u) + 1, so the two are tracked separately and the even one keeps exact ties.
    typedef float f3v __attribute__((ext_vector_type(3)));
    typedef f3v f3u __attribute__((aligned(4)));
    __shared__ float4 s_cand[2][8];
    __shared__ int2 s_kk[2][8];  // {k, key}
    const int b = blockIdx.x;
    xyz += (size_t)b * n * 3;
    if (temp) temp += (size_t)b * n;
    idx += (size_t)b * m;
    if (new_xyz) new_xyz += (size_t)b * m * 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int key0 = bitrev_bits(tid, 10);
    float t[SPT];
#pragma unroll
    for (int s = 0; s < SPT; ++s) {
        const int k = tid + 512 * s;
        t[s] = k < n ? (temp ? temp[k] : 1e10f) : -1.0f;     // -1: never beats a real candidate, min(d, -1) stays -1
    }
    int old = 0;
    float ox = xyz[0], oy = xyz[1], oz = xyz[2];
    if (tid == 0) {
        idx[0] = 0;
        if (new_xyz) { new_xyz[0] = ox; new_xyz[1] = oy; new_xyz[2] = oz; }
    }
    for (int j = 1; j < m; ++j) {
        float bestA = -1.0f, bestB = -1.0f;
        int iA = 0, iB = 0;
        // 8 independent 12-byte loads in flight per lane and trip (clamped, never branched).  The byte offsets are
        // recomputed every step from an opaque start: hoisted out of the step loop they would cost SPT registers
        unsigned off = (unsigned)tid * 12u;
        asm volatile("" : "+v"(off));
        const unsigned last = (unsigned)(n - 1) * 12u;
        const char *xb = reinterpret_cast<const char *>(xyz);
#pragma unroll
        for (int s0 = 0; s0 < SPT; s0 += 8) {
            f3v p[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { p[u] = *reinterpret_cast<const f3u *>(xb + min(off, last)); off += 512u * 12u; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float d = sqdist3(p[u].x - ox, p[u].y - oy, p[u].z - oz);
                const float d2 = min_f32(d, t[s0 + u]);
                t[s0 + u] = d2;
                if ((u & 1) == 0) {   // reference thread u: ascending k, strict '>' keeps the smallest k among exact ties
                    const bool gt = d2 > bestA;
                    iA = gt ? tid + 512 * (s0 + u) : iA;
                    bestA = gt ? d2 : bestA;
                } else {              // reference thread u + 512
                    const bool gt = d2 > bestB;
                    iB = gt ? tid + 512 * (s0 + u) : iB;
                    bestB = gt ? d2 : bestB;
                }
            }
        }
        const bool useB = bestB > bestA;                 // the lower key (A) keeps exact ties
        const float best = useB ? bestB : bestA;
        const int besti = useB ? iB : iA;
        const int key = key0 + (useB ? 1 : 0);
        const float wmax = wave_max(best);
        const int mykey = (best == wmax) ? key : (1 << 20);
        const int wkey = wave_min_i(mykey);
        const uint64_t eq = __ballot(mykey == wkey);
        const int wl = (int)__builtin_ctzll(eq);
        const int kw = __builtin_amdgcn_readlane(besti, wl);
        const int buf = j & 1;
        if (lane == 0) {
            const float cx = xyz[kw * 3 + 0], cy = xyz[kw * 3 + 1], cz = xyz[kw * 3 + 2];
            s_cand[buf][w] = make_float4(wmax, cx, cy, cz);
            s_kk[buf][w] = make_int2(kw, wkey);
        }
        __syncthreads();
        const int e = lane & 7;
        const float4 c = s_cand[buf][e];
        const int2 kk = s_kk[buf][e];
        const float vm = row16_max(c.x);
        const int k2 = (c.x == vm) ? kk.y : (1 << 20);
        const int kmin = row16_min_i(k2);
        const uint64_t eq2 = __ballot(k2 == kmin);
        const int sel = (int)__builtin_ctzll(eq2);
        ox = readlane_f(c.y, sel);
        oy = readlane_f(c.z, sel);
        oz = readlane_f(c.w, sel);
        old = __builtin_amdgcn_readlane(kk.x, sel);
        if (tid == 0) {
            idx[j] = old;
            if (new_xyz) { new_xyz[j * 3 + 0] = ox; new_xyz[j * 3 + 1] = oy; new_xyz[j * 3 + 2] = oz; }
        }
    }
    if (temp) {
#pragma unroll
        for (int s = 0; s < SPT; ++s) {
            const int k = tid + 512 * s;
            if (k < n) temp[k] = t[s];
        }
    }
}

// host: cuda_utils.h:10-14 (same libm expression as the reference's launcher)
static int opt_n_threads(int work_size) {
    const int pow_2 = (int)(std::log(static_cast<double>(work_size)) / std::log(2.0));
    int v = 1 << pow_2;
    if (v > 1024) v = 1024;
    if (v < 1) v = 1;
    return v;
}

template <int PPT, int NT>
static void launch_reg(int b, int n, int m, const float *xyz, float *temp, int32_t *idx,
                       float *new_xyz, int bs, int log2bs, int S, hipStream_t st) {
    hipLaunchKernelGGL((fps_reg_kernel<PPT, NT>), dim3(b), dim3(NT), 0, st, xyz, temp, idx, new_xyz,
                       n, m, bs, log2bs, S);
}

int fps_bucket_launch(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, float *new_xyz,
                      int bs, int log2bs, int S, hipStream_t st);  // fps_bucket.hip
bool fps_rounds_covers(int m);                                       // fps_bucket.hip
bool fps_v3_launch(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, float *new_xyz, int bs, int log2bs, int S,
                   long R, bool pair, hipStream_t st);             // fps_v3.hip

// WS3D_FPS_PAIR: 0 = never, 1 = always, unset = when the batch exceeds the number of CUs
static bool fps_pair_mode(int b) {
    static const int env = getenv("WS3D_FPS_PAIR") ? atoi(getenv("WS3D_FPS_PAIR")) : -1;
    if (env >= 0) return env != 0;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return b > cus;
}

static int fps_dispatch(int b, int n, int m, const float *xyz, float *temp, int32_t *idx,
                        float *new_xyz, hipStream_t st) {
    if (b < 0 || n <= 0 || m < 0 || !xyz || (!idx && m > 0)) {
        set_error("ws3d_furthest_point_sampling: invalid argument (b=%d n=%d m=%d)", b, n, m);
        return WS3D_E_INVALID;
    }
    if (b == 0 || m == 0) return WS3D_OK;  // sampling_gpu.cu:101
    const int bs = opt_n_threads(n);
    int log2bs = 0;
    while ((1 << log2bs) < bs) ++log2bs;
    const int S = (n + bs - 1) / bs;
    const long R = (long)bs * S;  // number of tie-order positions
    // pruned kernels, fps_bucket.hip (one scene per CU).  With several certified samples per exchange (fps_rounds2_kernel, m <= 6144)
    // they beat the dense sweep from 2049 points up (scripts/ab_fps.py, profiles/r04_fps_ab_small_clouds.txt; us per sample, rounds /
    // dense): 16384 points 0.38 / 1.02 at every batch size (512 scenes: 3.2 vs 6.2 ms), 8192: 0.40 / 0.73 (512 scenes 1.65 vs 2.19 ms),
    // 4096: 0.45 / 0.58 while the scenes do not outnumber the CUs (512 scenes: 0.96 vs 0.79 ms -- several dense workgroups share a CU),
    // 2049: 0.55 / 0.59; at 1025 the dense kernel wins (0.66 / 0.57).  m > 6144 (the samples do not fit in LDS beside the sort
    // tables): the one-sample-per-exchange kernel above 8192 points while the scenes do not outnumber the CUs, else the dense sweep.
    // WS3D_FPS_BUCKET=0 / 1 forces the choice for 2049 .. 16384 points (A/B runs, tests).
    static const int use_bucket = getenv("WS3D_FPS_BUCKET") ? atoi(getenv("WS3D_FPS_BUCKET")) : -1;
    const bool many = fps_pair_mode(b), rounds = fps_rounds_covers(m);
    const bool bucket = use_bucket >= 0 ? use_bucket != 0
                                        : (R > 8192 ? (rounds || !many) : (rounds && (R > 4096 || (R > 2048 && !many))));
    if (bucket && n > 2048 && n <= 16384 && m > 1)
        return fps_bucket_launch(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
    // round-2 kernels (fps_v3.hip: hand-scheduled sweep, winner-only lookup) for 2048 < R <= 16384; the kernels of this file serve the
    // small clouds, the large ones, and every size of the un-contracted distance convention (WS3D_DIST_MODE == 1: fps_v3 declines)
    if (R <= 1024L * 16 && fps_v3_launch(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, R, fps_pair_mode(b), st))
        return check_launch("furthest_point_sampling");
    if (R <= 64L * 16) {
        const int ppt = (int)((R + 63) / 64);
        if (ppt <= 1) launch_reg<1, 64>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
        else if (ppt <= 2) launch_reg<2, 64>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
        else if (ppt <= 4) launch_reg<4, 64>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
        else if (ppt <= 8) launch_reg<8, 64>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
        else launch_reg<16, 64>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
    } else if (R <= 256L * 16) {
        const int ppt = (int)((R + 255) / 256);
        if (ppt <= 8) launch_reg<8, 256>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
        else launch_reg<8, 512>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);   // 512 x 8 beats 256 x 16 by 5 % (measured)
    } else if (R <= 1024L * 16 && R > 512L * 16 && fps_pair_mode(b)) {
        // two scenes per CU: only pays when there are more scenes than CUs
        constexpr size_t lds = (size_t)32 * 512 * sizeof(float);
        if (int rc = raise_lds_cap((const void *)fps_zlds_kernel<32, 512>, lds, "furthest_point_sampling")) return rc;
        hipLaunchKernelGGL((fps_zlds_kernel<32, 512>), dim3(b), dim3(512), lds, st, xyz, temp, idx, new_xyz, n, m, bs, log2bs, S);
    } else if (R <= 1024L * 16) {
        const int ppt = (int)((R + 1023) / 1024);
        // Fewer, fatter waves win: the cross-lane reduction chain (DPP/readlane/ballot) does not
        // overlap between waves of one SIMD, so 8 waves x 32 points/lane (2 waves/SIMD, 231
        // VGPRs) beat 16 waves x 16 points/lane by 18 % per step (measured 1.17 vs 1.38 us).
        if (ppt <= 8) launch_reg<16, 512>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
        else launch_reg<32, 512>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st);
    } else if (n <= 65536 && bs == 1024) {
        // min-distance in registers, xyz streamed from L2
        if (n <= 32768) hipLaunchKernelGGL(fps_big_kernel<64>, dim3(b), dim3(512), 0, st, xyz, temp, idx, new_xyz, n, m);
        else hipLaunchKernelGGL(fps_big_kernel<128>, dim3(b), dim3(512), 0, st, xyz, temp, idx, new_xyz, n, m);
    } else {
        if (!temp) {
            set_error("ws3d_furthest_point_sampling: n=%d exceeds the on-chip capacity (65536); "
                      "the streaming path needs the caller's temp (b,n) buffer", n);
            return WS3D_E_WORKSPACE;
        }
        hipLaunchKernelGGL(fps_stream_kernel, dim3(b), dim3(1024), 0, st, xyz, temp, idx, new_xyz, n, m);
    }
    return check_launch("furthest_point_sampling");
}

}  // namespace ws3d

extern "C" int ws3d_furthest_point_sampling(int b, int n, int m, const float *xyz, float *temp,
                                            int32_t *idx, ws3d_stream_t stream) {
    return ws3d::fps_dispatch(b, n, m, xyz, temp, idx, nullptr, ws3d::as_stream(stream));
}

extern "C" int ws3d_furthest_point_sampling_gather(int b, int n, int m, const float *xyz,
                                                   float *temp, int32_t *idx, float *new_xyz,
                                                   ws3d_stream_t stream) {
    return ws3d::fps_dispatch(b, n, m, xyz, temp, idx, new_xyz, ws3d::as_stream(stream));
}
