// fps_v3.hip -- furthest point sampling, round-2 kernels (gfx950).  Same contract and the same
// position order as fps.hip (thread u owns tie-order positions [u*PPT, (u+1)*PPT), see the header
// of fps.hip), different step:
//
//   sweep   hand-scheduled: one asm block per 4 points, the four distance chains interleaved,
//           7 VALU per point (3 v_subrev against the SGPR-held sample, v_mul, 2 v_fmac, v_min) plus one
//           v_max3 per TWO points into 4 per-lane group maxima -- 7.5 issue slots per point instead of
//           8 (group maxima) or 10-13 (running value + slot with v_cmp / 2 v_cndmask).  No wait
//           states: the compiler pads every dependent pair of single-instruction asm statements with
//           s_nop (it must assume a dst_sel forwarding hazard inside unknown asm); one block = one
//           statement, so nothing is padded.
//   reduce  every wave: 2 v_max3/v_max, wave maximum by 6 DPP stages, ONE dword to LDS, barrier; every
//           wave reads the <= 16 wave maxima, reduces them with 3-4 DPP stages and knows the winning
//           WAVE (lowest wave holding the maximum).
//   lookup  ONLY the winning wave finds the winner: ballot -> lane, 4 readlanes -> group, one block of
//           independent v_cmp / v_cndmask + a v_min3 tree -> slot, VGPR-indexed moves + readlanes ->
//           coordinates; it writes the index / gathered coordinates and publishes {x, y, z} in LDS;
//           second barrier; every wave picks the sample up with one broadcast ds_read_b128.
//
// fps.hip's step made every wave do the lookup (group select, switch, slot search, 2 VGPR-indexed moves,
// 7 readlanes, a 20-byte record) before its single barrier: 437 issued instructions per wave and step at
// 32 points per lane, 256 of them sweep.  Here a non-winning wave issues ~290 (248 sweep), which is what
// counts when two scenes share a CU and the SIMDs are issue-bound; the exposed chain (one scene per CU)
// is about as long as before: one more LDS exchange, a much shorter lookup.
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef WS3D_FPS3_NG8
#define WS3D_FPS3_NG8 0      // A/B: 1 = one group per 4-point chunk in the one-scene-per-CU kernels (8 groups at 32 points per lane)
#endif
#ifndef WS3D_FPS3_VSAMPLE
#define WS3D_FPS3_VSAMPLE 1  // A/B: sample coordinates as VGPR operands of the sweep (one-scene-per-CU kernels) instead of SGPRs
#endif
#ifndef WS3D_FPS3_ZSG
#define WS3D_FPS3_ZSG 0      // A/B: 1 = SGPR sample operands in the two-scenes-per-CU kernel (3 VGPRs less)
#endif
#ifdef WS3D_FPS_PROF
// per-segment clock accumulators (scripts/ubench/fps3_prof.hip): [block slot 0..3][wave 0..15][segment 0..7]
__device__ long long g_fps3_prof[4 * 16 * 8];
#define V3PROF_DECL long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long prof_last = __builtin_readcyclecounter();
#define V3PROF(i) { const long long now_ = __builtin_readcyclecounter(); prof_acc[i] += now_ - prof_last; prof_last = now_; }
#define V3PROF_STORE { const int pb_ = (blockIdx.x == 0) ? 0 : (blockIdx.x == 1) ? 1 : (blockIdx.x == gridDim.x / 2) ? 2 : (blockIdx.x == gridDim.x - 1) ? 3 : -1; \
    if (lane == 0 && pb_ >= 0) for (int i_ = 0; i_ < 8; ++i_) g_fps3_prof[(pb_ * 16 + w) * 8 + i_] = prof_acc[i_]; }
#else
#define V3PROF_DECL
#define V3PROF(i)
#define V3PROF_STORE
#endif

namespace ws3d {

template <int N> struct V3Vec { typedef float type __attribute__((ext_vector_type(N))); };
template <> struct V3Vec<1> { typedef float type; };
template <int N> using v3vec = typename V3Vec<N>::type;
template <int N> __device__ __forceinline__ float v3get(const v3vec<N> &v, int i) { return v[i]; }
template <> __device__ __forceinline__ float v3get<1>(const v3vec<1> &v, int) { return v; }
template <int N> __device__ __forceinline__ void v3set(v3vec<N> &v, int i, float x) { v[i] = x; }
template <> __device__ __forceinline__ void v3set<1>(v3vec<1> &v, int, float x) { v = x; }

__device__ __forceinline__ int v3_bitrev(int v, int bits) {
    return bits == 0 ? 0 : (int)(__builtin_bitreverse32((uint32_t)v) >> (32 - bits));
}

// d = fma(dz,dz, fma(dx,dx, dy*dy)) with d* = p* - o*  (== sqdist3 of common.h, bit for bit);
// t = min(d, t); g = max3(g, t_a, t_b).  ox/oy/oz are wave-uniform (SGPRs).
#define WS3D_SWEEP4_ASM                                  \
    "v_subrev_f32 %[d0], %[oy], %[y0]\n\t"               \
    "v_subrev_f32 %[d1], %[oy], %[y1]\n\t"               \
    "v_subrev_f32 %[e0], %[ox], %[x0]\n\t"               \
    "v_subrev_f32 %[e1], %[ox], %[x1]\n\t"               \
    "v_mul_f32 %[d0], %[d0], %[d0]\n\t"                  \
    "v_mul_f32 %[d1], %[d1], %[d1]\n\t"                  \
    "v_fmac_f32 %[d0], %[e0], %[e0]\n\t"                 \
    "v_fmac_f32 %[d1], %[e1], %[e1]\n\t"                 \
    "v_subrev_f32 %[e0], %[oz], %[z0]\n\t"               \
    "v_subrev_f32 %[e1], %[oz], %[z1]\n\t"               \
    "v_fmac_f32 %[d0], %[e0], %[e0]\n\t"                 \
    "v_fmac_f32 %[d1], %[e1], %[e1]\n\t"                 \
    "v_subrev_f32 %[e0], %[ox], %[x2]\n\t"               \
    "v_subrev_f32 %[e1], %[ox], %[x3]\n\t"               \
    "v_min_f32 %[t0], %[d0], %[t0]\n\t"                  \
    "v_min_f32 %[t1], %[d1], %[t1]\n\t"                  \
    "v_subrev_f32 %[d0], %[oy], %[y2]\n\t"               \
    "v_subrev_f32 %[d1], %[oy], %[y3]\n\t"               \
    "v_max3_f32 %[g], %[g], %[t0], %[t1]\n\t"            \
    "v_mul_f32 %[d0], %[d0], %[d0]\n\t"                  \
    "v_mul_f32 %[d1], %[d1], %[d1]\n\t"                  \
    "v_fmac_f32 %[d0], %[e0], %[e0]\n\t"                 \
    "v_fmac_f32 %[d1], %[e1], %[e1]\n\t"                 \
    "v_subrev_f32 %[e0], %[oz], %[z2]\n\t"               \
    "v_subrev_f32 %[e1], %[oz], %[z3]\n\t"               \
    "v_fmac_f32 %[d0], %[e0], %[e0]\n\t"                 \
    "v_fmac_f32 %[d1], %[e1], %[e1]\n\t"                 \
    "v_min_f32 %[t2], %[d0], %[t2]\n\t"                  \
    "v_min_f32 %[t3], %[d1], %[t3]\n\t"                  \
    "v_max3_f32 %[g], %[g], %[t2], %[t3]"

template <bool SG>   // SG: the sample is held in SGPRs (two-scenes-per-CU kernel: VGPR budget), else in VGPRs (no readlanes)
__device__ __forceinline__ void sweep4(float x0, float x1, float x2, float x3, float y0, float y1, float y2, float y3,
                                       float z0, float z1, float z2, float z3, float &t0, float &t1, float &t2, float &t3,
                                       float &g, float ox, float oy, float oz) {
    float d0, d1, e0, e1;     // two points in flight are enough: a wave issues one instruction per ~4.2 clk anyway
    if constexpr (SG) {
        asm(WS3D_SWEEP4_ASM
            : [d0] "=&v"(d0), [d1] "=&v"(d1), [e0] "=&v"(e0), [e1] "=&v"(e1), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [g] "+v"(g)
            : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3),
              [z0] "v"(z0), [z1] "v"(z1), [z2] "v"(z2), [z3] "v"(z3), [ox] "s"(ox), [oy] "s"(oy), [oz] "s"(oz));
    } else {
        asm(WS3D_SWEEP4_ASM
            : [d0] "=&v"(d0), [d1] "=&v"(d1), [e0] "=&v"(e0), [e1] "=&v"(e1), [t0] "+v"(t0), [t1] "+v"(t1), [t2] "+v"(t2), [t3] "+v"(t3), [g] "+v"(g)
            : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3),
              [z0] "v"(z0), [z1] "v"(z1), [z2] "v"(z2), [z3] "v"(z3), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz));
    }
}

// Winner lookup by SELECTION (winning wave only): for the 4 slots QB..QB+3, highest first so that the lowest
// matching slot is the last writer:  if (t_q == vm) { k = q; cx = x_q; cy = y_q; cz = z_q; }.
// Four compares into VCC / three SGPR pairs, then 4 independent select chains -- every mask is consumed
// >= 3 instructions after the v_cmp that wrote it (gfx950: 2 wait states VALU-writes-SGPR -> VALU-reads-it).
// Replaces the VGPR-indexed moves (s_set_gpr_idx + v_mov + v_readlane: ~235 clk each, measured) of fps.hip.
template <int QB>
__device__ __forceinline__ void pick4(float t0, float t1, float t2, float t3, float x0, float x1, float x2, float x3,
                                      float y0, float y1, float y2, float y3, float z0, float z1, float z2, float z3,
                                      float vm, int &k, float &cx, float &cy, float &cz) {
    unsigned long long m0, m1, m2;
    asm("v_cmp_eq_f32 vcc, %[vm], %[t3]\n\t"
        "v_cmp_eq_f32 %[m2], %[vm], %[t2]\n\t"
        "v_cmp_eq_f32 %[m1], %[vm], %[t1]\n\t"
        "v_cmp_eq_f32 %[m0], %[vm], %[t0]\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+3, vcc\n\t"
        "v_cndmask_b32 %[cx], %[cx], %[x3], vcc\n\t"
        "v_cndmask_b32 %[cy], %[cy], %[y3], vcc\n\t"
        "v_cndmask_b32 %[cz], %[cz], %[z3], vcc\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+2, %[m2]\n\t"
        "v_cndmask_b32_e64 %[cx], %[cx], %[x2], %[m2]\n\t"
        "v_cndmask_b32_e64 %[cy], %[cy], %[y2], %[m2]\n\t"
        "v_cndmask_b32_e64 %[cz], %[cz], %[z2], %[m2]\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+1, %[m1]\n\t"
        "v_cndmask_b32_e64 %[cx], %[cx], %[x1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[cy], %[cy], %[y1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[cz], %[cz], %[z1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+0, %[m0]\n\t"
        "v_cndmask_b32_e64 %[cx], %[cx], %[x0], %[m0]\n\t"
        "v_cndmask_b32_e64 %[cy], %[cy], %[y0], %[m0]\n\t"
        "v_cndmask_b32_e64 %[cz], %[cz], %[z0], %[m0]"
        : [k] "+v"(k), [cx] "+v"(cx), [cy] "+v"(cy), [cz] "+v"(cz), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3),
          [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3), [z0] "v"(z0), [z1] "v"(z1), [z2] "v"(z2), [z3] "v"(z3),
          [vm] "s"(vm), [qb] "n"(QB)
        : "vcc");
}

// the same without z (two-scenes-per-CU kernel: z is in LDS)
template <int QB>
__device__ __forceinline__ void pick4xy(float t0, float t1, float t2, float t3, float x0, float x1, float x2, float x3,
                                        float y0, float y1, float y2, float y3, float vm, int &k, float &cx, float &cy) {
    unsigned long long m0, m1, m2;
    asm("v_cmp_eq_f32 vcc, %[vm], %[t3]\n\t"
        "v_cmp_eq_f32 %[m2], %[vm], %[t2]\n\t"
        "v_cmp_eq_f32 %[m1], %[vm], %[t1]\n\t"
        "v_cmp_eq_f32 %[m0], %[vm], %[t0]\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+3, vcc\n\t"
        "v_cndmask_b32 %[cx], %[cx], %[x3], vcc\n\t"
        "v_cndmask_b32 %[cy], %[cy], %[y3], vcc\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+2, %[m2]\n\t"
        "v_cndmask_b32_e64 %[cx], %[cx], %[x2], %[m2]\n\t"
        "v_cndmask_b32_e64 %[cy], %[cy], %[y2], %[m2]\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+1, %[m1]\n\t"
        "v_cndmask_b32_e64 %[cx], %[cx], %[x1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[cy], %[cy], %[y1], %[m1]\n\t"
        "v_cndmask_b32_e64 %[k], %[k], %[qb]+0, %[m0]\n\t"
        "v_cndmask_b32_e64 %[cx], %[cx], %[x0], %[m0]\n\t"
        "v_cndmask_b32_e64 %[cy], %[cy], %[y0], %[m0]"
        : [k] "+v"(k), [cx] "+v"(cx), [cy] "+v"(cy), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [t0] "v"(t0), [t1] "v"(t1), [t2] "v"(t2), [t3] "v"(t3), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3),
          [y0] "v"(y0), [y1] "v"(y1), [y2] "v"(y2), [y3] "v"(y3), [vm] "s"(vm), [qb] "n"(QB)
        : "vcc");
}

// the PPT < 4 shapes (tiny clouds, one wave): plain per-point form
__device__ __forceinline__ void sweep1(float x, float y, float z, float &t, float &g, float ox, float oy, float oz) {
    const float d = sqdist3(x - ox, y - oy, z - oz);
    t = min_f32(d, t);
    g = max_f32(g, t);
}

__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max over aligned groups of W lanes (the wave maxima of the workgroup), replicated inside the group; ONE asm
// statement, so the compiler adds no s_nop of its own between the stages (s_nop 1 = the 2 wait states VALU write ->
// DPP read)
template <int W>
__device__ __forceinline__ float lanes_max(float v) {
    float r;
    if constexpr (W <= 2)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v));
    else if constexpr (W <= 4)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v));
    else if constexpr (W <= 8)
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v));
    else
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
            "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "=&v"(r) : "v"(v));
    return r;
}

// wave maximum -> SGPR: 4 row stages, row_bcast:15 / row_bcast:31 fold the four rows, lane 63 holds the result
__device__ __forceinline__ float wave_max_row(float v) {
    float r;
    asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "=&v"(r) : "v"(v));
    return r;     // lane 63 holds the wave maximum
}

// Wave maximum AND the per-lane group index in one statement: the six DPP stages need 2 wait states each between the
// VALU write and the DPP read of the same register; instead of s_nop those slots carry the compares / selects that find
// the lane's LOWEST group holding its own maximum (gsel) -- free: a lone wave issues one instruction per ~4.2 clk anyway.
// r: lane 63 holds the wave maximum.
#define WS3D_DPP(ctrl) " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_max_gsel8(float best, const float (&gm)[8], float &r, int &gsel) {
    unsigned long long m0, m1, m2, m3, m4, m5, m6;
    asm("v_cmp_eq_f32 %[m6], %[g6], %[b]\n\t"
        "v_cmp_eq_f32 %[m5], %[g5], %[b]\n\t"
        "v_max_f32_dpp %[r], %[b], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], 7, 6, %[m6]\n\t"
        "v_cmp_eq_f32 %[m4], %[g4], %[b]\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 5, %[m5]\n\t"
        "v_cmp_eq_f32 %[m3], %[g3], %[b]\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 4, %[m4]\n\t"
        "v_cmp_eq_f32 %[m2], %[g2], %[b]\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 3, %[m3]\n\t"
        "v_cmp_eq_f32 %[m1], %[g1], %[b]\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 2, %[m2]\n\t"
        "v_cmp_eq_f32 %[m0], %[g0], %[b]\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 1, %[m1]\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 0, %[m0]"
        : [r] "=&v"(r), [s] "=&v"(gsel), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4),
          [m5] "=&s"(m5), [m6] "=&s"(m6)
        : [b] "v"(best), [g0] "v"(gm[0]), [g1] "v"(gm[1]), [g2] "v"(gm[2]), [g3] "v"(gm[3]), [g4] "v"(gm[4]), [g5] "v"(gm[5]),
          [g6] "v"(gm[6]));
}
__device__ __forceinline__ void wave_max_gsel4(float best, const float (&gm)[4], float &r, int &gsel) {
    unsigned long long m0, m1, m2;
    asm("v_cmp_eq_f32 %[m2], %[g2], %[b]\n\t"
        "v_cmp_eq_f32 %[m1], %[g1], %[b]\n\t"
        "v_max_f32_dpp %[r], %[b], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], 3, 2, %[m2]\n\t"
        "v_cmp_eq_f32 %[m0], %[g0], %[b]\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 1, %[m1]\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], %[s], 0, %[m0]\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_bcast:31 row_mask:0xc bank_mask:0xf"
        : [r] "=&v"(r), [s] "=&v"(gsel), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2)
        : [b] "v"(best), [g0] "v"(gm[0]), [g1] "v"(gm[1]), [g2] "v"(gm[2]));
}
__device__ __forceinline__ void wave_max_gsel2(float best, const float (&gm)[2], float &r, int &gsel) {
    unsigned long long m0;
    asm("v_cmp_eq_f32 %[m0], %[g0], %[b]\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %[r], %[b], %[b] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_cndmask_b32_e64 %[s], 1, 0, %[m0]\n\t"
        "s_nop 0\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %[r], %[r], %[r] row_bcast:31 row_mask:0xc bank_mask:0xf"
        : [r] "=&v"(r), [s] "=&v"(gsel), [m0] "=&s"(m0)
        : [b] "v"(best), [g0] "v"(gm[0]));
}

// PPT points per lane, NT threads.
//   ZLDS: z lives in LDS (x, y, min-dist in VGPRs, <= 128 VGPRs): two workgroups = two scenes per CU (throughput mode).
//   TWO exchanges per step -- wave maxima, then the winner -- so that only the winning wave does the lookup.
//   ONEX (A/B only, WS3D_FPS_ONEX=1): ONE exchange per step, every wave looks its own candidate up and publishes {max, x, y,
//   z, position}.  Measured SLOWER (1.12 vs 1.08 us/step at 512 x 32, 1.37 vs 1.03 at 1024 x 16): the lookup is scalar-heavy
//   (readlanes, mask SGPRs, branches) and the CU has ONE scalar unit, so 8-16 lookups at once serialise; one lookup by the
//   winning wave plus a second LDS exchange is cheaper.
//   DUO (with ZLDS): TWO scenes per 2 NT-thread workgroup (waves 0-7 / 8-15), half a step out of phase.  Two independent
//   workgroups on one CU lock IN phase -- both sweep at half rate, then both sit through their reduction chains with the
//   SIMDs idle; priorities and start-up offsets do not break that (measured).  Here both halves run the SAME loop, the
//   second half merely passes ONE extra workgroup barrier before it: from then on every barrier pairs one scene's
//   "wave maxima are in LDS" with the other scene's "winner is in LDS", so while one scene sweeps the other one reduces
//   and looks its winner up, and a scene-step costs about one sweep instead of a sweep plus a chain.
template <int PPT, int NT, bool ZLDS, bool ONEX = false, bool DUO = false>
__global__ __launch_bounds__(DUO ? 2 * NT : NT) __attribute__((amdgpu_waves_per_eu(ZLDS ? 4 : 1, ZLDS ? 4 : 8))) void fps_v3_kernel(
    const float *__restrict__ xyz, float *__restrict__ temp, int32_t *__restrict__ idx, float *__restrict__ new_xyz, int n, int m,
    int bs, int log2bs, int S, int prio_mode) {
    constexpr int NW = NT / WS3D_WAVE;
    // groups per lane: one maximum per 4-point chunk (latency kernel: the lookup touches 4 slots), one per 8 points in the
    // two-scenes-per-CU kernel (VGPR budget; only the winning wave pays for the longer lookup)
    constexpr int NC = PPT >= 4 ? PPT / 4 : 1;                               // 4-point chunks
    constexpr int NG = ((ZLDS || !WS3D_FPS3_NG8) && PPT >= 16) ? PPT / 8 : NC;
    constexpr int GS = PPT / NG;                                              // 4 or 8 (or PPT for the 1- / 2-point shapes)
    // two-scenes-per-CU kernel: the y of the LAST 4-point chunk lives in LDS beside z (4 VGPRs: the 128-register budget is
    // otherwise one or two registers short and the allocator spills a coordinate that the sweep reloads every step)
    constexpr bool YLDS = ZLDS && PPT == 32;
    static_assert(NG <= 8, "at most 32 points per lane");
    static_assert(!DUO || (ZLDS && !ONEX), "DUO is a variant of the two-exchange ZLDS kernel");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];   // ZLDS: [DUO: 2 scenes][NC (+1: YLDS)][NT] float4
    __shared__ float s_wmax_all[DUO ? 2 : 1][16];
    __shared__ float4 s_rec_all[DUO ? 2 : 1][2][16];  // ONEX: {wave max, x, y, z} per wave, double-buffered by step parity (ONE
    __shared__ int s_pos[2][16];                      //       barrier per step);  two-exchange kernel: [0][0] = the winner's {x, y, z}
    __shared__ int s_rank;

    // DUO: grp (which scene of the workgroup) is wave-uniform -- say so, or every per-scene pointer lives in VGPRs
    const int grp = DUO ? __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT) : 0;
    const int b = DUO ? 2 * (int)blockIdx.x + grp : (int)blockIdx.x;
    xyz += (size_t)b * n * 3;
    idx += (size_t)b * m;
    if (temp) temp += (size_t)b * n;
    if (new_xyz) new_xyz += (size_t)b * m * 3;
    const int u = DUO ? (int)threadIdx.x - grp * NT : (int)threadIdx.x, lane = u & 63, w = u >> 6;
    char *smem_z = smem_all + (size_t)grp * (NC + (PPT == 32 ? 1 : 0)) * NT * sizeof(float4);
    float *s_wmax = s_wmax_all[grp];
    float4 (*s_rec)[16] = s_rec_all[grp];

    v3vec<PPT> px, py;
    v3vec<ZLDS ? 1 : PPT> pz;
    float t[PPT];
    float4 *zs4 = reinterpret_cast<float4 *>(smem_z);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float z4[4] = {0.f, 0.f, 0.f, 0.f}, y4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < (PPT >= 4 ? 4 : PPT); ++q) {
            const int s = 4 * c + q;
            const int p = u * PPT + s;
            const int rb = p / S, sl = p - rb * S;
            const int k = v3_bitrev(rb, log2bs) + sl * bs;
            const bool valid = (rb < bs) && (k < n);
            v3set<PPT>(px, s, valid ? xyz[k * 3 + 0] : 0.f);
            const float yv = valid ? xyz[k * 3 + 1] : 0.f;
            if (YLDS && c == NC - 1) y4[q] = yv; else v3set<PPT>(py, s, yv);
            const float z = valid ? xyz[k * 3 + 2] : 0.f;
            if constexpr (ZLDS) z4[q] = z; else v3set<PPT>(pz, s, z);
            t[s] = valid ? (temp ? temp[k] : 1e10f) : -1.0f;   // -1 never beats a real candidate (>= 0)
        }
        if constexpr (ZLDS) zs4[c * NT + u] = make_float4(z4[0], z4[1], z4[2], z4[3]);
        if (YLDS && c == NC - 1) zs4[NC * NT + u] = make_float4(y4[0], y4[1], y4[2], y4[3]);
    }
    float ox = xyz[0], oy = xyz[1], oz = xyz[2];
    if (u == 0) {
        idx[0] = 0;
        if (new_xyz) { new_xyz[0] = ox; new_xyz[1] = oy; new_xyz[2] = oz; }
    }
    // Two workgroups share the CU (ZLDS).  Issue priority: the reduction chain of either scene outranks a sweep (few
    // instructions, all latency); the scene holding the low wave slots of the CU outranks the other while both sweep.
    int sweep_prio = -1;
    if constexpr (DUO) sweep_prio = prio_mode != 0 ? 0 : -1;      // chain priority only: the barriers fix the phase
    if constexpr (ZLDS && !DUO) {
        if (u == 0) s_rank = 1 << 20;
        __syncthreads();
        // HW_REG_HW_ID (id 4) bits [3:0] = wave slot on its SIMD: the first workgroup of the CU holds the low slots
        const int slot = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
        if (lane == 0) atomicMin(&s_rank, slot);
        __syncthreads();
        if (prio_mode != 0) sweep_prio = (s_rank == 0 && (prio_mode & 3) == 1) ? 1 : 0;      // mode 2: chain priority only (A/B runs)
        // start-up offset of the second scene of the CU (prio_mode >> 2, units of 64 clk): A/B runs of the anti-phase start
        if (s_rank != 0) for (int d = prio_mode >> 2; d > 0; --d) __builtin_amdgcn_s_sleep(1);
    }
    const float *zs = reinterpret_cast<const float *>(smem_z);
    if constexpr (DUO) {
        __syncthreads();
        if (grp == 1) lds_barrier();      // the second scene runs one barrier behind (paired with the first scene's barrier 1)
    }
    // two-exchange kernel: the winning wave's stores of step j are issued at the top of step j+1 (off the chain)
    // (its coordinates are the current sample ox, oy, oz by then)
    bool pend = false;
    int pend_pos = 0;

    V3PROF_DECL
    for (int j = 1; j < m; ++j) {
        V3PROF(6)
        if constexpr (ZLDS) { if (sweep_prio == 1) __builtin_amdgcn_s_setprio(1); else if (sweep_prio == 0) __builtin_amdgcn_s_setprio(0); }
        if constexpr (!ONEX && NW > 1) {
            if (pend) {
                if (lane == 0) {
                    idx[j - 1] = pend_pos;
                    if (new_xyz) { new_xyz[(j - 1) * 3 + 0] = ox; new_xyz[(j - 1) * 3 + 1] = oy; new_xyz[(j - 1) * 3 + 2] = oz; }
                }
                pend = false;
            }
        }
        // wave-uniform sample as VGPR operands of the sweep
        constexpr bool SG = !WS3D_FPS3_VSAMPLE || (ZLDS && WS3D_FPS3_ZSG);   // measured: SGPR operands cost 15-25 % of the step (an SGPR source halves the issue rate)
        float sx = ox, sy = oy, sz = oz;
        if constexpr (SG) { sx = readlane_f(ox, 0); sy = readlane_f(oy, 0); sz = readlane_f(oz, 0); }
        float gm[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) gm[g] = -1.0f;
        if constexpr (PPT >= 4) {
            float4 zn = make_float4(0.f, 0.f, 0.f, 0.f), yl = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (ZLDS) zn = zs4[u];   // software pipeline: chunk c+1 is in flight while c is consumed
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                float z0, z1, z2, z3;
                if constexpr (ZLDS) {
                    z0 = zn.x; z1 = zn.y; z2 = zn.z; z3 = zn.w;
                    if (c + 1 < NC) zn = zs4[(c + 1) * NT + u];
                } else {
                    z0 = v3get<ZLDS ? 1 : PPT>(pz, 4 * c); z1 = v3get<ZLDS ? 1 : PPT>(pz, 4 * c + 1);
                    z2 = v3get<ZLDS ? 1 : PPT>(pz, 4 * c + 2); z3 = v3get<ZLDS ? 1 : PPT>(pz, 4 * c + 3);
                }
                float y0, y1, y2, y3;
                if (YLDS && c == NC - 1) { y0 = yl.x; y1 = yl.y; y2 = yl.z; y3 = yl.w; }
                else { y0 = v3get<PPT>(py, 4 * c); y1 = v3get<PPT>(py, 4 * c + 1); y2 = v3get<PPT>(py, 4 * c + 2); y3 = v3get<PPT>(py, 4 * c + 3); }
                if (YLDS && c == NC - 2) yl = zs4[NC * NT + u];       // in flight while chunk NC-2 is consumed
#if WS3D_DIST_MODE == 2
                // the asm computes (A - a)^2, then fma with (B - b), then fma with (z - oz); mode 0: A = y, B = x; mode 2: A = x, B = y
                sweep4<SG>(y0, y1, y2, y3,
                             v3get<PPT>(px, 4 * c), v3get<PPT>(px, 4 * c + 1), v3get<PPT>(px, 4 * c + 2), v3get<PPT>(px, 4 * c + 3),
                             z0, z1, z2, z3, t[4 * c], t[4 * c + 1], t[4 * c + 2], t[4 * c + 3], gm[(4 * c) / GS], sy, sx, sz);
#else
                sweep4<SG>(v3get<PPT>(px, 4 * c), v3get<PPT>(px, 4 * c + 1), v3get<PPT>(px, 4 * c + 2), v3get<PPT>(px, 4 * c + 3),
                             y0, y1, y2, y3,
                             z0, z1, z2, z3, t[4 * c], t[4 * c + 1], t[4 * c + 2], t[4 * c + 3], gm[(4 * c) / GS], sx, sy, sz);
#endif
                if constexpr (ZLDS) __builtin_amdgcn_sched_barrier(0);   // keep two z chunks live at most (128-VGPR budget)
            }
        } else {
#pragma unroll
            for (int s = 0; s < PPT; ++s)
                sweep1(v3get<PPT>(px, s), v3get<PPT>(py, s), v3get<ZLDS ? 1 : PPT>(pz, s), t[s], gm[0], sx, sy, sz);
        }
        V3PROF(0)
        if constexpr (ZLDS) { if (sweep_prio >= 0) __builtin_amdgcn_s_setprio(3); }
        float best, rmax;
        int gsel = 0;
        if constexpr (NG == 8) {
            best = max3_f32(max3_f32(gm[0], gm[1], gm[2]), max3_f32(gm[3], gm[4], gm[5]), max_f32(gm[6], gm[7]));
            wave_max_gsel8(best, gm, rmax, gsel);
        } else if constexpr (NG == 4) {
            best = max_f32(max3_f32(gm[0], gm[1], gm[2]), gm[3]);
            wave_max_gsel4(best, gm, rmax, gsel);
        } else if constexpr (NG == 2) {
            best = max_f32(gm[0], gm[1]);
            wave_max_gsel2(best, gm, rmax, gsel);
        } else {
            best = gm[0];
            rmax = wave_max_row(best);
        }
        const float wmax = readlane_f(rmax, 63);      // wave-uniform (SGPR)

        // the wave's (or, two-exchange kernel: the workgroup's) winner: lane, group, slot, coordinates
        auto lookup = [&](float vm, int &pos, float &cx, float &cy, float &cz) {
            const uint64_t eq = __ballot(best == vm);
            const int wl = (int)__builtin_ctzll(eq);              // lowest lane among ties
            int wslot;
            if constexpr (GS >= 4) {
                const int gw = NG > 1 ? __builtin_amdgcn_readlane(gsel, wl) : 0;    // the winning lane's lowest group holding vm
                int k = 0;
                float vx = 0.f, vy = 0.f, vz = 0.f;
                auto pick1 = [&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    if constexpr (YLDS && q == PPT - 4) {
                        const float4 yq = zs4[NC * NT + u];
                        pick4xy<q>(t[q], t[q + 1], t[q + 2], t[q + 3], v3get<PPT>(px, q), v3get<PPT>(px, q + 1), v3get<PPT>(px, q + 2),
                                   v3get<PPT>(px, q + 3), yq.x, yq.y, yq.z, yq.w, vm, k, vx, vy);
                    } else if constexpr (ZLDS)
                        pick4xy<q>(t[q], t[q + 1], t[q + 2], t[q + 3], v3get<PPT>(px, q), v3get<PPT>(px, q + 1), v3get<PPT>(px, q + 2),
                                   v3get<PPT>(px, q + 3), v3get<PPT>(py, q), v3get<PPT>(py, q + 1), v3get<PPT>(py, q + 2),
                                   v3get<PPT>(py, q + 3), vm, k, vx, vy);
                    else
                        pick4<q>(t[q], t[q + 1], t[q + 2], t[q + 3], v3get<PPT>(px, q), v3get<PPT>(px, q + 1), v3get<PPT>(px, q + 2),
                                 v3get<PPT>(px, q + 3), v3get<PPT>(py, q), v3get<PPT>(py, q + 1), v3get<PPT>(py, q + 2),
                                 v3get<PPT>(py, q + 3), v3get<ZLDS ? 1 : PPT>(pz, q), v3get<ZLDS ? 1 : PPT>(pz, q + 1),
                                 v3get<ZLDS ? 1 : PPT>(pz, q + 2), v3get<ZLDS ? 1 : PPT>(pz, q + 3), vm, k, vx, vy, vz);
                };
                auto pick = [&](auto G) {
                    if constexpr (GS == 8) pick1(std::integral_constant<int, GS * decltype(G)::value + 4>{});   // high half first
                    pick1(std::integral_constant<int, GS * decltype(G)::value>{});
                };
                if constexpr (NG == 1) pick(std::integral_constant<int, 0>{});
                else if constexpr (NG == 2) {
                    if (gw == 0) pick(std::integral_constant<int, 0>{}); else pick(std::integral_constant<int, 1>{});
                } else if constexpr (NG == 4) {
                    switch (gw) {
                        case 0: pick(std::integral_constant<int, 0>{}); break;
                        case 1: pick(std::integral_constant<int, 1>{}); break;
                        case 2: pick(std::integral_constant<int, 2>{}); break;
                        default: pick(std::integral_constant<int, 3>{}); break;
                    }
                } else {
                    switch (gw) {
                        case 0: pick(std::integral_constant<int, 0>{}); break;
                        case 1: pick(std::integral_constant<int, 1>{}); break;
                        case 2: pick(std::integral_constant<int, 2>{}); break;
                        case 3: pick(std::integral_constant<int, 3>{}); break;
                        case 4: pick(std::integral_constant<int, 4>{}); break;
                        case 5: pick(std::integral_constant<int, 5>{}); break;
                        case 6: pick(std::integral_constant<int, 6>{}); break;
                        default: pick(std::integral_constant<int, NG == 8 ? 7 : 0>{}); break;
                    }
                }
                wslot = __builtin_amdgcn_readlane(k, wl);
                cx = readlane_f(vx, wl);
                cy = readlane_f(vy, wl);
                if constexpr (ZLDS) cz = zs[(((wslot >> 2) * NT) + (w * 64 + wl)) * 4 + (wslot & 3)];   // wave-uniform address
                else cz = readlane_f(vz, wl);
            } else {   // 1 or 2 points per lane (tiny clouds, one wave)
                wslot = 0;
                if constexpr (GS == 2) wslot = __builtin_amdgcn_readlane((t[0] == vm) ? 0 : 1, wl);
                cx = readlane_f(v3get<PPT>(px, wslot), wl);
                cy = readlane_f(v3get<PPT>(py, wslot), wl);
                cz = readlane_f(v3get<ZLDS ? 1 : PPT>(pz, wslot), wl);
            }
            pos = (w * 64 + wl) * PPT + wslot;   // tie-order POSITION; -> point index after the loop
        };

        if constexpr (NW == 1) {
            int pos;
            float cx, cy, cz;
            lookup(wmax, pos, cx, cy, cz);
            ox = cx; oy = cy; oz = cz;
            if (lane == 0) {
                idx[j] = pos;
                if (new_xyz) { new_xyz[j * 3 + 0] = cx; new_xyz[j * 3 + 1] = cy; new_xyz[j * 3 + 2] = cz; }
            }
        } else if constexpr (ONEX) {
            int pos;
            float cx, cy, cz;
            lookup(wmax, pos, cx, cy, cz);
            const int buf = j & 1;
            if (lane == 0) {
                s_rec[buf][w] = make_float4(wmax, cx, cy, cz);
                s_pos[buf][w] = pos;
            }
            V3PROF(1)
            lds_barrier();
            V3PROF(2)
            const float4 rec = s_rec[buf][lane & (NW - 1)];
            const int rpos = s_pos[buf][lane & (NW - 1)];
            const float vmax = lanes_max<NW>(rec.x);
            const uint64_t eqw = __ballot(rec.x == vmax);
            const int sel = (int)__builtin_ctzll(eqw);            // lowest wave among ties
            ox = readlane_f(rec.y, sel);
            oy = readlane_f(rec.z, sel);
            oz = readlane_f(rec.w, sel);
            const int wpos = __builtin_amdgcn_readlane(rpos, sel);   // outside the branch: every lane's rpos must be loaded
            V3PROF(3)
            if (u == 0) {
                idx[j] = wpos;
                if (new_xyz) { new_xyz[j * 3 + 0] = ox; new_xyz[j * 3 + 1] = oy; new_xyz[j * 3 + 2] = oz; }
            }
            V3PROF(4)
        } else {
            if (lane == 0) s_wmax[w] = wmax;
            V3PROF(1)
            lds_barrier();
            V3PROF(2)
            const float v = s_wmax[lane & (NW - 1)];
            const float vmax = lanes_max<NW>(v);
            const uint64_t eqw = __ballot(v == vmax);
            const int ww = (int)__builtin_ctzll(eqw);            // lowest wave among ties
            V3PROF(3)
            if (w == ww) {   // wave-uniform: ONLY the winning wave looks the winner up
                float cx, cy, cz;
                lookup(readlane_f(vmax, 0), pend_pos, cx, cy, cz);
                if (lane == 0) s_rec[0][0] = make_float4(cx, cy, cz, 0.f);
                pend = true;
                V3PROF(4)
            }
            V3PROF(7)
            lds_barrier();
            V3PROF(5)
            const float4 c4 = s_rec[0][0];
            ox = c4.x; oy = c4.y; oz = c4.z;
        }
    }
    if constexpr (DUO) { if (grp == 0) lds_barrier(); }      // pairs with the second scene's last barrier
    if constexpr (!ONEX && NW > 1) {
        if (pend && lane == 0) {
            idx[m - 1] = pend_pos;
            if (new_xyz) { new_xyz[(m - 1) * 3 + 0] = ox; new_xyz[(m - 1) * 3 + 1] = oy; new_xyz[(m - 1) * 3 + 2] = oz; }
        }
    }

    V3PROF_STORE
    // positions -> point indices, off the critical path.  idx[] was written by lane 0 of changing waves:
    // make those global stores visible to the whole workgroup first.
    __syncthreads();
    __threadfence_block();
    for (int j = 1 + u; j < m; j += NT) {
        const int p = idx[j];
        const int rb = p / S, sl = p - rb * S;
        idx[j] = v3_bitrev(rb, log2bs) + sl * bs;
    }
    if (temp) {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int p = u * PPT + s;
            const int rb = p / S, sl = p - rb * S;
            const int k = v3_bitrev(rb, log2bs) + sl * bs;
            if ((rb < bs) && (k < n)) temp[k] = t[s];
        }
    }
}

template <int PPT, int NT, bool ZLDS, bool ONEX = false, bool DUO = false>
static void launch_v3(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, float *new_xyz, int bs, int log2bs, int S,
                      hipStream_t st) {
    constexpr size_t lds = ZLDS ? (size_t)(DUO ? 2 : 1) * (PPT + (PPT == 32 ? 4 : 0)) * NT * sizeof(float) : 0;   // z (+ the last y quad)
    if constexpr (ZLDS) {
        (void)raise_lds_cap((const void *)fps_v3_kernel<PPT, NT, ZLDS, ONEX, DUO>, lds, "furthest_point_sampling(v3)");   // (a failure surfaces at the launch check)
    }
    constexpr int prio_mode = 1;       // chain priority + sweep priority of the first scene (0 / 2 / start-up offsets were A/B runs: round 2)
    hipLaunchKernelGGL((fps_v3_kernel<PPT, NT, ZLDS, ONEX, DUO>), dim3(DUO ? b / 2 : b), dim3(DUO ? 2 * NT : NT), lds, st, xyz, temp, idx, new_xyz, n, m, bs, log2bs, S,
                       prio_mode);
}

// R = number of tie-order positions (bs * S); pair = more scenes than CUs (two workgroups per CU pay)
// returns false when the shape is not covered (caller falls through to the streaming kernel)
bool fps_v3_launch(int b, int n, int m, const float *xyz, float *temp, int32_t *idx, float *new_xyz, int bs, int log2bs, int S,
                   long R, bool pair, hipStream_t st) {
#if WS3D_DIST_MODE == 1
    return false;     // the hand-scheduled sweep spells the two contracted forms only; the un-contracted build uses fps.hip
#endif
#define V3(P, T, Z) launch_v3<P, T, Z>(b, n, m, xyz, temp, idx, new_xyz, bs, log2bs, S, st)
    // small clouds (<= 2048 positions: one to four waves) stay on fps.hip's kernels: with so few waves a step is a serial
    // instruction stream at ~4.2 clk per instruction, and their packed-math sweep issues fewer instructions (measured:
    // 0.42 vs 0.53 us/step at 1024 points)
    if (R <= 2048) return false;
    if (R <= 4096) V3(8, 512, false);
    else if (R <= 8192) V3(16, 512, false);
    else if (R <= 16384) {
        // (measured and removed in round 4's clean-up, see profiles/r02_fps_ab_old_vs_v3.txt: a one-exchange variant, 256 / 512 /
        // 1024-thread geometries, two scenes per WORKGROUP half a step out of phase -- 6.55-6.85 vs 6.14 ms for 512 scenes)
        if (pair) V3(32, 512, true);
        else V3(16, 1024, false);
    } else return false;
#undef V3
    return true;
}

}  // namespace ws3d
