// chain_mlp.hip -- SharedMLP chains with the activations kept in REGISTERS between the layers and the weights resident in LDS
// (round 6).  Replaces, where the shapes fit, the LDS-tiled kernels of gemm_pool.hip (pgather_gemm3_compact: SA2's two scales) that
// ran at 0.26 of the fp32 matrix peak: a 64-row tile per workgroup, the weights re-staged through LDS per tile with a workgroup
// barrier every 16 k-steps, the activations written to LDS and read back between the layers, 11 VALU instructions per MFMA.
//
// Here a WAVE owns 32 rows from the gather to the pooled atomics and never meets a barrier after the weights are in LDS:
//   * layers 1 .. L-1 are computed TRANSPOSED, out^T = W^T x^T: A = the weights (row = output channel, from LDS), B = the
//     activations (col = the wave's row, from registers).  The accumulator of v_mfma_f32_32x32x2_f32 then holds, in lane (p, h),
//     channels 8 q + 4 h + i of row p in registers 4 q + i -- and eight v_permlane32_swap_b32 turn those 16 registers into the 16
//     operand registers of the next layer's k-pairs (lane (p, h) <- channel 2 s + h) IN ASCENDING k ORDER.  The operand of a
//     k-pair is the same register whether it is used as B (next transposed layer) or as A (the last layer, below).
//   * the LAST layer runs the straight way round, out = x W (A = those registers, B = the weights): lane = output channel,
//     registers = rows -- the layout compact_pool.h pools over registers and reduces with one atomic per centre.
//   * every dot product is the fmaf chain of the kernels this replaces: ascending k, two per matrix instruction, zero padding
//     behind o2 -- a * b is commutative, so swapping the operands changes no bit.  The pooled rows are BIT-IDENTICAL to
//     ws3d_pgather_gemm3_compact's (and so to the dense kernels': tests/test_gpu_parity.py).
//   * the weights of BOTH scales of the level sit in LDS (SA2: 123 KB) in the order the lanes read them -- one ds_read_b128 per
//     lane feeds four k-steps of one 32-channel block -- loaded once per workgroup; one workgroup per CU, 16 waves, which take
//     32-row tiles of either scale from ONE ticket counter (no tail of a static split, no empty workgroups for the rows the lists
//     did not fill, a workgroup that gets its CU late finds the work done).
// Reference semantics: pointnet2_modules.py:38-44 (SharedMLP + max_pool over nsample), pytorch_utils.py:20-32.
#include <algorithm>

#include "common.h"
#include "compact_pool.h"

namespace ws3d {

typedef float chain_f16 __attribute__((ext_vector_type(16)));
#ifndef WS3D_CHAIN_G3
#define WS3D_CHAIN_G3 1
#endif
constexpr int CHAIN_G3 = WS3D_CHAIN_G3;        // 32-channel blocks of the last layer accumulated per pass over k

// accumulator tile (lane (p, h): register 4 q + i = channel 8 q + 4 h + i) -> ops[t] = (half 0: channel 2 t, half 1: channel 2 t + 1)
__device__ __forceinline__ void chain_pair_operands(const chain_f16 &c, float (&ops)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // permlane32_swap(a, b): lanes 32-63 of a <-> lanes 0-31 of b.  r[0] = (a.low, b.low), r[1] = (a.high, b.high)
        const auto r01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[4 * q + 0]), __float_as_uint(c[4 * q + 1]), false, false);
        const auto r23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(c[4 * q + 2]), __float_as_uint(c[4 * q + 3]), false, false);
        ops[4 * q + 0] = __uint_as_float(r01[0]);      // channels 8 q + 0, 8 q + 1
        ops[4 * q + 1] = __uint_as_float(r23[0]);      // 8 q + 2, 8 q + 3
        ops[4 * q + 2] = __uint_as_float(r01[1]);      // 8 q + 4, 8 q + 5
        ops[4 * q + 3] = __uint_as_float(r23[1]);      // 8 q + 6, 8 q + 7
    }
}

struct ChainScale {
    const float *pmat, *xyz, *new_xyz;
    const int32_t *rowc, *rowsrc, *total;
    const float *w1x, *b1, *w2t, *b2, *w3t, *b3;
    float *out;
    long limit;
    int o2, n, m, p_stride, relu1, relu2, out_stride, j2;      // j2 = ceil(o2 / 32)
};

// LDS block of one scale (floats): w1x [4][64] (row 3 zero) | b1 [64] | b2 [32 J2] | b3 [128] | W2L [8][2][32 J2][4] | W3L [4 J2][2][128][4]
// W2L[blk][h][o][e] = w2t[k = 2 (4 blk + e) + h][o]  (o >= o2: 0);  W3L[blk][h][o][e] = w3t[k = 2 (4 blk + e) + h][o]  (k >= o2: 0)
__host__ __device__ constexpr int chain_lds_floats(int j2) { return 4 * 64 + 64 + 32 * j2 + 128 + 64 * 32 * j2 + 32 * j2 * 128; }

__device__ __forceinline__ void chain_load_scale(const ChainScale &a, float *lds, int tid, int nthreads) {
    const int o2p = 32 * a.j2;
    float *w1s = lds, *b1s = w1s + 256, *b2s = b1s + 64, *b3s = b2s + o2p, *w2l = b3s + 128, *w3l = w2l + 64 * o2p;
    for (int i = tid; i < 256; i += nthreads) w1s[i] = i < 192 ? a.w1x[i] : 0.f;
    for (int i = tid; i < 64; i += nthreads) b1s[i] = a.b1 ? a.b1[i] : 0.f;
    for (int i = tid; i < o2p; i += nthreads) b2s[i] = (a.b2 && i < a.o2) ? a.b2[i] : 0.f;
    for (int i = tid; i < 128; i += nthreads) b3s[i] = a.b3 ? a.b3[i] : 0.f;
    for (int i = tid; i < 64 * o2p; i += nthreads) {
        const int k = i / o2p, o = i - k * o2p;
        const int s = k >> 1, h = k & 1;
        w2l[(((s >> 2) * 2 + h) * o2p + o) * 4 + (s & 3)] = o < a.o2 ? a.w2t[(long)k * a.o2 + o] : 0.f;
    }
    for (int i = tid; i < o2p * 128; i += nthreads) {
        const int k = i >> 7, o = i & 127;
        const int s = k >> 1, h = k & 1;
        w3l[(((s >> 2) * 2 + h) * 128 + o) * 4 + (s & 3)] = k < a.o2 ? a.w3t[(long)k * 128 + o] : 0.f;
    }
}

// one 32-row tile of a scale: gather + layer 1 (P row + xyz term), layer 2, layer 3 + pool.  O1 = 64, O3 = 128, O2 <= 32 J2.
template <int J2>
__device__ __forceinline__ void chain3_tile(const ChainScale &a, const float *__restrict__ lds, const long tile, const long T, const int h, const int c) {
    constexpr int O2P = 32 * J2;
    const float *w1s = lds, *b1s = w1s + 256, *b2s = b1s + 64, *b3s = b2s + O2P;
    const float4 *w2l = reinterpret_cast<const float4 *>(b3s + 128), *w3l = w2l + 64 * O2P / 4;
    // ---- the rows: compact row -> (centre, source point) -> P row (64 channels: this half's 32) + centred coordinates
    chain_f16 acc1[2];
    float bxy, bz;
    {
        const long t = min(tile * 32 + c, T - 1);            // rows behind the end repeat the last one (never pooled)
        const int cm = a.rowc[t];
        const int src = a.rowsrc[t];
        const size_t pnt = (size_t)(cm / a.m) * a.n + (size_t)src;
        const float *prow = a.pmat + pnt * a.p_stride + 4 * h;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(prow + 32 * j + 8 * q);
                acc1[j][4 * q + 0] = v.x; acc1[j][4 * q + 1] = v.y; acc1[j][4 * q + 2] = v.z; acc1[j][4 * q + 3] = v.w;
            }
        const float *pr = a.xyz + pnt * 3, *cr = a.new_xyz + (size_t)cm * 3;
        const float dx = pr[0] - cr[0], dy = pr[1] - cr[1], dz = pr[2] - cr[2];
        bxy = h ? dy : dx;
        bz = h ? 0.f : dz;
    }
    // ---- layer 1: + (dx, dy, dz) . w1x, bias, ReLU
    float ops1[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const float wa0 = w1s[h * 64 + 32 * j + c], wa1 = w1s[(2 + h) * 64 + 32 * j + c];      // (row 3 is zero)
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa0, bxy, acc1[j], 0, 0, 0);
        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa1, bz, acc1[j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4 *>(b1s + 32 * j + 8 * q + 4 * h);
            const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = acc1[j][4 * q + i] + bb[i];
                if (a.relu1) y = y < 0.f ? 0.f : y;
                acc1[j][4 * q + i] = y;
            }
        }
        chain_pair_operands(acc1[j], ops1[j]);
    }
    // ---- layer 2 (transposed): 32 k-steps, J2 output blocks
    float ops2[J2][16];
    {
        chain_f16 acc2[J2];
#pragma unroll
        for (int jo = 0; jo < J2; ++jo)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[jo][i] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
            float4 w[J2];
#pragma unroll
            for (int jo = 0; jo < J2; ++jo) w[jo] = w2l[(blk * 2 + h) * O2P + 32 * jo + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float b = ops1[blk >> 2][4 * (blk & 3) + e];
#pragma unroll
                for (int jo = 0; jo < J2; ++jo) {
                    const float wa = e == 0 ? w[jo].x : e == 1 ? w[jo].y : e == 2 ? w[jo].z : w[jo].w;
                    acc2[jo] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa, b, acc2[jo], 0, 0, 0);
                }
            }
            if (blk & 1) __builtin_amdgcn_sched_barrier(0);      // keeps the LDS reads of later blocks from piling up in registers
        }
#pragma unroll
        for (int jo = 0; jo < J2; ++jo) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bv = *reinterpret_cast<const float4 *>(b2s + 32 * jo + 8 * q + 4 * h);
                const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float y = acc2[jo][4 * q + i] + bb[i];
                    if (a.relu2) y = y < 0.f ? 0.f : y;
                    acc2[jo][4 * q + i] = y;
                }
            }
            chain_pair_operands(acc2[jo], ops2[jo]);
        }
    }
    // ---- layer 3 (straight) + pool, CHAIN_G3 32-channel blocks per pass over k
    int cen[16];                                                 // the centres of the 16 consecutive rows this half pools
    compact_centres16(a.rowc, tile * 32 + 16 * h, T, cen);
#pragma unroll
    for (int g = 0; g < 4 / CHAIN_G3; ++g) {
        chain_f16 acc3[CHAIN_G3];
#pragma unroll
        for (int j = 0; j < CHAIN_G3; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc3[j][i] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4 * J2; ++blk) {
            float4 w[CHAIN_G3];
#pragma unroll
            for (int j = 0; j < CHAIN_G3; ++j) w[j] = w3l[(blk * 2 + h) * 128 + 32 * CHAIN_G3 * g + 32 * j + c];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float x = ops2[blk >> 2][4 * (blk & 3) + e];
#pragma unroll
                for (int j = 0; j < CHAIN_G3; ++j) {
                    const float wb = e == 0 ? w[j].x : e == 1 ? w[j].y : e == 2 ? w[j].z : w[j].w;
                    acc3[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, wb, acc3[j], 0, 0, 0);
                }
            }
            if (blk & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < CHAIN_G3; ++j) {
            const int col = 32 * CHAIN_G3 * g + 32 * j + c;
            compact_pool_atomic(acc3[j], b3s[col], cen, a.out + col, a.out_stride);
        }
    }
}

#ifndef WS3D_CHAIN_THREADS
#define WS3D_CHAIN_THREADS 768
#endif
constexpr int CHAIN_THREADS = WS3D_CHAIN_THREADS;      // 12 waves, 3 per SIMD (<= 168 registers)

#ifndef WS3D_CHAIN_ABL
#define WS3D_CHAIN_ABL 0       // timing ablations (NOT the operator): 1 = no weight staging, 2 = no stealing, 4 = static tile assignment (no tickets)
#endif
constexpr int CHAIN_TICKET_STRIDE = 32;
constexpr int CHAIN_COUNTERS = 32;      // ticket counters: tiles t = k (mod 32) are handed out by counter k (one address serialises its atomics:
                                        // 4,432 tickets on ONE counter cost the first form of this kernel ~45 us of a 136 us launch)

// The ticket walk of a wave (shared by the kernels below).  Tiles k, k + 32, k + 64, .. of a launch's tile space belong to counter k
// (the counters sit CHAIN_TICKET_STRIDE ints apart: one cache line each; a bit mask of dry counters behind them).  A wave starts at
// its home counter and, when that one runs dry, moves on to the counters that are still live (two memory-side round trips per move:
// the tail only); every counter is reached whatever the grid, so any number of workgroups completes the launch.  The ticket of the
// NEXT tile is requested before the current tile's work, so its round trip runs under that work.
struct ChainTickets {
    int *ticket;
    long tiles;
    int k, next, lane;
    __device__ __forceinline__ long subset(int kk) const { return (tiles - kk + CHAIN_COUNTERS - 1) / CHAIN_COUNTERS; }
    __device__ __forceinline__ void start(int *t, long n, int home, int lane_) {
        ticket = t; tiles = n; k = home % CHAIN_COUNTERS; lane = lane_; next = 0;
        if (lane == 0) next = atomicAdd(ticket + k * CHAIN_TICKET_STRIDE, 1);
    }
    // -> the next tile of this wave, or -1; requests the ticket after it
    __device__ __forceinline__ long take() {
        unsigned *dry = reinterpret_cast<unsigned *>(ticket + CHAIN_COUNTERS * CHAIN_TICKET_STRIDE);
        const unsigned never = tiles >= CHAIN_COUNTERS ? 0u : ~0u << (unsigned)tiles;
        long i = __builtin_amdgcn_readfirstlane(next);
        while (k >= tiles || i >= subset(k)) {
            unsigned m = 0;
            if (lane == 0) m = atomicOr(dry, 1u << k) | (1u << k) | never;
            m = __builtin_amdgcn_readfirstlane(m);
#if WS3D_CHAIN_ABL & 2
            m = ~0u;
#endif
            if (m == ~0u) return -1;
            const unsigned rot = (m >> ((k + 1) & 31)) | (k == 31 ? 0u : m << (31 - k));      // bit j = counter (k + 1 + j) % 32
            k = (k + 1 + __builtin_ctz(~rot)) & 31;
            int t2 = 0;
            if (lane == 0) t2 = atomicAdd(ticket + k * CHAIN_TICKET_STRIDE, 1);
            i = __builtin_amdgcn_readfirstlane(t2);
        }
        const long tile = k + i * CHAIN_COUNTERS;
        if (lane == 0) next = atomicAdd(ticket + k * CHAIN_TICKET_STRIDE, 1);
        return tile;
    }
};
constexpr int CHAIN_TICKET_INTS = CHAIN_COUNTERS * CHAIN_TICKET_STRIDE + 32;      // one ticket block (counters + dry mask, padded)

// blob -> LDS, 16 bytes per lane, the start rotated per workgroup: 256 workgroups sweeping the same 124 KB from the same first
// address at the same time queue on one L2 channel after the other
__device__ __forceinline__ void chain_stage_blob(const float *__restrict__ blob, float *lds, int floats, int tid, int rot) {
    const int nvec = floats >> 2;
    const float4 *src = reinterpret_cast<const float4 *>(blob);
    float4 *dst = reinterpret_cast<float4 *>(lds);
    const int start = (int)(((long)rot * nvec) / CHAIN_COUNTERS);
    for (int i0 = 0; i0 < nvec; i0 += 4 * CHAIN_THREADS) {
        float4 v[4];
        int idx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * CHAIN_THREADS + tid;
            idx[u] = i < nvec ? (i + start) % nvec : -1;
            if (idx[u] >= 0) v[u] = src[idx[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (idx[u] >= 0) dst[idx[u]] = v[u];
    }
}

__global__ __launch_bounds__(256) void chain_pack_kernel(const ChainScale a, float *__restrict__ blob) {
    chain_load_scale(a, blob, threadIdx.x, 256);
}

// grid = workgroups (one per CU fits: 124 KB of LDS at SA2), CHAIN_THREADS / 64 waves each; ticket[0 .. ws3d_chain_mlp3_ticket_ints()) must be ZERO on entry
template <int J2A, int J2B>
__global__ __launch_bounds__(CHAIN_THREADS) void chain_mlp3_pair_kernel(const ChainScale a0, const ChainScale a1, const int nscales, const float *__restrict__ blob0,
                                                                        const float *__restrict__ blob1, int *__restrict__ ticket) {
    extern __shared__ __attribute__((aligned(16))) float chain_smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    long T0 = *a0.total, T1 = nscales > 1 ? (long)*a1.total : 0;
    if (a0.limit >= 0 && T0 > a0.limit) T0 = 0;               // beyond the limit the dense kernels run instead (launch gates)
    if (nscales > 1 && a1.limit >= 0 && T1 > a1.limit) T1 = 0;
    const long tiles0 = (T0 + 31) / 32, tiles = tiles0 + (T1 + 31) / 32;
    ChainTickets tk;
    tk.start(ticket, tiles, (int)blockIdx.x, lane);            // first ticket: its round trip runs under the staging of the weights
    float *lds0 = chain_smem, *lds1 = chain_smem + chain_lds_floats(J2A);
#if !(WS3D_CHAIN_ABL & 1)
    if (T0 > 0) chain_stage_blob(blob0, lds0, chain_lds_floats(J2A), tid, (int)(blockIdx.x % CHAIN_COUNTERS));
    if (T1 > 0) chain_stage_blob(blob1, lds1, chain_lds_floats(J2B), tid, (int)(blockIdx.x % CHAIN_COUNTERS));
#endif
    __syncthreads();
#if WS3D_CHAIN_ABL & 4
    for (long tile = (long)blockIdx.x * (CHAIN_THREADS / 64) + (tid >> 6); tile < tiles; tile += (long)gridDim.x * (CHAIN_THREADS / 64)) {
        if (tile < tiles0) chain3_tile<J2A>(a0, lds0, tile, T0, h, c);
        else chain3_tile<J2B>(a1, lds1, tile - tiles0, T1, h, c);
    }
    return;
#endif
    for (long tile = tk.take(); tile >= 0; tile = tk.take()) {
        if (tile < tiles0) chain3_tile<J2A>(a0, lds0, tile, T0, h, c);
        else chain3_tile<J2B>(a1, lds1, tile - tiles0, T1, h, c);
    }
}

}  // namespace ws3d

static int chain_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

static int chain_scale_from(const ws3d_compact_mlp_args &q, ws3d::ChainScale &a, bool rows) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(q.pmat) | reinterpret_cast<uintptr_t>(q.out);
    bool ok = q.o1 == 64 && q.o2 > 0 && q.o2 <= 96 && q.o3 == 128 && q.w1x && q.w2t && q.w3t;
    if (rows)
        ok = ok && q.max_rows > 0 && q.max_rows <= 0x3fffffffL && q.b > 0 && q.n > 0 && q.m > 0 && q.p_stride >= 64 && !(q.p_stride & 3) && !(al & 15) && q.pmat &&
             q.xyz && q.new_xyz && q.rowc && q.rowsrc && q.total && q.out && q.out_stride >= 128;
    if (!ok) {
        ws3d::set_error("ws3d_chain_mlp3: block not covered (rows<=%ld o1=%d o2=%d o3=%d p_stride=%d; o1 = 64, o2 <= 96, o3 = 128, 16-byte aligned P rows)",
                        q.max_rows, q.o1, q.o2, q.o3, q.p_stride);
        return WS3D_E_UNSUPPORTED;
    }
    a = ws3d::ChainScale{q.pmat, q.xyz, q.new_xyz, q.rowc, q.rowsrc, q.total, q.w1x, q.b1, q.w2t, q.b2, q.w3t, q.b3, q.out, q.limit,
                         q.o2, q.n, q.m, q.p_stride, q.relu1, q.relu2, q.out_stride, (q.o2 + 31) / 32};
    return WS3D_OK;
}

extern "C" int ws3d_chain_mlp3_ticket_ints(void) { return ws3d::CHAIN_TICKET_INTS; }

extern "C" size_t ws3d_chain_mlp3_blob_floats(int o2) { return o2 > 0 && o2 <= 96 ? (size_t)ws3d::chain_lds_floats((o2 + 31) / 32) : 0; }

// The weights of one scale in the order the kernel's lanes read them out of LDS (w1x, b1, b2, b3, w2t and w3t of `scale`; its row
// arguments are not read): ws3d_chain_mlp3_blob_floats(o2) floats at `blob` (16-byte aligned).  Once per weight set.
extern "C" int ws3d_chain_mlp3_pack(const ws3d_compact_mlp_args *scale, float *blob, ws3d_stream_t stream) {
    using namespace ws3d;
    if (!scale || !blob || (reinterpret_cast<uintptr_t>(blob) & 15)) { set_error("ws3d_chain_mlp3_pack: invalid argument"); return WS3D_E_INVALID; }
    ChainScale a;
    if (int rc = chain_scale_from(*scale, a, false)) return rc;
    hipLaunchKernelGGL(chain_pack_kernel, dim3(1), dim3(256), 0, as_stream(stream), a, blob);
    return check_launch("ws3d_chain_mlp3_pack");
}

// The whole SharedMLP (three layers + pool) of one or two ball-query scales of a set-abstraction level over their compact rows:
// the argument blocks of ws3d_compact_mlp_pair / ws3d_pgather_gemm3_compact (o1 = 64, o2 <= 96, o3 = 128: SA2 of the Stage-1 network),
// scale1 may be NULL; blob0 / blob1 = ws3d_chain_mlp3_pack of the scales' weights.  ticket: ws3d_chain_mlp3_ticket_ints() int32, ZERO on entry; workgroups = 0
// picks ws3d_tune key 0, else one per CU (a caller with many batches in flight asks for fewer: ws3d_amd/pipeline.py).
extern "C" int ws3d_chain_mlp3(const ws3d_compact_mlp_args *p0, const ws3d_compact_mlp_args *p1, const float *blob0, const float *blob1, int32_t *ticket,
                               int workgroups, ws3d_stream_t stream) {
    using namespace ws3d;
    if (!p0 || !ticket || !blob0 || (p1 && !blob1) || ((reinterpret_cast<uintptr_t>(blob0) | reinterpret_cast<uintptr_t>(blob1)) & 15)) {
        set_error("ws3d_chain_mlp3: invalid argument");
        return WS3D_E_INVALID;
    }
    const ws3d_compact_mlp_args *ps[2] = {p0, p1};
    ChainScale a[2] = {};
    const int nscales = p1 ? 2 : 1;
    size_t lds = 0;
    long max_tiles = 0;
    for (int i = 0; i < nscales; ++i) {
        if (int rc = chain_scale_from(*ps[i], a[i], true)) return rc;
        lds += sizeof(float) * (size_t)chain_lds_floats(a[i].j2);
        max_tiles += (ps[i]->max_rows + 31) / 32;
    }
    if (nscales == 1) { a[1] = a[0]; blob1 = blob0; }
    if (lds > 160 * 1024) { set_error("ws3d_chain_mlp3: %zu B of LDS", lds); return WS3D_E_UNSUPPORTED; }
    long wgs = workgroups > 0 ? workgroups : (g_tune[TUNE_CHAIN_WGS] > 0 ? g_tune[TUNE_CHAIN_WGS] : chain_cu_count());
    wgs = std::max(1L, std::min(wgs, (max_tiles + CHAIN_THREADS / 64 - 1) / (CHAIN_THREADS / 64)));
    hipStream_t st = as_stream(stream);
#define WS3D_CHAIN_GO(JA, JB)                                                                                                       \
    do {                                                                                                                            \
        if (int rc = raise_lds_cap((const void *)chain_mlp3_pair_kernel<JA, JB>, lds, "ws3d_chain_mlp3")) return rc;                \
        hipLaunchKernelGGL((chain_mlp3_pair_kernel<JA, JB>), dim3((unsigned)wgs), dim3(CHAIN_THREADS), lds, st, a[0], a[1], nscales, blob0, blob1, ticket); \
    } while (0)
    const int key = a[0].j2 * 4 + a[1].j2;
    switch (key) {
    case 1 * 4 + 1: WS3D_CHAIN_GO(1, 1); break;
    case 1 * 4 + 2: WS3D_CHAIN_GO(1, 2); break;
    case 1 * 4 + 3: WS3D_CHAIN_GO(1, 3); break;
    case 2 * 4 + 1: WS3D_CHAIN_GO(2, 1); break;
    case 2 * 4 + 2: WS3D_CHAIN_GO(2, 2); break;
    case 2 * 4 + 3: WS3D_CHAIN_GO(2, 3); break;
    case 3 * 4 + 1: WS3D_CHAIN_GO(3, 1); break;
    case 3 * 4 + 2: WS3D_CHAIN_GO(3, 2); break;
    default: WS3D_CHAIN_GO(3, 3); break;
    }
#undef WS3D_CHAIN_GO
    return check_launch("ws3d_chain_mlp3");
}

