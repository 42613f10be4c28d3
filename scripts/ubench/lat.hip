// Micro-latency probes for the FPS reduction design (gfx950).  Each probe runs R repetitions
// of a dependent sequence in ONE workgroup and reports cycles per repetition (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define R 2000

__device__ __forceinline__ float readlane_f(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }

template <int MODE>
__global__ void probe(float *out, long long *cyc, float seed) {
    __shared__ float lds[1024];
    __shared__ unsigned long long cell[4];
    const int tid = threadIdx.x, lane = tid & 63;
    lds[tid % 1024] = seed + tid;
    if (tid < 4) cell[tid] = 0;
    __syncthreads();
    float v = seed + lane * 0.001f;
    long long t0 = clock64();
    for (int r = 0; r < R; ++r) {
        if (MODE == 0) {  // 16 dependent VALU fma
#pragma unroll
            for (int i = 0; i < 16; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
        } else if (MODE == 1) {  // 4 DPP max stages
            float r2;
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r2) : "v"(v));
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(r2));
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(r2) : "v"(v));
            asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(r2));
            v += 1.0f;
        } else if (MODE == 2) {  // readlane -> VALU dependent round trip x4
#pragma unroll
            for (int i = 0; i < 4; ++i) v = v * 0.5f + readlane_f(v, (i * 17) & 63);
        } else if (MODE == 3) {  // ballot + ff1 + readlane with computed lane
            const unsigned long long m = __ballot(v > 0.5f);
            const int l = (int)__builtin_ctzll(m | (1ull << 63));
            v = v * 0.5f + readlane_f(v, l);
        } else if (MODE == 4) {  // dependent LDS read (pointer chase through values)
            v = lds[((int)v) & 1023] + 1.0f;
        } else if (MODE == 5) {  // LDS write -> wait -> read (same wave)
            lds[tid] = v;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            v = lds[(tid + 1) & 1023] + 1.0f;
        } else if (MODE == 6) {  // LDS atomic max u64 (all lanes of row winners ~ 4 per wave) + wait + read
            if ((lane & 15) == 0) __hip_atomic_fetch_max(&cell[r & 3], (unsigned long long)(r * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            v += (float)(unsigned)cell[r & 3];
        } else if (MODE == 7) {  // barrier only (lgkm wait + s_barrier)
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v += 1.0f;
        } else if (MODE == 8) {  // write + barrier + read (cross-wave exchange)
            if (lane == 0) lds[tid >> 6] = v;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            v += lds[(lane & 15)];
        } else if (MODE == 10) {  // 32x dependent {v_cmp_gt ; v_cndmask value ; v_cndmask index}
            float best = -1.0f; int bi = 0;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float x = v + (float)(i * 7 % 11);
                const bool gt = x > best;
                bi = gt ? i : bi;
                best = gt ? x : best;
            }
            v = best * 0.001f + (float)bi;
        } else if (MODE == 11) {  // 32x dependent v_max_f32 (same inputs)
            float best = -1.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float x = v + (float)(i * 7 % 11);
                asm("v_max_f32 %0, %1, %2" : "=v"(best) : "v"(best), "v"(x));
            }
            v = best * 0.001f;
        } else if (MODE == 12) {  // 32 independent v_cmp_eq -> SGPR masks, scalar OR + ff1
            unsigned long long any = 0; int slot = 0;
            const float key = readlane_f(v, 5) + 3.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const float x = v + (float)(i * 7 % 11);
                const unsigned long long mk = __ballot(x == key);
                slot = (any == 0 && mk != 0) ? i : slot;
                any |= mk;
            }
            v = v * 0.5f + (float)slot + (float)__builtin_ctzll(any | (1ull << 63));
        } else if (MODE == 13) {  // throughput: 8 independent chains x 8 fma = 64 VALU
            float a[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) a[c] = v + c;
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int c = 0; c < 8; ++c) a[c] = __builtin_fmaf(a[c], 1.0001f, 0.5f);
            v = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
        } else if (MODE == 14) {  // throughput: 8 independent chains x 4 x (cmp + 2 cndmask) = 96 VALU
            float bv[8]; int bi[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) { bv[c] = v + c; bi[c] = c; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float x = lds[(c * 4 + i + lane) & 1023];
                    const bool gt = x > bv[c];
                    bi[c] = gt ? i : bi[c];
                    bv[c] = gt ? x : bv[c];
                }
            float sv = 0; int si = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) { sv += bv[c]; si += bi[c]; }
            v = sv * 0.01f + si;
        } else if (MODE == 15) {  // throughput: 8 independent chains x 4 x (xor, min_u32, lshl_add, min_u32) = 128 VALU int ops
            unsigned k[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) k[c] = 1000u + c;
            const unsigned mbits = __float_as_uint(v);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const unsigned x = __float_as_uint(lds[(c * 4 + i + lane) & 1023]) ^ mbits;
                    const unsigned y = min(x, 1u);
                    k[c] = min(k[c], (y << 6) + (unsigned)(c * 4 + i));
                }
            unsigned sk = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) sk += k[c];
            v = v * 0.5f + (float)(sk & 1023);
        } else if (MODE == 9) {  // movrel-style dynamic register index (uniform)
            typedef float f16 __attribute__((ext_vector_type(16)));
            f16 a; 
#pragma unroll
            for (int i = 0; i < 16; ++i) a[i] = v + i;
            const int s = __builtin_amdgcn_readfirstlane(((int)v) & 15);
            v = a[s] * 0.5f;
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + tid] = v;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int threads) {
    float *out; long long *cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    probe<MODE><<<1, threads>>>(out, cyc, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    probe<MODE><<<1, threads>>>(out, cyc, 1.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s threads=%4d  %8.1f clk/rep  %7.1f ns/rep  (kernel %.3f ms, clk counter %.2f GHz)\n", name, threads, (double)c / R, ms * 1e6 / R, ms, c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int th : {64, 512, 1024}) {
        run<0>("16 dependent v_fma", th);
        run<1>("4 DPP max stages (+1 add)", th);
        run<2>("4x readlane->VALU round trips", th);
        run<3>("ballot+ff1+readlane(computed lane)", th);
        run<4>("dependent LDS read", th);
        run<5>("LDS write, wait, read", th);
        run<6>("LDS atomic max u64 (4/wave) + wait + read", th);
        run<7>("waitcnt+s_barrier", th);
        run<8>("lane0 write + barrier + read", th);
        run<9>("readfirstlane + movrel index", th);
        run<10>("32x dep cmp+2cndmask (+32 add)", th);
        run<11>("32x dep v_max (+32 add)", th);
        run<12>("32x indep cmp_eq->sgpr + scalar", th);
        run<13>("TP 64 fma (8 chains)", th);
        run<14>("TP 32 lds + 96 cmp/cndmask (8 chains)", th);
        run<15>("TP 32 lds + 128 int ops (8 chains)", th);
    }
    return 0;
}
