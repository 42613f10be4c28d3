// A lead, measured: C = relu?(A W) for the head / FP shape of the Stage-1 step (M = 131072 rows, K = N = 128) two ways --
//   f32   v_mfma_f32_32x32x2_f32, the path every own SharedMLP / FP / head kernel uses (157 TFLOP/s peak on MI355X);
//   split each fp32 operand written EXACTLY as three bf16 pieces (x = x1 + x2 + x3, 8 significant bits each), the six largest of the nine
//         partial products on v_mfma_f32_32x32x16_bf16 (a3 b1, a2 b2, a1 b3, a2 b1, a1 b2, a1 b1; bf16 x bf16 is exact in the fp32
//         accumulator): 6 / 16 of the fp32 instruction's time for a product that is accurate to fp32's own rounding level.
// Reports ms per launch, TFLOP/s of useful (fp32-equivalent) work, and the error of both against a float64 product on 2048 rows.
// Both kernels: 256 threads = 4 waves, a wave owns 32 rows x all 128 columns (four 32 x 32 accumulators), W staged once per workgroup in
// LDS (the split form: its three bf16 pieces, k-contiguous per column), A read from global memory, persistent workgroups over row tiles.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/bf16x6 scripts/ubench/bf16x6_gemm.hip && /tmp/bf16x6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int K = 128, N = 128;
constexpr int WS = N + 4;          // fp32 W rows in LDS: [K][WS]
constexpr int KS = K + 8;          // bf16 W columns in LDS: [3][N][KS] (k contiguous; 272-byte rows: 16-byte aligned, bank-spread)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void gemm_f32(const float *__restrict__ A, const float *__restrict__ W, float *__restrict__ C, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ws = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < K * N; i += 256) ws[(i / N) * WS + (i % N)] = W[i];
    __syncthreads();
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t row = (size_t)t * 128 + w * 32 + r;
        const float *ar = A + row * K;
        floatx16 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;
#pragma unroll 4
        for (int k4 = 0; k4 < K; k4 += 4) {
            const float4 q = *reinterpret_cast<const float4 *>(ar + k4);       // the row's k4 .. k4 + 3; this lane feeds k4 + h and k4 + 2 + h
            const float a0 = h ? q.y : q.x, a1 = h ? q.w : q.z;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, ws[(k4 + h) * WS + nb * 32 + r], acc[nb], 0, 0, 0);
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, ws[(k4 + 2 + h) * WS + nb * 32 + r], acc[nb], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const size_t orow = (size_t)t * 128 + w * 32 + (v & 3) + 8 * (v >> 2) + 4 * h;
                C[orow * N + nb * 32 + r] = acc[nb][v];
            }
    }
}

// x = p1 + p2 + p3 (+ a remainder below 2^-24 |x|): each piece is the bf16 nearest to what the pieces before it left; every
// subtraction is exact in fp32 (the piece agrees with the minuend in its leading bits)
__device__ __forceinline__ void split3(float x, __bf16 &p1, __bf16 &p2, __bf16 &p3) {
    p1 = (__bf16)x;
    const float r1 = x - (float)p1;
    p2 = (__bf16)r1;
    const float r2 = r1 - (float)p2;
    p3 = (__bf16)r2;
}

__global__ __launch_bounds__(256) void gemm_split(const float *__restrict__ A, const __bf16 *__restrict__ Wp, float *__restrict__ C, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16 *ws = reinterpret_cast<__bf16 *>(smem);                             // [3][N][KS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, g = lane >> 5;
    for (int i = tid; i < 3 * N * K; i += 256) {                               // Wp is [3][N][K]
        const int p = i / (N * K), rem = i - p * N * K;
        ws[(p * N + rem / K) * KS + (rem % K)] = Wp[i];
    }
    __syncthreads();
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const size_t row = (size_t)t * 128 + w * 32 + r;
        const float *ar = A + row * K;
        floatx16 acc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[nb][v] = 0.f;
#pragma unroll 2
        for (int kc = 0; kc < K; kc += 16) {
            // this lane's operand: 8 consecutive k of its row (lanes 32 .. 63: the upper half of the 16-chunk)
            const float4 q0 = *reinterpret_cast<const float4 *>(ar + kc + 8 * g), q1 = *reinterpret_cast<const float4 *>(ar + kc + 8 * g + 4);
            const float x[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            bf16x8 a1, a2, a3;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                __bf16 p1, p2, p3;
                split3(x[j], p1, p2, p3);
                a1[j] = p1; a2[j] = p2; a3[j] = p3;
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int col = nb * 32 + r;
                const bf16x8 b1 = *reinterpret_cast<const bf16x8 *>(ws + (0 * N + col) * KS + kc + 8 * g);
                const bf16x8 b2 = *reinterpret_cast<const bf16x8 *>(ws + (1 * N + col) * KS + kc + 8 * g);
                const bf16x8 b3 = *reinterpret_cast<const bf16x8 *>(ws + (2 * N + col) * KS + kc + 8 * g);
                floatx16 c = acc[nb];
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, c, 0, 0, 0);       // smallest terms first
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
                acc[nb] = c;
            }
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const size_t orow = (size_t)t * 128 + w * 32 + (v & 3) + 8 * (v >> 2) + 4 * g;
                C[orow * N + nb * 32 + r] = acc[nb][v];
            }
    }
}

static uint16_t bf16_rne(float f) {                      // host: round to nearest even, as the device conversion
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    const int M = 131072, tiles = M / 128;
    std::vector<float> hA((size_t)M * K), hW((size_t)K * N);
    srand(7);
    auto rnd = []() { float s = 0.f; for (int i = 0; i < 4; ++i) s += (float)((double)rand() / RAND_MAX) - 0.5f; return s * 1.7f; };       // ~unit variance
    for (auto &v : hA) v = rnd() * (rand() % 16 == 0 ? 8.f : 1.f);                  // activations: a heavy tail
    for (auto &v : hW) v = rnd() / sqrtf((float)K);
    std::vector<uint16_t> hWp((size_t)3 * N * K);                                    // [3][N][K]
    for (int k = 0; k < K; ++k)
        for (int n = 0; n < N; ++n) {
            float x = hW[(size_t)k * N + n];
            for (int p = 0; p < 3; ++p) {
                const uint16_t h = bf16_rne(x);
                hWp[((size_t)p * N + n) * K + k] = h;
                x -= bf16_to_f(h);
            }
        }
    float *A, *W, *C0, *C1; __bf16 *Wp;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&Wp, hWp.size() * 2));
    CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(Wp, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice));
    const size_t lds_f = sizeof(float) * K * WS, lds_s = 2 * (size_t)3 * N * KS;
    CK(hipFuncSetAttribute((const void *)gemm_f32, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f));
    CK(hipFuncSetAttribute((const void *)gemm_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double flop = 2.0 * M * N * K;
    float ms[2];
    for (int which = 0; which < 2; ++which) {
        for (int grid : {256, 512}) {
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (int i = 0; i < 10; ++i) {
                    if (which == 0) hipLaunchKernelGGL(gemm_f32, dim3(grid), dim3(256), lds_f, 0, A, W, C0, tiles);
                    else hipLaunchKernelGGL(gemm_split, dim3(grid), dim3(256), lds_s, 0, A, Wp, C1, tiles);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            CK(hipGetLastError());
            float t; CK(hipEventElapsedTime(&t, e0, e1)); t /= 10;
            printf("%-6s grid %4d: %.4f ms per launch, %.1f TFLOP/s of fp32-equivalent work (%zu B of LDS per workgroup)\n", which ? "split" : "f32", grid, t,
                   flop / (t * 1e-3) / 1e12, which ? lds_s : lds_f);
            ms[which] = t;
        }
    }
    // accuracy on the first 2048 rows against float64
    const int R = 2048;
    std::vector<float> h0((size_t)R * N), h1((size_t)R * N);
    CK(hipMemcpy(h0.data(), C0, h0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), C1, h1.size() * 4, hipMemcpyDeviceToHost));
    double e_f = 0, e_s = 0, m_f = 0, m_s = 0, scale = 0, d01 = 0;
    for (int i = 0; i < R; ++i)
        for (int n = 0; n < N; ++n) {
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)i * K + k] * (double)hW[(size_t)k * N + n]; ref += p; mag += fabs(p); }
            const double df = fabs(h0[(size_t)i * N + n] - ref) / mag, ds = fabs(h1[(size_t)i * N + n] - ref) / mag;      // relative to sum |a_k w_k|: the condition-free measure
            e_f += df * df; e_s += ds * ds; m_f = fmax(m_f, df); m_s = fmax(m_s, ds);
            d01 = fmax(d01, fabs((double)h0[(size_t)i * N + n] - (double)h1[(size_t)i * N + n]) / mag);
            scale += 1;
        }
    printf("error / sum|a w| against float64 on %d x %d outputs:  f32 rms %.3e max %.3e   split rms %.3e max %.3e   |f32 - split| max %.3e  (fp32 epsilon 5.96e-08)\n",
           R, N, sqrt(e_f / scale), m_f, sqrt(e_s / scale), m_s, d01);
    printf("split / f32 time: %.3f\n", ms[1] / ms[0]);
    return 0;
}
