// What the fp32 matrix cores sustain on this chip: a register-only loop of v_mfma_f32_32x32x2_f32 (or 16x16x4) on every SIMD,
// ACC independent accumulators per wave, WPS waves per SIMD, random or zero operands.  Reports TFLOP/s from the wall clock and
// the effective shader clock = s_memtime cycles / wall time (the chip clocks to its power budget: MI355X_MICROARCH.md, DVFS).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak scripts/ubench/mfma_f32_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int ACC, bool SMALL>
__global__ __launch_bounds__(256) void spin(int iters, const float *__restrict__ in, float *__restrict__ out, long long *__restrict__ cyc) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    float a = in[tid & 4095], b = in[(tid * 7 + 1) & 4095];
    long long t0 = __builtin_readcyclecounter();
    float sum = 0.f;
    if (!SMALL) {
        f16v acc[ACC];
        for (int j = 0; j < ACC; ++j) for (int v = 0; v < 16; ++v) acc[j][v] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < ACC; ++j) for (int v = 0; v < 16; ++v) sum += acc[j][v];
    } else {
        f4v acc[ACC];
        for (int j = 0; j < ACC; ++j) for (int v = 0; v < 4; ++v) acc[j][v] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < ACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
        }
        for (int j = 0; j < ACC; ++j) for (int v = 0; v < 4; ++v) sum += acc[j][v];
    }
    long long t1 = __builtin_readcyclecounter();
    out[tid] = sum;
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

template <int ACC, bool SMALL>
void run(const char *name, int wps, bool zeros, int iters) {
    const int waves = 256 * 4 * wps, threads = waves * 64;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = zeros ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
    float *in, *out; long long *cyc;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, threads * 4); hipMalloc(&cyc, waves * 8);
    hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {            // the last (warm, clocks settled) repetition is reported
        hipEventRecord(e0);
        hipLaunchKernelGGL((spin<ACC, SMALL>), dim3(threads / 256), dim3(256), 0, 0, iters, in, out, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> c(waves); hipMemcpy(c.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : c) mean += v; mean /= waves;
    const double flop = (double)waves * iters * ACC * (SMALL ? 2048.0 : 4096.0);
    // s_memtime / readcyclecounter ticks at a constant 100 MHz on this part: the shader clock follows from the instruction count instead
    const double mfma_clk = SMALL ? 32.0 : 64.0;
    const double busy_cycles = (double)iters * ACC * wps * mfma_clk;      // per SIMD if the pipe never idles
    printf("%-10s acc %d  waves/SIMD %d  %-6s: %7.3f ms  %6.1f TFLOP/s  pipe-bound clock >= %.2f GHz (cycles at 100%% pipe use / wall)  counter %.0f ticks\n", name, ACC, wps,
           zeros ? "zeros" : "random", ms, flop / ms / 1e9, busy_cycles / (ms * 1e-3) / 1e9, mean);
    hipFree(in); hipFree(out); hipFree(cyc);
}

int main() {
    const int it = 40000;
    for (int z = 0; z < 2; ++z) {
        run<1, false>("32x32x2", 1, z, it * 4); run<4, false>("32x32x2", 1, z, it); run<4, false>("32x32x2", 2, z, it / 2); run<2, false>("32x32x2", 4, z, it / 2);
        run<4, true>("16x16x4", 1, z, it * 2); run<4, true>("16x16x4", 2, z, it); run<8, true>("16x16x4", 4, z, it / 4);
    }
    return 0;
}
