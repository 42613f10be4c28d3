// What does the STORE PATTERN of the fused query + group kernel reach by itself?  (c2 block: 512 scenes x 4096 centres x 64 samples,
// (B, 4, M, ns) grouped rows + (B, M, ns) lists = 2.68 GB.)  One workgroup of 256 threads per tile of 64 centres writes five 16 KB runs
// (four channel planes 1 MB apart + the list) with 16-byte stores, in the kernel's own blockIdx -> (scene, tile) order -- no search, no
// gathers, constant data.  Variants: streaming (nt) / plain stores; 1 / 2 / 4 tiles per workgroup.   hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4v __attribute__((ext_vector_type(4)));
template <bool NT, int TPW>
__global__ __launch_bounds__(256) void pattern(float *out, int *idx, int nb, int m, int ns) {
    const int tiles = m / 64;
    const int g = blockIdx.x, j = g >> 3;
    const int b = (j / (tiles / TPW)) * 8 + (g & 7);
    const int tile0 = (j % (tiles / TPW)) * TPW;
    const size_t plane = (size_t)m * ns;
    for (int t = 0; t < TPW; ++t) {
        const int m0 = (tile0 + t) * 64;
        float *ob = out + (size_t)b * 4 * plane + (size_t)m0 * ns;
        int *ib = idx + ((size_t)b * m + m0) * ns;
        for (int q = threadIdx.x; q < 64 * ns / 4; q += 256) {
            const f4v v = {1.f, 2.f, 3.f, (float)q};
            if (NT) {
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(ob) + q);
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(ob + plane) + q);
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(ob + 2 * plane) + q);
                __builtin_nontemporal_store(v, reinterpret_cast<f4v *>(ob + 3 * plane) + q);
            } else {
                reinterpret_cast<f4v *>(ob)[q] = v; reinterpret_cast<f4v *>(ob + plane)[q] = v;
                reinterpret_cast<f4v *>(ob + 2 * plane)[q] = v; reinterpret_cast<f4v *>(ob + 3 * plane)[q] = v;
            }
            reinterpret_cast<int4 *>(ib)[q] = make_int4(q, q, q, q);
        }
    }
}
template <bool NT, int TPW> static void run(const char *name, float *out, int *idx, int B, int M, int NS) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> ts;
    for (int it = 0; it < 9; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL((pattern<NT, TPW>), dim3(B * (M / 64) / TPW), dim3(256), 0, 0, out, idx, B, M, NS);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (it >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const double gb = (double)B * M * NS * 5 * 4 / 1e9;
    printf("%-28s %.3f ms  %.2f TB/s (%.2f GB)\n", name, ts[ts.size() / 2], gb / ts[ts.size() / 2], gb);
}
int main() {
    const int B = 512, M = 4096, NS = 64;
    float *out; int *idx;
    hipMalloc(&out, (size_t)B * 4 * M * NS * 4); hipMalloc(&idx, (size_t)B * M * NS * 4);
    run<true, 1>("nt, 1 tile per workgroup", out, idx, B, M, NS);
    run<false, 1>("plain, 1 tile per workgroup", out, idx, B, M, NS);
    run<true, 2>("nt, 2 tiles per workgroup", out, idx, B, M, NS);
    run<true, 4>("nt, 4 tiles per workgroup", out, idx, B, M, NS);
    run<false, 4>("plain, 4 tiles per workgroup", out, idx, B, M, NS);
    hipMemsetAsync(out, 0, (size_t)B * 4 * M * NS * 4, 0);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a); hipMemsetAsync(out, 0, (size_t)B * 4 * M * NS * 4, 0); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s %.3f ms  %.2f TB/s\n", "hipMemsetAsync 2.15 GB", ms, (double)B * 4 * M * NS * 4 / 1e9 / ms);
    return 0;
}
