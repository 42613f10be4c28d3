// Where do the workgroups of a 2-per-CU launch land, and which wave slots do they get?  (HW_REG_HW_ID / XCC_ID)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void probe(int *out, int spin) {
    extern __shared__ char smem[];
    __shared__ int s_min, s_max;
    if (threadIdx.x == 0) { s_min = 1 << 20; s_max = -1; }
    __syncthreads();
    const int hw = (int)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_ID, all 32 bits
    const int xcc = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);    // XCC_ID
    const int slot = hw & 15;
    if ((threadIdx.x & 63) == 0) { atomicMin(&s_min, slot); atomicMax(&s_max, slot); }
    __syncthreads();
    long long t0 = clock64();
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;      // stay resident
    smem[threadIdx.x] = (char)v;
    if (threadIdx.x == 0) {
        out[blockIdx.x * 8 + 0] = hw; out[blockIdx.x * 8 + 1] = xcc; out[blockIdx.x * 8 + 2] = s_min; out[blockIdx.x * 8 + 3] = s_max;
        out[blockIdx.x * 8 + 4] = (int)(t0 & 0x7fffffff);
    }
}
int main() {
    const int B = 512;
    int *d; (void)hipMalloc(&d, B * 8 * 4);
    (void)hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    probe<<<B, 512, 65536>>>(d, 200000);
    (void)hipDeviceSynchronize();
    std::vector<int> h(B * 8); (void)hipMemcpy(h.data(), d, B * 8 * 4, hipMemcpyDeviceToHost);
    std::map<int, std::vector<int>> cu;     // (xcc, se, sh, cu) -> blocks
    int n0 = 0;
    for (int b = 0; b < B; ++b) {
        const int hw = h[b * 8], xcc = h[b * 8 + 1];
        const int cuid = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cu[(xcc << 12) | (se << 8) | (sh << 4) | cuid].push_back(b);
        n0 += h[b * 8 + 2] == 0;
        if (b < 6 || b >= B - 3) printf("block %3d: hw=%08x xcc=%d se=%d sh=%d cu=%2d simd=%d slot(min,max)=(%d,%d)\n", b, hw, xcc, se, sh, cuid, (hw >> 4) & 3, h[b * 8 + 2], h[b * 8 + 3]);
    }
    int pairs = 0, good = 0;
    for (auto &kv : cu) {
        if (kv.second.size() == 2) {
            ++pairs;
            const int a = kv.second[0], b2 = kv.second[1];
            good += (h[a * 8 + 2] == 0) != (h[b2 * 8 + 2] == 0);
            if (pairs <= 4) printf("CU %05x: blocks %d (slots %d..%d) and %d (slots %d..%d)\n", kv.first, a, h[a * 8 + 2], h[a * 8 + 3], b2, h[b2 * 8 + 2], h[b2 * 8 + 3]);
        }
    }
    printf("distinct CUs %zu, CUs holding exactly 2 blocks %d, of which exactly one has min slot 0: %d; blocks with min slot 0: %d of %d\n",
           cu.size(), pairs, good, n0, B);
    return 0;
}
