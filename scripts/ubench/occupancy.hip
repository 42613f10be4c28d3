// How many workgroups with X KB of LDS does a gfx950 CU really keep resident?  Every workgroup
// spins for a fixed wall time; the launch has `per_cu` workgroups per CU, so the kernel takes
// ceil(per_cu / resident) spins.  hipcc --offload-arch=gfx950 -O2 occupancy.hip -o occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(long long ticks, int *sink) {
    extern __shared__ int lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    if (lds[(threadIdx.x + 1) % blockDim.x] == -1) *sink = 1;
}

int main() {
    int dev_cus = 0;
    hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0);
    int *sink; hipMalloc(&sink, 4);
    const long long ticks = 100 * 100;  // wall_clock64 runs at 100 MHz: 100 us
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d\n", dev_cus);
    for (int threads : {256, 512, 1024}) {
        for (int kb : {1, 8, 16, 20, 32, 35, 40, 64, 80, 160}) {
            const int per_cu = 8;
            hipFuncSetAttribute((const void *)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            hipLaunchKernelGGL(spin_kernel, dim3(dev_cus * per_cu), dim3(threads), kb * 1024, 0, ticks, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(spin_kernel, dim3(dev_cus * per_cu), dim3(threads), kb * 1024, 0, ticks, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            printf("threads %4d lds %3d KB: %.3f ms for %d WG/CU of 0.1 ms => ~%.1f resident per CU\n", threads, kb, ms,
                   per_cu, per_cu / (ms / 0.1));
        }
    }
    return 0;
}
