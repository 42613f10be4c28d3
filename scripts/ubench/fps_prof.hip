// Per-segment cycle anatomy of one FPS step (wave 0..7 of block 0), via clock64() hooks
// compiled into fps.hip with -DWS3D_FPS_PROF.  Build: hipcc --offload-arch=gfx950 -O3
//   -ffp-contract=off -DWS3D_FPS_PROF scripts/ubench/fps_prof.hip -o scripts/ubench/fps_prof
#include "../../ws3d_amd/csrc/core.hip"
#include "../../ws3d_amd/csrc/fps.hip"
#include <vector>
#include <random>
namespace ws3d { int fps_bucket_launch(int, int, int, const float *, float *, int32_t *, float *, int, int, int, hipStream_t) { return -4; } }
int main() {
    const int B = 8, N = 16384, M = 4096;
    std::vector<float> h((size_t)B * N * 3);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> ux(-40, 40), uy(-3, 3), uz(0, 70);
    for (size_t i = 0; i < h.size(); i += 3) { h[i] = ux(g); h[i + 1] = uy(g); h[i + 2] = uz(g); }
    float *xyz, *nx; int *idx;
    hipMalloc(&xyz, h.size() * 4); hipMalloc(&nx, (size_t)B * M * 12); hipMalloc(&idx, (size_t)B * M * 4);
    hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        ws3d_furthest_point_sampling_gather(B, N, M, xyz, nullptr, idx, nx, nullptr);
        hipDeviceSynchronize();
    }
    long long prof[16 * 8];
    hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_fps_prof), sizeof(prof));
    const char *names[8] = {"loop/store", "sweep", "wave argmax", "coords(movrel)", "lds write", "barrier", "read+reduce", "-"};
    for (int w = 0; w < 8; ++w) {
        printf("wave %d:", w);
        long long tot = 0;
        for (int i = 0; i < 7; ++i) { printf(" %s=%.0f", names[i], (double)prof[w * 8 + i] / (M - 1)); tot += prof[w * 8 + i]; }
        printf("  total=%.0f clk/step\n", (double)tot / (M - 1));
    }
    return 0;
}
