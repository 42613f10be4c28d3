// Per-segment cycle anatomy of the fps_v3 step (s_memtime hooks compiled in with -DWS3D_FPS_PROF).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DWS3D_FPS_PROF scripts/ubench/fps3_prof.hip -o /tmp/fps3_prof
//   /tmp/fps3_prof <batch> <pair 0|1>
#include "../../ws3d_amd/csrc/core.hip"
#include "../../ws3d_amd/csrc/fps_v3.hip"
#include <random>
#include <vector>
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, pair = argc > 2 ? atoi(argv[2]) : 0, N = 16384, M = 4096;
    std::vector<float> h((size_t)16 * N * 3);
    std::mt19937 g(1);
    std::uniform_real_distribution<float> ux(-40, 40), uy(-3, 3), uz(0, 70);
    for (size_t i = 0; i < h.size(); i += 3) { h[i] = ux(g); h[i + 1] = uy(g); h[i + 2] = uz(g); }
    float *xyz, *nx; int *idx;
    (void)hipMalloc(&xyz, (size_t)B * N * 12); (void)hipMalloc(&nx, (size_t)B * M * 12); (void)hipMalloc(&idx, (size_t)B * M * 4);
    for (int b = 0; b < B; ++b) (void)hipMemcpy(xyz + (size_t)b * N * 3, h.data() + (size_t)(b % 16) * N * 3, (size_t)N * 12, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        ws3d::fps_v3_launch(B, N, M, xyz, nullptr, idx, nx, 1024, 10, 16, 16384, pair != 0, nullptr);
        (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("B=%d pair=%d: %.3f ms, %.3f us/step\n", B, pair, ms, ms * 1e3 / (M - 1));
#ifdef WS3D_FPS_PROF
    long long prof[4 * 16 * 8];
    (void)hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_fps3_prof), sizeof(prof));
    const char *names[8] = {"sweep", "max+wavemax+write", "barrier1", "read+reduce", "lookup(winner)", "barrier2", "read coords+loop", "pre-b2"};
    for (int pb = 0; pb < 4; ++pb) {
        for (int w : {0, 3, 7}) {
            long long tot = 0;
            printf("blockslot %d wave %2d:", pb, w);
            for (int i = 0; i < 8; ++i) { printf(" %s=%.0f", names[i], (double)prof[(pb * 16 + w) * 8 + i] / (M - 1)); tot += prof[(pb * 16 + w) * 8 + i]; }
            printf("  total=%.0f ticks/step\n", (double)tot / (M - 1));
        }
    }
#endif
    return 0;
}
