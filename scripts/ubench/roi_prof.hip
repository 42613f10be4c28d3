// Wall-clock timeline of roipool3d workgroups at the c5 shape (scan end / copy end per workgroup),
// via the ROI_PROF hooks.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -DWS3D_ROI_PROF
//   scripts/ubench/roi_prof.hip -o scratch/roi_prof
#include "../../ws3d_amd/csrc/core.hip"
#include "../../ws3d_amd/csrc/roipool3d.hip"
#include <algorithm>
#include <random>
#include <vector>
int main() {
    const int B = 8, N = 65536, M = 512, C = 128, S = 512;
    std::mt19937 g(1);
    std::uniform_real_distribution<float> ux(-40, 40), uy(-3, 3), uz(0, 70), ua(-3.14f, 3.14f);
    std::vector<float> h((size_t)B * N * 3), hb((size_t)B * M * 7);
    for (size_t i = 0; i < h.size(); i += 3) { h[i] = ux(g); h[i + 1] = uy(g); h[i + 2] = uz(g); }
    for (size_t i = 0; i < hb.size(); i += 7) { hb[i] = ux(g); hb[i + 1] = 1.7f; hb[i + 2] = uz(g); hb[i + 3] = 2.5f; hb[i + 4] = 3.6f; hb[i + 5] = 5.9f; hb[i + 6] = ua(g); }
    float *xyz, *boxes, *feat, *out; int *empty;
    hipMalloc(&xyz, h.size() * 4); hipMalloc(&boxes, hb.size() * 4); hipMalloc(&feat, (size_t)B * N * C * 4);
    hipMalloc(&out, (size_t)B * M * S * (3 + C) * 4); hipMalloc(&empty, (size_t)B * M * 4);
    hipMemcpy(xyz, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(boxes, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemset(feat, 0, (size_t)B * N * C * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(empty, 0, (size_t)B * M * 4);
        hipEventRecord(e0);
        int rc = ws3d_roipool3d(B, N, M, C, S, xyz, boxes, feat, out, empty, nullptr, nullptr);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rc) { printf("rc %d %s\n", rc, ws3d_last_error()); return 1; }
    }
    const int wgs = B * ((M + 3) / 4);
    std::vector<long long> p((size_t)wgs * 4);
    hipMemcpyFromSymbol(p.data(), HIP_SYMBOL(g_roi_prof), p.size() * 8);
    long long t0 = p[0];
    for (int i = 0; i < wgs; ++i) t0 = std::min(t0, p[i * 4]);
    std::vector<double> st, sc, cp, fr;
    for (int i = 0; i < wgs; ++i) { st.push_back((p[i*4] - t0) / 100.0); sc.push_back((p[i*4+1] - p[i*4+3]) / 100.0); fr.push_back((p[i*4+3] - p[i*4]) / 100.0); cp.push_back((p[i*4+2] - p[i*4+1]) / 100.0); }
    auto q = [](std::vector<double> v, double f) { std::sort(v.begin(), v.end()); return v[(size_t)(f * (v.size() - 1))]; };
    printf("kernel %.3f ms (events), %d workgroups\n", ms, wgs);
    printf("start  us: min %.1f med %.1f p90 %.1f max %.1f\n", q(st, 0), q(st, .5), q(st, .9), q(st, 1));
    printf("frames us: min %.1f med %.1f p90 %.1f max %.1f\n", q(fr, 0), q(fr, .5), q(fr, .9), q(fr, 1));
    printf("scan   us: min %.1f med %.1f p90 %.1f max %.1f\n", q(sc, 0), q(sc, .5), q(sc, .9), q(sc, 1));
    printf("copy   us: min %.1f med %.1f p90 %.1f max %.1f\n", q(cp, 0), q(cp, .5), q(cp, .9), q(cp, 1));
    return 0;
}
